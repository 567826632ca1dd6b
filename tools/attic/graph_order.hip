// Does this runtime serialise consecutive hipGraphLaunch calls on ONE stream?  Graph A: a slow kernel that
// writes `it` into buf[0] at its end.  Graph B: a kernel that copies buf[0] into out[it].  Launch A(it), B(it)
// back to back; every out[it] must equal it.  The value `it` travels through device memory (a counter the
// slow kernel increments), as in the step descriptors of ammsb_loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void slow_writer(unsigned* counter, unsigned* buf, int spin) {
  // many blocks; the last one to finish publishes counter+1
  __shared__ unsigned dummy;
  unsigned x = threadIdx.x;
  for (int i = 0; i < spin; ++i) x = x * 1664525u + 1013904223u;
  if (x == 0xdeadbeef) dummy = x;
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
    const unsigned c = *counter + 1;
    *counter = c;
    buf[0] = c;
    buf[1] = c * 7u;
  }
}
__global__ void reader(const unsigned* buf, unsigned* out, const unsigned* counter) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const unsigned c = *counter;
    out[2 * c] = buf[0];
    out[2 * c + 1] = buf[1];
  }
}

int main() {
  const int R = 4000;
  unsigned *counter, *buf, *out;
  CK(hipMalloc(&counter, 64));
  CK(hipMalloc(&buf, 64));
  CK(hipMalloc(&out, 8 * (R + 2)));
  CK(hipMemset(counter, 0, 64));
  CK(hipMemset(buf, 0, 64));
  CK(hipMemset(out, 0, 8 * (R + 2)));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  for (int variant = 0; variant < 3; ++variant) {
    hipGraph_t ga, gb;
    hipGraphExec_t ea, eb;
    const int blocks = variant == 0 ? 1 : 1024, spin = variant == 2 ? 20000 : 2000;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    slow_writer<<<blocks, 64, 0, s>>>(counter, buf, spin);
    CK(hipStreamEndCapture(s, &ga));
    CK(hipGraphInstantiate(&ea, ga, nullptr, nullptr, 0));
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    reader<<<4, 64, 0, s>>>(buf, out, counter);
    reader<<<4, 64, 0, s>>>(buf, out, counter);
    CK(hipStreamEndCapture(s, &gb));
    CK(hipGraphInstantiate(&eb, gb, nullptr, nullptr, 0));
    CK(hipMemset(counter, 0, 64));
    CK(hipMemset(out, 0, 8 * (R + 2)));
    CK(hipDeviceSynchronize());
    for (int it = 0; it < R; ++it) {
      CK(hipGraphLaunch(ea, s));
      CK(hipGraphLaunch(eb, s));
    }
    CK(hipStreamSynchronize(s));
    std::vector<unsigned> h(2 * (R + 2));
    CK(hipMemcpy(h.data(), out, 8 * (R + 2), hipMemcpyDeviceToHost));
    int bad = 0;
    for (int it = 1; it <= R; ++it)
      if (h[2 * it] != (unsigned)it || h[2 * it + 1] != (unsigned)it * 7u) {
        if (bad < 5) printf("  variant %d it %d: got %u %u\n", variant, it, h[2 * it], h[2 * it + 1]);
        ++bad;
      }
    printf("variant %d (blocks %d, spin %d): %d of %d hand-offs wrong\n", variant, blocks, spin, bad, R);
  }
  return 0;
}
