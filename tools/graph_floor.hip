// Microbenchmark: what does one kernel node cost inside a replayed hipGraph on this runtime / GPU?
//   graph of NK dependent kernels (trivial / with a dependent-load chain), replayed R times; the same eagerly.
// build: hipcc --offload-arch=gfx950 -O3 -o gpurun_out/graph_floor tools/graph_floor.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void trivial(unsigned* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
// a dependent chain of `hops` global loads (pointer chasing) like desc -> seeds -> data
__global__ void chase(const unsigned* idx, unsigned* out, int hops) {
  unsigned i = threadIdx.x;
  for (int h = 0; h < hops; ++h) i = idx[i];
  if (i == 0xFFFFFFFFu) out[0] = i;
}

int main(int argc, char** argv) {
  const int NK = argc > 1 ? atoi(argv[1]) : 10, R = 2000;
  unsigned *p, *idx;
  CK(hipMalloc(&p, 4096));
  CK(hipMemset(p, 0, 4096));
  std::vector<unsigned> h(1 << 20);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)((i * 2654435761u + 12345u) % h.size());
  CK(hipMalloc(&idx, h.size() * 4));
  CK(hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  for (int variant = 0; variant < 4; ++variant) {
    const int blocks = (variant & 1) ? 256 : 1;
    const int hops = (variant & 2) ? 3 : 0;
    auto enqueue = [&]() {
      for (int k = 0; k < NK; ++k) {
        if (hops) chase<<<blocks, 64, 0, s>>>(idx, p, hops);
        else trivial<<<blocks, 64, 0, s>>>(p);
      }
    };
    hipGraph_t g;
    hipGraphExec_t ex;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    enqueue();
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    for (int w = 0; w < 50; ++w) CK(hipGraphLaunch(ex, s));
    CK(hipStreamSynchronize(s));
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < R; ++r) CK(hipGraphLaunch(ex, s));
    auto t1 = std::chrono::steady_clock::now();
    CK(hipStreamSynchronize(s));
    auto t2 = std::chrono::steady_clock::now();
    const double enq = std::chrono::duration<double, std::micro>(t1 - t0).count() / R;
    const double tot = std::chrono::duration<double, std::micro>(t2 - t0).count() / R;
    for (int w = 0; w < 50; ++w) enqueue();
    CK(hipStreamSynchronize(s));
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < R; ++r) enqueue();
    t1 = std::chrono::steady_clock::now();
    CK(hipStreamSynchronize(s));
    t2 = std::chrono::steady_clock::now();
    const double eenq = std::chrono::duration<double, std::micro>(t1 - t0).count() / R;
    const double etot = std::chrono::duration<double, std::micro>(t2 - t0).count() / R;
    printf("NK=%d blocks=%3d hops=%d | graph: host %.2f us/replay, total %.2f us/replay = %.2f us/kernel | eager: host %.2f, total %.2f = %.2f us/kernel\n",
           NK, blocks, hops, enq, tot, tot / NK, eenq, etot, etot / NK);
    CK(hipGraphExecDestroy(ex));
    CK(hipGraphDestroy(g));
  }
  return 0;
}
