// Development micro-benchmark: how fast can one wave per "node" stream n random K-float rows of a
// large table into registers on gfx950, as a function of load width, rows in flight and occupancy?
// Build: hipcc --offload-arch=gfx950 -O3 tools/gather_bench.hip -o gpurun_out/gather_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int KPT, int DEPTH, int VEC, int PAD_VGPR>
__global__ __launch_bounds__(64) void gather(const float* __restrict__ table, const uint32_t* __restrict__ idx,
                                              uint32_t n_nodes, uint32_t n, uint32_t K, float* out) {
  const uint32_t node = blockIdx.x;
  if (node >= n_nodes) return;
  const int l = threadIdx.x;
  float acc[KPT];
#pragma unroll
  for (int j = 0; j < KPT; ++j) acc[j] = 0.f;
  float buf[DEPTH][KPT];
  auto load = [&](float (&dst)[KPT], uint32_t q) {
    const uint32_t w = __builtin_amdgcn_readfirstlane(idx[node * n + q]);
    const float* row = table + (uint64_t)w * K;
    if constexpr (VEC == 1) {
#pragma unroll
      for (int j = 0; j < KPT; ++j) dst[j] = row[l + 64 * j];
    } else {
#pragma unroll
      for (int j = 0; j < KPT / 4; ++j) {
        const float4 v = *reinterpret_cast<const float4*>(row + 4 * l + 256 * j);
        dst[4 * j] = v.x; dst[4 * j + 1] = v.y; dst[4 * j + 2] = v.z; dst[4 * j + 3] = v.w;
      }
    }
  };
#pragma unroll
  for (int d = 0; d < DEPTH - 1; ++d) load(buf[d], d);
  for (uint32_t q0 = 0; q0 < n; q0 += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const uint32_t q = q0 + d, qn = q + DEPTH - 1;
      if (qn < n) load(buf[(d + DEPTH - 1) % DEPTH], qn);
      if (q < n) {
#pragma unroll
        for (int j = 0; j < KPT; ++j) acc[j] += buf[d][j];
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int j = 0; j < KPT; ++j) s += acc[j];
  if (PAD_VGPR > 0) {  // inflate register use to lower occupancy
    float pad[PAD_VGPR > 0 ? PAD_VGPR : 1];
#pragma unroll
    for (int i = 0; i < PAD_VGPR; ++i) pad[i] = s * (float)i;
#pragma unroll
    for (int i = 0; i < PAD_VGPR; ++i) asm volatile("" : "+v"(pad[i]));
#pragma unroll
    for (int i = 0; i < PAD_VGPR; ++i) s += pad[i];
  }
  out[node * 64 + l] = s;
}

template <int KPT, int DEPTH, int VEC, int PAD>
void run(const char* name, const float* table, const uint32_t* idx, uint32_t nodes, uint32_t n, uint32_t K, float* out,
         size_t lds) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e9;
  for (int r = 0; r < 4; ++r) {
    CK(hipEventRecord(a));
    gather<KPT, DEPTH, VEC, PAD><<<nodes, 64, lds>>>(table, idx, nodes, n, K, out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  const double bytes = (double)nodes * n * K * 4;
  printf("%-34s lds=%6zu  %.3f ms  %.0f GB/s\n", name, lds, best, bytes / best / 1e6);
}

int main() {
  const uint32_t N = 1000000, K = 1024, nodes = 65537, n = 33;
  float* table; uint32_t* idx; float* out;
  CK(hipMalloc(&table, (size_t)N * K * 4));
  CK(hipMemset(table, 0, (size_t)N * K * 4));
  std::vector<uint32_t> h((size_t)nodes * n);
  uint64_t s = 88172645463325252ull;
  for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint32_t)(s % N); }
  CK(hipMalloc(&idx, h.size() * 4)); CK(hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&out, (size_t)nodes * 64 * 4));
  // occupancy control through dynamic LDS per 64-thread block: 160 KB / lds = blocks per CU
  const size_t ldss[] = {0, 5000, 6600, 10000, 13000, 20000};
  for (size_t lds : ldss) {
    run<16, 2, 1, 0>("dword  depth2", table, idx, nodes, n, K, out, lds);
    run<16, 4, 1, 0>("dword  depth4", table, idx, nodes, n, K, out, lds);
    run<16, 6, 1, 0>("dword  depth6", table, idx, nodes, n, K, out, lds);
    run<16, 2, 4, 0>("dwordx4 depth2", table, idx, nodes, n, K, out, lds);
    run<16, 4, 4, 0>("dwordx4 depth4", table, idx, nodes, n, K, out, lds);
    run<16, 6, 4, 0>("dwordx4 depth6", table, idx, nodes, n, K, out, lds);
  }
  return 0;
}
