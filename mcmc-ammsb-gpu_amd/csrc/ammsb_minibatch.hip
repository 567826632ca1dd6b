// Device-side mini-batch sampling for the "Node" strategy.
//
// The reference draws every mini-batch on one host thread (mcmc/sample.cc:249-303: rand_r, an
// std::unordered_set<Edge> and two host cuckoo probes per candidate, then learner.cc:162-173 builds
// the node list through another unordered_set).  At m = 65536 that is ~10 ms per batch, four times
// the device work it feeds.  Here the non-link half runs as three small kernels:
//   (the workspace arrives with an empty table: all bytes 0xFF before the first call, left so by every call)
//   1. candidate j draws v_j from its own xorshift128+ stream, checks v_j != u and both cuckoo sets,
//      and registers (v_j -> min j) in an open-addressing table (atomicCAS / atomicMin);
//   2. a candidate is kept iff it is valid and the table names it as the first occurrence of v_j;
//      per-block keep counts;
//   3. every block sums the counts of the blocks before it and writes its kept candidates in candidate
//      order (first m of them), then the node list.
// The result is a deterministic function of (stream states, u, sets).
#include "ammsb_ctx.h"
#include "ammsb_dev.h"
#include "ammsb_step.h"

#include <math.h>

using namespace ammsb;

namespace {

constexpr int MB_BLOCK = 256;
constexpr uint64_t EMPTY = ~0ull;

struct MbWork {        // layout of the caller's workspace
  uint64_t* table;     // [H] (v << 32 | j), EMPTY = all ones
  uint32_t* cand;      // [C] v_j | valid << 31
  uint32_t* blk;       // [C / MB_BLOCK + 1] per-block keep counts, then exclusive offsets
  uint32_t H, C;       // C = candidates of this call; with a descriptor: the launch's capacity, the call's own
                       // count (<= C, multiple of MB_BLOCK) comes from desc->n_cand
  const ammsb_step_desc* desc;
};

__device__ __forceinline__ uint32_t mb_active(const MbWork& w) { return w.desc ? w.desc->n_cand : w.C; }

inline uint32_t table_size(uint32_t C) {
  uint32_t h = 1024;
  while (h < 4u * C) h <<= 1;
  return h;
}

// cap: the candidate count the workspace was sized for (fixes the table size whatever this call draws)
inline MbWork carve(void* ws, uint32_t C, uint32_t cap) {
  MbWork w;
  w.C = C;
  w.H = table_size(cap);
  w.desc = nullptr;
  char* p = static_cast<char*>(ws);
  w.table = reinterpret_cast<uint64_t*>(p);
  p += sizeof(uint64_t) * w.H;
  w.cand = reinterpret_cast<uint32_t*>(p);
  p += sizeof(uint32_t) * cap;
  w.blk = reinterpret_cast<uint32_t*>(p);
  return w;
}

__device__ __forceinline__ uint32_t mix32(uint32_t x) {  // table slot hash (murmur3 finaliser)
  x ^= x >> 16;
  x *= 0x85ebca6bu;
  x ^= x >> 13;
  x *= 0xc2b2ae35u;
  x ^= x >> 16;
  return x;
}

__device__ __forceinline__ void mb_draw_one(ammsb_seed* seeds, const MbWork& w, uint32_t j, uint32_t u, uint32_t N,
                                            const DevSet& training, const DevSet& heldout, int has_heldout) {
  ammsb_seed s = seeds[j];
  const uint32_t v = (uint32_t)fast_mod(rng_next(s), fast_mod_init(N));
  seeds[j] = s;
  bool valid = v != u;
  if (valid) {
    const uint64_t e = make_edge(u, v);
    valid = !set_has(training, e) && !(has_heldout && set_has(heldout, e));
  }
  w.cand[j] = v | (valid ? 0x80000000u : 0u);
  if (!valid) return;
  const uint64_t packed = ((uint64_t)v << 32) | j;
  uint32_t h = mix32(v) & (w.H - 1);
  for (uint32_t probes = 0; probes < w.H; ++probes) {  // H >= 4C: the table can never fill up
    const unsigned long long old =
        atomicCAS(reinterpret_cast<unsigned long long*>(&w.table[h]), (unsigned long long)EMPTY,
                  (unsigned long long)packed);
    if (old == EMPTY) break;
    if ((uint32_t)(old >> 32) == v) {
      atomicMin(reinterpret_cast<unsigned long long*>(&w.table[h]), (unsigned long long)packed);
      break;
    }
    h = (h + 1) & (w.H - 1);
  }
}

__global__ __launch_bounds__(MB_BLOCK) void mb_draw_kernel(ammsb_seed* seeds, MbWork w, uint32_t u, uint32_t N,
                                                            DevSet training, DevSet heldout, int has_heldout) {
  const uint32_t j = blockIdx.x * MB_BLOCK + threadIdx.x;
  if (w.desc) u = w.desc->u;
  if (j >= mb_active(w)) return;
  mb_draw_one(seeds, w, j, u, N, training, heldout, has_heldout);
}

__device__ __forceinline__ bool mb_keep(const MbWork& w, uint32_t j) {
  if (j >= mb_active(w)) return false;
  const uint32_t c = w.cand[j];
  if (!(c >> 31)) return false;
  const uint32_t v = c & 0x7fffffffu;
  uint32_t h = mix32(v) & (w.H - 1);
  for (uint32_t probes = 0; probes < w.H; ++probes) {
    const uint64_t t = __hip_atomic_load(&w.table[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((uint32_t)(t >> 32) == v) return (uint32_t)t == j;
    if (t == EMPTY) return false;  // cannot happen: this candidate inserted its v or met it
    h = (h + 1) & (w.H - 1);
  }
  return false;
}

template <int NT = MB_BLOCK>
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t flag, uint32_t* total) {
  __shared__ uint32_t wsum[NT / 64];
  const unsigned long long ball = __ballot(flag != 0);
  const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t before = __popcll(ball & ((1ull << lane) - 1));
  if (lane == 0) wsum[wv] = __popcll(ball);
  __syncthreads();
  uint32_t off = 0, tot = 0;
  for (int i = 0; i < NT / 64; ++i) {
    if (i < (int)wv) off += wsum[i];
    tot += wsum[i];
  }
  *total = tot;
  return off + before;
}

__global__ __launch_bounds__(MB_BLOCK) void mb_count_kernel(MbWork w) {
  const uint32_t j = blockIdx.x * MB_BLOCK + threadIdx.x;
  uint32_t total;
  block_exclusive_scan(mb_keep(w, j) ? 1u : 0u, &total);
  if (threadIdx.x == 0) w.blk[blockIdx.x] = total;
}

// Each block adds up the counts of the blocks before it (a few hundred at most) instead of waiting
// for a separate scan launch: a lone scan block cannot get a CU while update_phi fills the chip, which
// used to park the whole sampling chain behind it.
__global__ __launch_bounds__(MB_BLOCK) void mb_write_kernel(MbWork w, uint32_t u, uint32_t m, uint64_t* edges,
                                                             uint32_t* nodes, uint32_t* count_out) {
  __shared__ uint32_t part[MB_BLOCK / 64];
  if (ammsb_desc_skip(w.desc)) return;  // (block-uniform) a skipped mini-batch leaves the buffers and counters alone
  if (w.desc) u = w.desc->u;
  uint32_t before = 0;
  for (uint32_t i = threadIdx.x; i < blockIdx.x; i += MB_BLOCK) before += w.blk[i];
  for (int d = 32; d > 0; d >>= 1) before += __shfl_down(before, d, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = before;
  __syncthreads();
  uint32_t base = 0;
  for (int i = 0; i < MB_BLOCK / 64; ++i) base += part[i];
  const uint32_t j = blockIdx.x * MB_BLOCK + threadIdx.x;
  const bool keep = mb_keep(w, j);
  uint32_t total;
  const uint32_t rank = base + block_exclusive_scan(keep ? 1u : 0u, &total);
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
    count_out[0] = base + total;
    if (base + total < m) count_out[1] += 1;  // sticky: mini-batches that came up short since the caller cleared it
  }
  if (keep && rank < m) {
    const uint32_t v = w.cand[j] & 0x7fffffffu;
    edges[rank] = make_edge(u, v);
    nodes[1 + rank] = v;
  }
  if (j == 0) nodes[0] = u;
}

// Last kernel of a non-link mini-batch: (1) leaves the de-duplication table empty for the next call (the caller
// hands over a workspace filled with 0xFF once; no memset per call -- a memset node at the root of a captured
// graph is also not ordered reliably before the graph's first kernel on this runtime), (2) memory-safe tail
// when fewer than m candidates survived (the caller sees count < m and the sticky counter).
__global__ void mb_finish_kernel(uint32_t m, const uint32_t* count, uint64_t* edges, uint32_t* nodes, uint64_t* table,
                                 uint32_t H, const ammsb_step_desc* desc) {
  if (ammsb_desc_skip(desc)) return;  // nothing was drawn: the table is still empty, the buffers are not ours
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, nthreads = gridDim.x * blockDim.x;
  for (uint32_t h = tid; h < H; h += nthreads) table[h] = EMPTY;
  const uint32_t c = count[0];
  if (c >= m || c == 0) return;
  for (uint32_t r = c + tid; r < m; r += nthreads) {
    edges[r] = edges[r % c];
    nodes[1 + r] = nodes[1 + r % c];
  }
}

__global__ void mb_link_kernel(const uint64_t* offsets, const uint32_t* targets, uint32_t u, uint32_t n,
                               uint64_t* edges, uint32_t* nodes, const ammsb_step_desc* desc) {
  if (ammsb_desc_skip(desc)) return;
  if (desc) {  // captured graph: the grid covers the largest degree
    u = desc->u;
    n = desc->n_edges;
  }
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t == 0) nodes[0] = u;
  if (t >= n) return;
  const uint32_t v = targets[offsets[u] + t];
  edges[t] = make_edge(u, v);
  nodes[1 + t] = v;
}

}  // namespace

extern "C" uint32_t ammsb_minibatch_candidates(uint64_t N, uint32_t m) { return ammsb_minibatch_candidates_for(N, m, 0); }

extern "C" uint32_t ammsb_minibatch_candidates_for(uint64_t N, uint32_t m, uint64_t excluded) {
  if (m == 0 || N < 2ull * m) return 0;
  // C draws from N values leave about N (1 - exp(-C/N)) distinct ones.  Up to `excluded` of them can be invalid
  // partners of u (u itself, its training and held-out neighbours), so ask for m + excluded distinct values plus a
  // margin of 8 % + 1024 (N / 8 on small graphs) against an unlucky draw.
  const double slack = (double)N / 8.0 < 1024.0 ? (double)N / 8.0 : 1024.0;
  const double want = 1.08 * ((double)m + (double)excluded) + slack;
  if (want >= 0.95 * (double)N) return 0;
  const double c = -(double)N * log(1.0 - want / (double)N);
  uint64_t C = (uint64_t)ceil(c) + 256;
  C = (C + MB_BLOCK - 1) / MB_BLOCK * MB_BLOCK;
  return C > 0x40000000ull ? 0u : (uint32_t)C;
}

extern "C" uint64_t ammsb_minibatch_workspace_bytes(uint32_t C) {
  return sizeof(uint64_t) * table_size(C) + sizeof(uint32_t) * C + sizeof(uint32_t) * (C / MB_BLOCK + 2);
}

extern "C" int ammsb_minibatch_link(ammsb_ctx* ctx, const uint64_t* csr_offsets, const uint32_t* csr_targets,
                                    uint32_t u, uint32_t n, uint64_t* edges_out, uint32_t* nodes_out, void* stream) {
  AMMSB_CHECK_ARG(ctx, ctx && csr_offsets && csr_targets && edges_out && nodes_out, "null argument");
  AMMSB_CHECK_ARG(ctx, n > 0 && u < ctx->params.N, "vertex has no training edge");
  mb_link_kernel<<<(n + 255) / 256, 256, 0, as_stream(stream)>>>(csr_offsets, csr_targets, u, n, edges_out, nodes_out,
                                                                 nullptr);
  AMMSB_LAUNCH_CHECK(ctx);
  return AMMSB_OK;
}

int ammsb_minibatch_link_d(ammsb_ctx* ctx, const uint64_t* csr_offsets, const uint32_t* csr_targets, uint32_t n_cap,
                           uint64_t* edges_out, uint32_t* nodes_out, const ammsb_step_desc* desc, void* stream) {
  AMMSB_CHECK_ARG(ctx, ctx && csr_offsets && csr_targets && edges_out && nodes_out && desc && n_cap > 0, "null argument");
  mb_link_kernel<<<(n_cap + 255) / 256, 256, 0, as_stream(stream)>>>(csr_offsets, csr_targets, 0, 0, edges_out,
                                                                     nodes_out, desc);
  AMMSB_LAUNCH_CHECK(ctx);
  return AMMSB_OK;
}

static int minibatch_nonlink_common(ammsb_ctx* ctx, ammsb_seed* seeds, uint32_t n_candidates, uint32_t capacity,
                                    uint32_t u, uint32_t m, const ammsb_set* training_set,
                                    const ammsb_set* heldout_set, void* workspace, uint64_t* edges_out,
                                    uint32_t* nodes_out, uint32_t* count_out, const ammsb_step_desc* desc,
                                    void* stream) {
  AMMSB_CHECK_ARG(ctx, ctx && seeds && training_set && workspace && edges_out && nodes_out && count_out,
                  "null argument");
  AMMSB_CHECK_ARG(ctx, training_set->slots && training_set->num_bins > 0 && training_set->prime_idx < 4,
                  "bad training set");
  AMMSB_CHECK_ARG(ctx, !heldout_set || (heldout_set->slots && heldout_set->num_bins > 0 && heldout_set->prime_idx < 4),
                  "bad held-out set");
  AMMSB_CHECK_ARG(ctx, m > 0 && n_candidates >= m && n_candidates % MB_BLOCK == 0, "bad candidate count");
  AMMSB_CHECK_ARG(ctx, capacity >= n_candidates && capacity % MB_BLOCK == 0, "candidates exceed the workspace capacity");
  AMMSB_CHECK_ARG(ctx, u < ctx->params.N && ctx->params.N < (1ull << 31), "bad vertex / N");
  hipStream_t s = as_stream(stream);
  MbWork w = carve(workspace, n_candidates, capacity);
  w.desc = desc;
  const uint32_t nb = n_candidates / MB_BLOCK;
  const ammsb_set none = {nullptr, 1, 0};
  // (tried: draw / count / write / finish as ONE 1024-thread block for capacities <= 4096 candidates, to save three
  // launches per small mini-batch -- 33 us on one CU against 16 us + three boundaries for the four kernels: the
  // sampling chain became what bounds a C1 step, 0.0304 -> 0.0362 ms; removed)
  mb_draw_kernel<<<nb, MB_BLOCK, 0, s>>>(seeds, w, u, (uint32_t)ctx->params.N, dev_set(*training_set),
                                         dev_set(heldout_set ? *heldout_set : none), heldout_set ? 1 : 0);
  mb_count_kernel<<<nb, MB_BLOCK, 0, s>>>(w);
  mb_write_kernel<<<nb, MB_BLOCK, 0, s>>>(w, u, m, edges_out, nodes_out, count_out);
  mb_finish_kernel<<<64, 256, 0, s>>>(m, count_out, edges_out, nodes_out, w.table, w.H, desc);
  AMMSB_LAUNCH_CHECK(ctx);
  return AMMSB_OK;
}

extern "C" int ammsb_minibatch_nonlink(ammsb_ctx* ctx, ammsb_seed* seeds, uint32_t n_candidates, uint32_t capacity,
                                       uint32_t u, uint32_t m, const ammsb_set* training_set,
                                       const ammsb_set* heldout_set, void* workspace, uint64_t* edges_out,
                                       uint32_t* nodes_out, uint32_t* count_out, void* stream) {
  return minibatch_nonlink_common(ctx, seeds, n_candidates, capacity, u, m, training_set, heldout_set, workspace,
                                  edges_out, nodes_out, count_out, nullptr, stream);
}

int ammsb_minibatch_nonlink_d(ammsb_ctx* ctx, ammsb_seed* seeds, uint32_t n_candidates_cap, uint32_t m,
                              const ammsb_set* training_set, const ammsb_set* heldout_set, void* workspace,
                              uint64_t* edges_out, uint32_t* nodes_out, uint32_t* count_out,
                              const ammsb_step_desc* desc, void* stream) {
  AMMSB_CHECK_ARG(ctx, ctx && desc, "null descriptor");
  return minibatch_nonlink_common(ctx, seeds, n_candidates_cap, n_candidates_cap, 0, m, training_set, heldout_set,
                                  workspace, edges_out, nodes_out, count_out, desc, stream);
}
