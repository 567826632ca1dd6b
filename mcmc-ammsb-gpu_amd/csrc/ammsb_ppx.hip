// Held-out perplexity for gfx950.
//
// Replaces PerplexityCalculatorBase::operator() (mcmc/perplexity.cc:251-274): the work-group kernel
// calculate_ppx_partial_for_edge (:93-181) plus the four boost::compute / thrust reductions
// (:318-331, perplexity.cu:27-37).  The reference stages the K products of every edge in a global
// scratch buffer of H*K floats and re-reads them three times; here a lane keeps the products of the
// columns it owns in registers, the WG_SUM order is unchanged, and the four per-edge output arrays
// are folded into per-slot partial sums (binary64 for the log-likelihoods) reduced by a second
// single-block kernel in a fixed order.
#pragma clang fp contract(off)

#include <stdlib.h>

#include "ammsb_ctx.h"
#include "ammsb_dev.h"

using namespace ammsb;

namespace {

struct PpxArgs {
  const float* beta;
  ammsb_rpm pi;
  DevSet set;  // (ammsb_dev.h: the descriptor + the modulo magic)
  const uint64_t* edges;
  float* ppx_per_edge;
  double* ll_partials;               // [P, 2]
  unsigned long long* cnt_partials;  // [P, 2]
  uint32_t edge_begin, edge_end, P, K, call_count;
  float epsilon;
  // (ppx_lds_kernel) the launch reduces its own partials: every block draws a ticket when its partials are out, the
  // block that draws the last one adds them up in ppx_reduce_kernel's order and writes *out -- no second launch
  uint32_t* ticket;     // zero before the launch; the reducing block resets it
  ammsb_ppx_sums* out;  // device memory or host-mapped pinned memory
};

// The four reductions of perplexity.cc:318-331 over the P per-slot partials, in ONE fixed order whoever runs it:
// virtual thread t of 256 adds slots t, t + 256, ... ascending, then the halving tree over the 256.  Here one wave runs
// it (lane l carries virtual threads l, l + 64, l + 128, l + 192: the tree's levels 128 and 64 are in-lane adds, the
// levels 32 .. 1 go through `scratch`, >= 2 KiB of LDS the caller no longer needs) -- ppx_reduce_kernel below runs the
// same order with 256 real threads, so the folded and the two-launch forms agree bit for bit.
__device__ __forceinline__ void ppx_reduce_wave(const double* ll, const unsigned long long* cnt, uint32_t P, void* scratch,
                                                ammsb_ppx_sums* out) {
  const uint32_t l = threadIdx.x & 63u;
  double a0[4], a1[4];
  unsigned long long c0[4], c1[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    a0[j] = a1[j] = 0.0;
    c0[j] = c1[j] = 0ull;
    for (uint32_t p = l + 64u * j; p < P; p += 256u) {
      a0[j] += ll[2 * p];
      a1[j] += ll[2 * p + 1];
      c0[j] += cnt[2 * p];
      c1[j] += cnt[2 * p + 1];
    }
  }
  // level 128: s[t] += s[t + 128] (t < 128); level 64: s[t] += s[t + 64] (t < 64)
  a0[0] += a0[2]; a0[1] += a0[3]; a0[0] += a0[1];
  a1[0] += a1[2]; a1[1] += a1[3]; a1[0] += a1[1];
  c0[0] += c0[2]; c0[1] += c0[3]; c0[0] += c0[1];
  c1[0] += c1[2]; c1[1] += c1[3]; c1[0] += c1[1];
  double* s_ll = reinterpret_cast<double*>(scratch);                                   // [2][64]
  unsigned long long* s_c = reinterpret_cast<unsigned long long*>(s_ll + 128);         // [2][64]
  s_ll[l] = a0[0];
  s_ll[64 + l] = a1[0];
  s_c[l] = c0[0];
  s_c[64 + l] = c1[0];
  for (uint32_t p2 = 32; p2 > 0; p2 >>= 1) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (one wave: its LDS operations complete in order)
    __builtin_amdgcn_wave_barrier();
    if (l < p2) {
      s_ll[l] += s_ll[l + p2];
      s_ll[64 + l] += s_ll[64 + l + p2];
      s_c[l] += s_c[l + p2];
      s_c[64 + l] += s_c[64 + l + p2];
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  if (l == 0) {
    out->link_ll = s_ll[0];
    out->nonlink_ll = s_ll[64];
    out->link_cnt = s_c[0];
    out->nonlink_cnt = s_c[64];
  }
}

template <int L, int KPT>
__global__ __launch_bounds__(Group<L>::BLOCK) void ppx_kernel(const PpxArgs a) {
  using Grp = Group<L>;
  __shared__ float aux[Grp::AUX];
  const int l = Grp::lane();
  const uint32_t gs = blockIdx.x * Grp::PER_BLOCK + Grp::slot();
  const bool live = gs < a.P;
  const uint32_t K = a.K;

  float bk[KPT], omb[KPT];
#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    const uint32_t k = l + j * L;
    bk[j] = k < K ? a.beta[2 * k + 1] : 0.0f;
    omb[j] = 1.0f - bk[j];
  }

  double ll_link = 0.0, ll_non = 0.0;
  unsigned long long c_link = 0, c_non = 0;
  const uint32_t n_edges = a.edge_end - a.edge_begin;
  const uint32_t trips = (n_edges + a.P - 1) / a.P;
  int phase = 0;
  const float cm1 = (float)(a.call_count - 1), cc = (float)a.call_count;

  float pa[2][KPT], pb[2][KPT];
  uint64_t key[2] = {0, 0}, pos[2] = {0, 0};
  bool have[2] = {false, false};
  // unconditional loads: exhausted slots shadow their first edge, columns beyond K shadow column K-1
  auto fetch = [&](int b, uint32_t t) {
    const uint64_t e_raw = (uint64_t)a.edge_begin + gs + (uint64_t)t * a.P;
    have[b] = live && t < trips && e_raw < a.edge_end;
    const uint64_t e = have[b] ? e_raw : a.edge_begin;
    pos[b] = e;
    key[b] = a.edges[e];  // used as stored: no canonicalisation (perplexity.cc:45-47)
    const uint32_t u = (uint32_t)(key[b] >> 32), v = (uint32_t)(key[b] & 0xffffffffu);
    const float* ra = rpm_row(a.pi, u);
    const float* rb = rpm_row(a.pi, v);
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      const uint32_t k = l + j * L;
      const uint32_t ck = k < K ? k : K - 1;
      const float xa = __builtin_nontemporal_load(ra + ck), xb = __builtin_nontemporal_load(rb + ck);  // read once per pass
      pa[b][j] = k < K ? xa : 0.0f;
      pb[b][j] = xb;
    }
  };
  auto consume = [&](int b) {
    const bool y = set_has(a.set, key[b]);
    float s_part = 0.0f, f_part = 0.0f;
#pragma unroll
    for (int j = 0; j < KPT; ++j) {  // perplexity.cc:100-119
      const float f = pa[b][j] * pb[b][j];
      f_part += f;
      s_part += f * (y ? bk[j] : omb[j]);
    }
    const float fsum = Grp::sum(f_part, aux, phase);
    float s = Grp::sum(s_part, aux, phase);
    if (!y) {
      const float t = 1.0f - fsum;
      const float u1 = 1.0f - a.epsilon;
      s += t * u1;
    }
    if (s < 1.0e-30f) s = 1.0e-30f;
    if (have[b] && l == 0) {  // perplexity.cc:139-156
      float ppx = a.ppx_per_edge[pos[b]];
      float m = ppx * cm1;
      m = m + s;
      ppx = m / cc;
      const float ll = logf_cr(ppx);
      if (y) {
        c_link += 1;
        ll_link += (double)ll;
      } else {
        c_non += 1;
        ll_non += (double)ll;
      }
      a.ppx_per_edge[pos[b]] = ppx;
    }
  };

  fetch(0, 0);
  for (uint32_t t = 0; t < trips; t += 2) {
    fetch(1, t + 1);
    consume(0);
    fetch(0, t + 2);
    consume(1);
  }
  if (live && l == 0) {
    a.ll_partials[2 * gs] = ll_link;
    a.ll_partials[2 * gs + 1] = ll_non;
    a.cnt_partials[2 * gs] = c_link;
    a.cnt_partials[2 * gs + 1] = c_non;
  }
}

// ---------------------------------------------------------------------------------------------
// LDS-streamed form for L = 64 and K = 64 * KPT (the shape of update_phi_lds_kernel / beta_grads_lds_kernel):
// one wave per slot, both pi rows of the next edge arrive by LDS-DMA in a two-edge ring while the
// current edge is reduced; keys are loaded and probed in the held-out set 64 trips at a time, a window ahead
// (the probe -- two 64-bit modulos and two dependent 32-byte reads -- used to sit in front of every edge).
// Same arithmetic and operation order as ppx_kernel<64, KPT>.

typedef __attribute__((address_space(3))) void ppx_lds_void_t;
typedef const __attribute__((address_space(1))) void ppx_glb_void_t;

// VL = 32: the reference's default ppx_wg_size (main.cc:63) on the same one-wave-per-slot layout (VLane<32>).
template <int KPT, uint32_t D, int VL = 64, bool FOLD = false>
__global__ __launch_bounds__(64) void ppx_lds_kernel(const PpxArgs a) {
  using VLn = VLane<VL>;
  constexpr int K = 64 * KPT, HP = KPT / 2, PIECES = KPT / 4;
  extern __shared__ __align__(16) char smem[];  // [D edges][2 rows][K] floats
  float* ring = reinterpret_cast<float*>(smem);
  const int l = threadIdx.x;
  const uint32_t gs = blockIdx.x;  // the grid is exactly P blocks

  f32x2 bk[HP];
#pragma unroll
  for (int p = 0; p < HP; ++p) bk[p] = f32x2{a.beta[2 * (l + 128 * p) + 1], a.beta[2 * (l + 128 * p + 64) + 1]};

  double ll_link = 0.0, ll_non = 0.0;
  unsigned long long c_link = 0, c_non = 0;
  const uint32_t n_edges = a.edge_end - a.edge_begin;
  const uint32_t trips = gs < n_edges ? (n_edges - gs + a.P - 1) / a.P : 0;  // wave-uniform
  const float cm1 = (float)(a.call_count - 1), cc = (float)a.call_count;

  // lane i of a window holds the key, the link bit and the running mean (perplexity.cc:139-156) of trip tb + i: nothing
  // is LOADED inside the trip loop but the rows.  (The mean used to be read where it is used, by lane 0 behind the row
  // requests: hipcc's wait-count pass put a vmcnt(0) in front of its first use on every trip, which drained the next
  // edge's rows -- the ring never ran ahead; found in the disassembly in round 3, as in the gradient kernel.)
  auto load_keys = [&](uint32_t tb, unsigned long long* ymask, float* mean) -> unsigned long long {
    const bool ok = tb + l < trips;
    const uint64_t e = (uint64_t)a.edge_begin + gs + (uint64_t)(tb + l) * a.P;
    const unsigned long long key = a.edges[ok ? e : a.edge_begin];
    *mean = a.ppx_per_edge[ok ? e : a.edge_begin];
    *ymask = __ballot(set_has(a.set, key));  // used as stored: no canonicalisation (perplexity.cc:45-47)
    return key;
  };
  uint32_t tb = 0;
  unsigned long long ym = 0, ym_next = 0;
  float pm = 0.0f, pm_next = 0.0f;
  unsigned long long kv = load_keys(0, &ym, &pm), kv_next = load_keys(64, &ym_next, &pm_next);
  auto mean_of = [&](uint32_t t) -> float {
    const uint32_t rel = t - tb;
    const bool first = rel < 64u;
    const int src = __builtin_amdgcn_readfirstlane((int)(first ? rel : rel - 64u));
    return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(first ? pm : pm_next), src));
  };
  auto key_of = [&](uint32_t t, bool* y) -> unsigned long long {
    const uint32_t rel = t - tb;  // 0 .. 127 by construction
    const bool first = rel < 64u;
    const uint32_t src = first ? rel : rel - 64u;
    const unsigned long long key = wave_lane_u64(first ? kv : kv_next, src);  // (t, tb are wave-uniform)
    *y = (((first ? ym : ym_next) >> src) & 1ull) != 0;
    return key;
  };
  // aux 2 = nt: a held-out edge's rows are read once per pass; streaming them past the caches measured -7 %
  auto request = [&](uint32_t t) {
    bool y;
    const unsigned long long key = key_of(t, &y);
    const uint32_t u = __builtin_amdgcn_readfirstlane((uint32_t)(key >> 32));
    const uint32_t v = __builtin_amdgcn_readfirstlane((uint32_t)(key & 0xffffffffu));
    const float* ra = rpm_row(a.pi, u) + 4 * l;
    const float* rb = rpm_row(a.pi, v) + 4 * l;
    char* dst = smem + (t % D) * (2 * K * sizeof(float));
#pragma unroll
    for (int p = 0; p < PIECES; ++p)
      __builtin_amdgcn_global_load_lds((ppx_glb_void_t*)(ra + 256 * p), (ppx_lds_void_t*)(dst + 1024 * p), 16, 0, 2);
#pragma unroll
    for (int p = 0; p < PIECES; ++p)
      __builtin_amdgcn_global_load_lds((ppx_glb_void_t*)(rb + 256 * p),
                                       (ppx_lds_void_t*)(dst + K * sizeof(float) + 1024 * p), 16, 0, 2);
  };

  for (uint32_t t = 0; t < D - 1 && t < trips; ++t) request(t);
  for (uint32_t t = 0; t < trips; ++t) {
    const float* row_a = ring + (t % D) * 2 * K;
    const float* row_b = row_a + K;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // slot (t - 1) % D has been read for the last time
    if (t + D - 1 < trips) {
      if (t + D - 1 >= tb + 128) {  // the look-ahead leaves the two key windows: slide them
        kv = kv_next;
        ym = ym_next;
        pm = pm_next;
        tb += 64;
        kv_next = load_keys(tb + 64, &ym_next, &pm_next);
      }
      request(t + D - 1);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * 2 * PIECES) : "memory");  // edge t landed, t+1, t+2 in flight
    } else {
      const uint32_t ahead = trips - 1 - t;  // 0 .. D-2 edges still in flight behind edge t
      if (D > 2 && ahead >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PIECES) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    bool y;
    (void)key_of(t, &y);
    y = __builtin_amdgcn_readfirstlane((int)y) != 0;

    float s_part = 0.0f, f_part = 0.0f;
#pragma unroll
    for (int p = 0; p < HP; ++p) {  // perplexity.cc:100-119
      const f32x2 pa = f32x2{row_a[l + 128 * p], row_a[l + 128 * p + 64]};
      const f32x2 pb = f32x2{row_b[l + 128 * p], row_b[l + 128 * p + 64]};
      const f32x2 f = pa * pb;
      VLn::chain(f_part, f.x);
      VLn::chain(f_part, f.y);
      const f32x2 w = f * (y ? bk[p] : 1.0f - bk[p]);
      VLn::chain(s_part, w.x);
      VLn::chain(s_part, w.y);
    }
    const float fsum = VLn::tree(f_part);
    float s = VLn::tree(s_part);
    if (!y) {
      const float tt = 1.0f - fsum;
      const float u1 = 1.0f - a.epsilon;
      s += tt * u1;
    }
    if (s < 1.0e-30f) s = 1.0e-30f;
    const float mean_old = mean_of(t);
    if (l == 0) {  // perplexity.cc:139-156
      const uint64_t pos = (uint64_t)a.edge_begin + gs + (uint64_t)t * a.P;
      float ppx = mean_old;
      float m = ppx * cm1;
      m = m + s;
      ppx = m / cc;
      const float ll = logf_cr(ppx);
      if (y) {
        c_link += 1;
        ll_link += (double)ll;
      } else {
        c_non += 1;
        ll_non += (double)ll;
      }
      a.ppx_per_edge[pos] = ppx;
    }
  }
  if (l == 0) {
    a.ll_partials[2 * gs] = ll_link;
    a.ll_partials[2 * gs + 1] = ll_non;
    a.cnt_partials[2 * gs] = c_link;
    a.cnt_partials[2 * gs + 1] = c_non;
  }
  if constexpr (FOLD) {
  // The launch adds up its own partials (cdna_hip_programming.md, split-K in-launch reduction / Guideline 16): partial
  // stores drained -> agent-scope release -> ticket; the block that draws the last ticket makes one agent-scope acquire
  // and reads every slot's partials with plain loads.  One wave per block: no barrier, the "I am last" word travels
  // by readfirstlane (no second __shared__ object beside the ring).
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  uint32_t last = 0;
  if (l == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (kept: ROCm 7.2 can drop the fence's own wait)
    last = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.P - 1u ? 1u : 0u;
  }
  last = __builtin_amdgcn_readfirstlane(last);
  if (!last) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the ring's last reads are done: its first 2 KiB are the scratch
  ppx_reduce_wave(a.ll_partials, a.cnt_partials, a.P, smem, a.out);
  if (l == 0) __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // for the next launch
  }
}

// ---------------------------------------------------------------------------------------------------------
// Generic form: any K <= 1024 * (blockDim.x / 64), any power-of-two reference work-group size L <= blockDim.x / 2
// (the reference's ppx_wg_size defaults to 32, main.cc:63; at K = 4096 that is 128 columns per work-item,
// perplexity.cc:93-157).  Elementwise work over all T = blockDim.x threads, the two WG_SUMs emulated lane by lane with
// the reference's association order (vgroup_sum<2>, ammsb_dev.h); slot -> edge assignment, the per-slot accumulation
// order and every operation are ppx_kernel<L, KPT>'s: bit-identical to it wherever both run.
template <int CPT>
__global__ __launch_bounds__(512) void ppx_gen_kernel(const PpxArgs a, uint32_t L, uint32_t lgL) {
  extern __shared__ __align__(16) char smem[];  // [K] f, [K] weighted f, [2 L] lane partials, [4] sums
  const uint32_t K = a.K, T = blockDim.x, t = threadIdx.x;
  float* s_f = reinterpret_cast<float*>(smem);
  float* s_w = s_f + K;
  float* s_aux = s_w + K;
  float* s_res = s_aux + 2 * L;
  const uint32_t gs = blockIdx.x;  // the grid is exactly P blocks

  auto col = [&](int j) -> uint32_t { return t + (uint32_t)j * T; };
  auto has = [&](int j) -> bool { return t + (uint32_t)j * T < K; };
  float bk[CPT];
#pragma unroll
  for (int j = 0; j < CPT; ++j) bk[j] = has(j) ? a.beta[2 * col(j) + 1] : 0.0f;

  double ll_link = 0.0, ll_non = 0.0;
  unsigned long long c_link = 0, c_non = 0;
  const uint32_t n_edges = a.edge_end - a.edge_begin;
  const uint32_t trips = gs < n_edges ? (n_edges - gs + a.P - 1) / a.P : 0;  // block-uniform
  int phase = 0;
  const float cm1 = (float)(a.call_count - 1), cc = (float)a.call_count;

  float pa[CPT], pb[CPT], na[CPT], nb[CPT];
  auto load_rows = [&](float (&da)[CPT], float (&db)[CPT], uint32_t r) -> unsigned long long {
    const uint64_t e = (uint64_t)a.edge_begin + gs + (uint64_t)r * a.P;
    const unsigned long long key = a.edges[e];  // used as stored: no canonicalisation (perplexity.cc:45-47)
    const uint32_t u = __builtin_amdgcn_readfirstlane((uint32_t)(key >> 32));
    const uint32_t v = __builtin_amdgcn_readfirstlane((uint32_t)(key & 0xffffffffu));
    const float* ra = rpm_row(a.pi, u);
    const float* rb = rpm_row(a.pi, v);
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      const uint32_t k = col(j), ck = k < K ? k : K - 1;
      da[j] = __builtin_nontemporal_load(ra + ck);
      db[j] = __builtin_nontemporal_load(rb + ck);
    }
    return key;
  };
  unsigned long long key = 0, nkey = 0;
  if (trips > 0) key = load_rows(pa, pb, 0);
  for (uint32_t r = 0; r < trips; ++r) {
    nkey = load_rows(na, nb, r + 1 < trips ? r + 1 : r);  // unconditional: the last trip re-requests its own rows
    const bool y = set_has(a.set, key);                   // block-uniform (every thread probes the same key)
#pragma unroll
    for (int j = 0; j < CPT; ++j) {  // perplexity.cc:100-119
      const float f = (has(j) ? pa[j] : 0.0f) * pb[j];
      if (has(j)) {
        s_f[col(j)] = f;
        s_w[col(j)] = f * (y ? bk[j] : 1.0f - bk[j]);
      }
    }
    __syncthreads();
    const float* const vv[2] = {s_f, s_w};
    float sums[2];
    vgroup_sum<2>(vv, K, L, lgL, s_aux, s_res, phase, sums);
    const float fsum = sums[0];
    float s = sums[1];
    if (!y) {
      const float tt = 1.0f - fsum;
      const float u1 = 1.0f - a.epsilon;
      s += tt * u1;
    }
    if (s < 1.0e-30f) s = 1.0e-30f;
    if (t == 0) {  // perplexity.cc:139-156
      const uint64_t pos = (uint64_t)a.edge_begin + gs + (uint64_t)r * a.P;
      float ppx = a.ppx_per_edge[pos];
      float m = ppx * cm1;
      m = m + s;
      ppx = m / cc;
      const float ll = logf_cr(ppx);
      if (y) {
        c_link += 1;
        ll_link += (double)ll;
      } else {
        c_non += 1;
        ll_non += (double)ll;
      }
      a.ppx_per_edge[pos] = ppx;
    }
    key = nkey;
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      pa[j] = na[j];
      pb[j] = nb[j];
    }
  }
  if (t == 0) {
    a.ll_partials[2 * gs] = ll_link;
    a.ll_partials[2 * gs + 1] = ll_non;
    a.cnt_partials[2 * gs] = c_link;
    a.cnt_partials[2 * gs + 1] = c_non;
  }
}

constexpr uint32_t kGenMaxK = 8192;  // 512 threads x 16 columns
inline int gen_cpt(uint64_t K) { return K <= 4096 ? 8 : 16; }  // columns per thread (as in ammsb_phi.hip)

inline uint32_t gen_threads(uint64_t K, uint32_t L) {
  if (K > kGenMaxK) return 0;
  const uint32_t per_wave = 64u * (uint32_t)gen_cpt(K);
  uint32_t T = 64u * (uint32_t)((K + per_wave - 1) / per_wave);
  if (T < 2 * L) T = 2 * L;
  if (T < 64) T = 64;
  return T <= 512 ? T : 0;
}

int launch_ppx_gen(ammsb_ctx* ctx, const PpxArgs& a, uint32_t wg, hipStream_t s) {
  const uint32_t T = gen_threads(a.K, wg);
  if (!T) return AMMSB_ERANGE;
  const size_t lds = sizeof(float) * (2 * (size_t)a.K + 2 * wg + 4);
  ctx->kernel_name[AMMSB_KN_PPX] = gen_cpt(a.K) == 8 ? "ppx_gen_kernel<8>" : "ppx_gen_kernel<16>";
  if (gen_cpt(a.K) == 8) ppx_gen_kernel<8><<<a.P, T, lds, s>>>(a, wg, ilog2_u32(wg));
  else ppx_gen_kernel<16><<<a.P, T, lds, s>>>(a, wg, ilog2_u32(wg));
  AMMSB_LAUNCH_CHECK(ctx);
  return AMMSB_OK;
}

template <int KPT, int VL = 64>
int launch_ppx_lds(ammsb_ctx* ctx, const PpxArgs& a, hipStream_t s) {
  // two-edge ring: 16 KiB per wave, so the 8 slots per CU the launch asks for are resident at once (a three-edge
  // ring at 24 KiB fits 6 and needs a second round: 0.335 vs 0.250 ms at C3; 1536 slots x 3 edges ties at 0.247)
  const size_t lds = (size_t)2 * 2 * sizeof(float) * 64 * KPT;
  static const std::string name = ammsb_kname("ppx_lds_kernel<%d, 2u, %d>", KPT, VL);
  ctx->kernel_name[AMMSB_KN_PPX] = name.c_str();
  // AMMSB_PPX_FOLD=1: the launch reduces its own partials (FOLD instantiation; no ppx_reduce_kernel launch)
  static const bool fold = getenv("AMMSB_PPX_FOLD") && atoi(getenv("AMMSB_PPX_FOLD")) == 1;
  if (fold) {
    PpxArgs f = a;
    f.ticket = ctx->ppx_ticket;
    ppx_lds_kernel<KPT, 2, VL, true><<<a.P, 64, lds, s>>>(f);
    AMMSB_LAUNCH_CHECK(ctx);
    return 1;  // done, sums written
  }
  ppx_lds_kernel<KPT, 2, VL, false><<<a.P, 64, lds, s>>>(a);
  AMMSB_LAUNCH_CHECK(ctx);
  return AMMSB_OK;
}

// fixed-order reduction of the P per-slot partials: thread t adds slots t, t+256, ... ascending,
// then a halving tree over the 256 threads.
__global__ __launch_bounds__(256) void ppx_reduce_kernel(const double* ll, const unsigned long long* cnt, uint32_t P,
                                                          ammsb_ppx_sums* out) {
  __shared__ double s_ll[2][256];
  __shared__ unsigned long long s_c[2][256];
  double a0 = 0, a1 = 0;
  unsigned long long c0 = 0, c1 = 0;
  for (uint32_t p = threadIdx.x; p < P; p += 256) {
    a0 += ll[2 * p];
    a1 += ll[2 * p + 1];
    c0 += cnt[2 * p];
    c1 += cnt[2 * p + 1];
  }
  s_ll[0][threadIdx.x] = a0;
  s_ll[1][threadIdx.x] = a1;
  s_c[0][threadIdx.x] = c0;
  s_c[1][threadIdx.x] = c1;
  __syncthreads();
  for (int p2 = 128; p2 > 0; p2 >>= 1) {
    if ((int)threadIdx.x < p2) {
      s_ll[0][threadIdx.x] += s_ll[0][threadIdx.x + p2];
      s_ll[1][threadIdx.x] += s_ll[1][threadIdx.x + p2];
      s_c[0][threadIdx.x] += s_c[0][threadIdx.x + p2];
      s_c[1][threadIdx.x] += s_c[1][threadIdx.x + p2];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out->link_ll = s_ll[0][0];
    out->nonlink_ll = s_ll[1][0];
    out->link_cnt = s_c[0][0];
    out->nonlink_cnt = s_c[1][0];
  }
}

template <int L, int KPT>
int launch_ppx(ammsb_ctx* ctx, const PpxArgs& a, hipStream_t s) {
  using Grp = Group<L>;
  const uint32_t blocks = (a.P + Grp::PER_BLOCK - 1) / Grp::PER_BLOCK;
  static const std::string name = ammsb_kname("ppx_kernel<%d, %d>", L, KPT);
  ctx->kernel_name[AMMSB_KN_PPX] = name.c_str();
  ppx_kernel<L, KPT><<<blocks, Grp::BLOCK, 0, s>>>(a);
  AMMSB_LAUNCH_CHECK(ctx);
  return AMMSB_OK;
}

inline int pick_kpt(uint64_t K, uint32_t L) {
  const uint64_t need = (K + L - 1) / L;
  for (int c : {1, 2, 4, 8, 16, 32})
    if ((uint64_t)c >= need) return c;
  return 0;
}

}  // namespace

#define AMMSB_DISPATCH_KPT(kpt, ...)                                  \
  switch (kpt) {                                                      \
    case 1: { constexpr int KPT_ = 1; __VA_ARGS__; } break;           \
    case 2: { constexpr int KPT_ = 2; __VA_ARGS__; } break;           \
    case 4: { constexpr int KPT_ = 4; __VA_ARGS__; } break;           \
    case 8: { constexpr int KPT_ = 8; __VA_ARGS__; } break;           \
    case 16: { constexpr int KPT_ = 16; __VA_ARGS__; } break;         \
    case 32: { constexpr int KPT_ = 32; __VA_ARGS__; } break;         \
    default: return AMMSB_ERANGE;                                     \
  }

#define AMMSB_DISPATCH_HOT_L(wg, ...)                                 \
  switch (wg) {                                                       \
    case 16: { constexpr int L_ = 16; __VA_ARGS__; } break;           \
    case 32: { constexpr int L_ = 32; __VA_ARGS__; } break;           \
    case 64: { constexpr int L_ = 64; __VA_ARGS__; } break;           \
    case 128: { constexpr int L_ = 128; __VA_ARGS__; } break;         \
    case 256: { constexpr int L_ = 256; __VA_ARGS__; } break;         \
    case 512: { constexpr int L_ = 512; __VA_ARGS__; } break;         \
    case 1024: { constexpr int L_ = 1024; __VA_ARGS__; } break;       \
    default: return AMMSB_EINVAL;                                     \
  }

extern "C" int ammsb_perplexity(ammsb_ctx* ctx, const float* beta, const ammsb_rpm* pi, const ammsb_set* heldout_set,
                                const uint64_t* edges, uint32_t n_edges, uint32_t edge_begin, uint32_t edge_end,
                                uint32_t call_count, uint32_t wg, float* ppx_per_edge, ammsb_ppx_sums* out,
                                void* stream) {
  AMMSB_CHECK_ARG(ctx, ctx && beta && pi && heldout_set && edges && ppx_per_edge && out, "null argument");
  AMMSB_CHECK_ARG(ctx, pi->num_blocks >= 1 && pi->num_blocks <= AMMSB_RPM_MAX_BLOCKS && pi->rows_in_block > 0,
                  "bad pi descriptor");
  AMMSB_CHECK_ARG(ctx, pi->num_cols == ctx->params.K, "pi cols != K");
  AMMSB_CHECK_ARG(ctx, heldout_set->slots && heldout_set->num_bins > 0 && heldout_set->prime_idx < 4,
                  "bad set descriptor");
  AMMSB_CHECK_ARG(ctx, call_count >= 1, "call_count is 1-based");
  AMMSB_CHECK_ARG(ctx, is_pow2(wg) && wg >= 16 && wg <= 1024, "ppx wg must be a power of two in [16, 1024]");
  if (edge_end > n_edges) edge_end = n_edges;
  hipStream_t s = as_stream(stream);
  if (edge_begin >= edge_end) {
    AMMSB_HIP(ctx, hipMemsetAsync(out, 0, sizeof(ammsb_ppx_sums), s));
    return AMMSB_OK;
  }
  const uint32_t K = (uint32_t)ctx->params.K;
  const int kpt = pick_kpt(K, wg);
  // AMMSB_PPX_FORM=g: the generic kernel wherever it fits (tests compare it with the specialised ones)
  static const bool force_gen = [] {
    const char* f = getenv("AMMSB_PPX_FORM");
    return f && f[0] == 'g';
  }();
  const bool generic = kpt == 0 || (force_gen && gen_threads(K, wg) != 0);
  if (generic && gen_threads(K, wg) == 0) {
    snprintf(ctx->err, sizeof ctx->err, "ammsb_perplexity: K=%u at wg=%u: more than 32 columns per work-item needs K <= %u",
             K, wg, kGenMaxK);
    return AMMSB_ERANGE;
  }
  PpxArgs a;
  a.beta = beta;
  a.pi = *pi;
  a.set = dev_set(*heldout_set);
  a.edges = edges;
  a.ppx_per_edge = ppx_per_edge;
  a.ll_partials = ctx->ppx_partials;
  a.cnt_partials = ctx->ppx_cnt_partials;
  a.edge_begin = edge_begin;
  a.edge_end = edge_end;
  a.K = K;
  a.call_count = call_count;
  a.epsilon = ctx->params.epsilon;
  a.ticket = nullptr;
  a.out = out;
  const uint32_t span = edge_end - edge_begin;
  uint32_t want = (uint32_t)ctx->num_cus * 8u * 64u / (wg < 64 ? 64u : wg) * (wg < 64 ? 64u / wg : 1u);
  // (the one-wave-per-slot LDS form at wg 32 wants the slot count of the wg 64 form: 8 resident waves per CU)
  if (wg == 32 && (K == 256 || K == 512 || K == 1024) && pi->num_cols % 4 == 0) want = (uint32_t)ctx->num_cus * 8u;
  if (want < 64) want = 64;
  if (want > ctx->max_ppx_blocks) want = ctx->max_ppx_blocks;
  a.P = span < want ? span : want;
  static const bool force_reg = [] {
    const char* f = getenv("AMMSB_PPX_FORM");
    return f && f[0] == 'r';
  }();
  if (wg == 32 && !force_reg && !force_gen && pi->num_cols % 4 == 0 && (K == 256 || K == 512 || K == 1024)) {
    // the reference's default work-group size on the LDS-streamed one-wave-per-slot kernel (VLane<32>)
    int rc = AMMSB_OK;
    if (K == 256) rc = launch_ppx_lds<4, 32>(ctx, a, s);
    else if (K == 512) rc = launch_ppx_lds<8, 32>(ctx, a, s);
    else rc = launch_ppx_lds<16, 32>(ctx, a, s);
    if (rc == 1) return AMMSB_OK;  // (the launch reduced its own partials)
    if (rc) return rc;
  } else if (generic) {
    const int rc = launch_ppx_gen(ctx, a, wg, s);
    if (rc) return rc;
  } else if (wg == 64 && !force_reg && K == 64u * (uint32_t)kpt && kpt >= 4 && kpt <= 16 && pi->num_cols % 4 == 0) {
    int rc = AMMSB_OK;
    switch (kpt) {
      case 4: rc = launch_ppx_lds<4>(ctx, a, s); break;
      case 8: rc = launch_ppx_lds<8>(ctx, a, s); break;
      default: rc = launch_ppx_lds<16>(ctx, a, s); break;
    }
    if (rc == 1) return AMMSB_OK;  // (the launch reduced its own partials)
    if (rc) return rc;
  } else {
    AMMSB_DISPATCH_HOT_L(wg, AMMSB_DISPATCH_KPT(kpt, {
                           int rc = launch_ppx<L_, KPT_>(ctx, a, s);
                           if (rc) return rc;
                         }));
  }
  ppx_reduce_kernel<<<1, 256, 0, s>>>(a.ll_partials, a.cnt_partials, a.P, out);
  AMMSB_LAUNCH_CHECK(ctx);
  return AMMSB_OK;
}
