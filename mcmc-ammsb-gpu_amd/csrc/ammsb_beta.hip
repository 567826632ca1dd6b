// beta / theta pipeline for gfx950.
//
// Replaces BetaUpdater::operator() (mcmc/beta.cc:334-384): sum_theta (:30-37),
// calculate_grads_partial work-group variant (:174-233), sum_grads (:39-49), update_theta (:51-82)
// and the theta->beta copy + pair normalisation (:376-383, normalize.cc:13-32).
//
// The reference gives every edge its own work-group, keeps a 2K-float accumulator per group in LDS,
// spills one partial row per group to HBM (as many bytes as the pi rows it read) and then sums the
// rows with 2K serial threads.  Here a virtual group keeps the 2K accumulators of the columns its
// lanes own in registers across all the edges it walks, so only P <= 2048 partial rows leave the
// chip; a second small kernel reduces them in a fixed order.  The per-edge arithmetic (two WG_SUMs,
// probs, gradient terms) is the reference's, lane for lane; the order in which edges are added is
// not (see include/ammsb.h, ammsb_beta_grads).
#pragma clang fp contract(off)

#include <stdlib.h>

#include <type_traits>

#include "ammsb_ctx.h"
#include "ammsb_dev.h"
#include "ammsb_step.h"

using namespace ammsb;

#ifdef AMMSB_BETA_TRACE
// development aid (tools/beta_trace.sh): shader-clock stamps of slot 0 of beta_grads_lds_kernel, read back with
// ammsb_debug_trace_beta
__device__ unsigned long long g_beta_trace[256];
#define BETA_TRACE(slot) do { if (blockIdx.x == 0 && threadIdx.x == 0 && (slot) < 256) g_beta_trace[(slot)] = __builtin_readcyclecounter(); } while (0)
extern "C" int ammsb_debug_trace_beta(unsigned long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_beta_trace), sizeof(unsigned long long) * (n < 256 ? n : 256)) == hipSuccess ? 0 : -2;
}
// per-block record: [block][0..1] shader clock at start / end, [2..3] 100 MHz wall clock at start / end, [4] HW_ID | XCC_ID << 32
__device__ unsigned long long g_beta_blk[4096 * 5];
#define BETA_BLK(end) do { if (threadIdx.x == 0 && blockIdx.x < 4096) { \
    g_beta_blk[blockIdx.x * 5 + (end)] = __builtin_readcyclecounter(); \
    g_beta_blk[blockIdx.x * 5 + 2 + (end)] = __builtin_amdgcn_s_memrealtime(); \
    if (!(end)) g_beta_blk[blockIdx.x * 5 + 4] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | \
                                                ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32); } } while (0)
extern "C" int ammsb_debug_blocks_beta(unsigned long long* out, int n_blocks) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_beta_blk), sizeof(unsigned long long) * 5 * (n_blocks < 4096 ? n_blocks : 4096)) == hipSuccess ? 0 : -2;
}
#else
#define BETA_TRACE(slot) do { } while (0)
#define BETA_BLK(end) do { } while (0)
#endif

namespace {

struct BetaArgs {
  const float* theta;
  const float* beta;
  ammsb_rpm pi;
  DevSet set;  // (ammsb_dev.h: the descriptor + the modulo magic)
  const uint64_t* edges;
  float* partials;   // [P, 2K]
  const float4* coef;  // [K] per-column constants of CALC_GRADS (theta_coef below): {beta_k, d0n, d1l, noo}
  uint32_t edge_begin, edge_end, P, K;
  float epsilon;
  const ammsb_step_desc* desc;  // non-null (captured graph): edges [0, desc->n_edges), P = min(n_edges, P)
  ammsb_pi_fusion fuse;         // update_pi folded in (LDS kernels at wg 64, with a descriptor): see ammsb_step.h
  uint32_t pi_nt;               // (fused LDS kernels) the pi rows are stored with the non-temporal hint
  unsigned long long* stamps;   // optional (with desc): block 0 notes the device time at which it starts
};

// block 0 notes when the gradient kernel starts; with update_pi folded in that is also when "update_pi" starts
template <bool FUSE>
__device__ __forceinline__ void beta_stamp(const BetaArgs& a) {
  if constexpr (FUSE) note_stamp(a.stamps, a.desc, AMMSB_STAMP_PI);
  note_stamp(a.stamps, a.desc, AMMSB_STAMP_GRADS);
}

struct BetaStep {
  uint32_t edge_begin, edge_end, P;
};
__device__ __forceinline__ BetaStep beta_step(const BetaArgs& a) {
  BetaStep st = {a.edge_begin, a.edge_end, a.P};
  if (a.desc) {
    st.edge_begin = 0;
    st.edge_end = a.desc->n_edges;
    st.P = st.edge_end < a.P ? st.edge_end : a.P;
  }
  return st;
}

// The per-column constants of a gradient launch (beta.cc:163-168 with sum_theta, beta.cc:30-37, folded in):
//   ts = theta_k0 + theta_k1 (= theta_sum[k]);  d0n = 1 / theta_k0 - 1 / ts  (y = 0: (1 - y) / Theta0 - 1 / theta_sum),
//   d1l = 1 / theta_k1 - 1 / ts  (y = 1: y / Theta1 - 1 / theta_sum),  noo = 0 - 1 / ts  (the other component).
// They depend on theta alone, so they are computed ONCE per theta -- by theta_coef_kernel in front of an eager gradient
// launch, by the theta step itself inside the descriptor loop -- into a [K] float4 table the gradient kernels load,
// instead of by every one of a launch's 2 048 waves (three IEEE divisions per column: 14 000 of a wave's 131 500 cycles
// at K = 1024, in-kernel stamps of round 3).  An IEEE division gives the same quotient wherever it is evaluated, so the
// table holds exactly what the kernels used to compute for themselves.
__device__ __forceinline__ void theta_coef(uint32_t k, float t0, float t1, float beta_k, float4* coef, float* theta_sum) {
  const float ts = t0 + t1;  // sum_theta, beta.cc:30-37
  theta_sum[k] = ts;
  const float oo = 1.0f / ts;
  coef[k] = make_float4(beta_k, 1.0f / t0 - oo, 1.0f / t1 - oo, 0.0f - oo);
}

__global__ __launch_bounds__(256) void theta_coef_kernel(const float* theta, const float* beta, float4* coef,
                                                         float* theta_sum, uint32_t K) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < K) theta_coef(k, theta[2 * k], theta[2 * k + 1], beta[2 * k + 1], coef, theta_sum);
}

// FUSE: update_pi folded in, as in beta_grads_lds_kernel<KPT, 1, true> below (which has the description): the rows come
// from phi_vec, are normalised with update_pi_kernel<L, KPT>'s arithmetic as they are consumed, and are stored to pi.
template <int L, int KPT, bool FUSE = false, bool ONE = false>
__global__ __launch_bounds__(Group<L>::BLOCK) void beta_grads_kernel(const BetaArgs a) {
  if constexpr (ONE) __builtin_assume(a.pi.num_blocks == 1);  // (pi is one block: see update_phi_lds2_kernel, ammsb_phi.hip)
  using Grp = Group<L>;
  __shared__ float aux[Grp::AUX];
  const int l = Grp::lane();
  const BetaStep st = beta_step(a);
  if (st.P == 0) return;  // (uniform) a skipped step
  beta_stamp<FUSE>(a);
  const uint32_t gs = blockIdx.x * Grp::PER_BLOCK + Grp::slot();  // partial-row slot
  const bool live = gs < st.P;
  const uint32_t K = a.K;
  const float EPS = a.epsilon;

  // per-lane constants for the columns this lane owns
  float bk[KPT], omb[KPT], d0n[KPT], d1l[KPT], noo[KPT];
#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    const uint32_t k = l + j * L;
    if (k < K) {
      const float4 c = a.coef[k];  // (theta_coef above)
      bk[j] = c.x;
      omb[j] = 1.0f - bk[j];
      d0n[j] = c.y;  // y = 0: (1 - y) / Theta0 - 1/theta_sum
      d1l[j] = c.z;  // y = 1:  y / Theta1      - 1/theta_sum
      noo[j] = c.w;  // the other component: 0 / Theta - 1/theta_sum
    } else {
      bk[j] = omb[j] = d0n[j] = d1l[j] = noo[j] = 0.0f;
    }
  }

  float acc0[KPT], acc1[KPT];
#pragma unroll
  for (int j = 0; j < KPT; ++j) acc0[j] = acc1[j] = 0.0f;

  const uint32_t n_edges = st.edge_end - st.edge_begin;
  const uint32_t trips = (n_edges + st.P - 1) / st.P;  // uniform
  int phase = 0;

  float pa_s[FUSE ? KPT : 1];
  uint32_t shared_node = 0;
  if constexpr (FUSE) {  // the shared end point (node 0): normalised by every slot, stored by slot 0
    shared_node = a.fuse.nodes[0];
    float partial = 0.0f;
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      const uint32_t k = l + j * L;
      const float x = a.fuse.phi_vec[k < K ? k : K - 1];
      pa_s[j] = k < K ? x : 0.0f;
      partial += pa_s[j];
    }
    const float sum = Grp::sum(partial, aux, phase);
    float* dst = rpm_row(a.pi, shared_node);
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      const uint32_t k = l + j * L;
      pa_s[j] = pa_s[j] / sum;
      if (gs == 0 && k < K) dst[k] = pa_s[j];
    }
    if (gs == 0 && l == 0) a.fuse.phi_sum[shared_node] = sum;
  }

  // Software pipeline.  A slot walks edges e(t) = edge_begin + gs + t * P.  Two dependent latencies sit
  // in front of every edge (its key, then its two rows); both are taken off the critical path:
  //   keys: lane i of the group loads the key of trip tb + i AND probes the cuckoo set for it, one
  //         batch of W = min(L, 64) trips at a time and one whole batch ahead; key and link bit are
  //         handed out with a cross-lane read / a ballot mask.  (The probe is two 64-bit modulos and
  //         two dependent 32-byte reads: done per edge by every lane it dominated the kernel.)
  //   rows: requested two trips ahead into a three-deep register ring.
  constexpr int W = L < 64 ? L : 64;
  const int kl = threadIdx.x & (W - 1);  // lane within the key batch
  const int wave_lane = threadIdx.x & 63;
  auto load_keys = [&](uint32_t tb, unsigned long long* ymask) -> unsigned long long {
    const uint64_t e = (uint64_t)st.edge_begin + gs + (uint64_t)(tb + kl) * st.P;
    const bool ok = live && (tb + kl) < trips && e < st.edge_end;
    const unsigned long long edge = a.edges[ok ? e : st.edge_begin];  // an exhausted trip shadows the first edge
    const uint32_t u = (uint32_t)(edge >> 32), v = (uint32_t)(edge & 0xffffffffu);
    *ymask = __ballot(set_has(a.set, make_edge(u, v)));  // bit = lane of the wave
    return edge;
  };
  uint32_t tb = 0;
  unsigned long long ym = 0, ym_next = 0;
  unsigned long long kv = load_keys(0, &ym), kv_next = load_keys(W, &ym_next);

  float pa[3][KPT], pb[3][KPT];
  bool link[3] = {false, false, false};
  bool have[3] = {false, false, false};
  uint32_t partner[3] = {0, 0, 0};
  auto fetch = [&](int b, uint32_t t) {
    const uint64_t e_raw = (uint64_t)st.edge_begin + gs + (uint64_t)t * st.P;
    have[b] = live && t < trips && e_raw < st.edge_end;
    const uint32_t rel = t - tb;  // 0 .. 2W-1 by construction
    const bool first = rel < (uint32_t)W;
    const uint32_t src = first ? rel : rel - W;                 // lane of the group's key batch
    const unsigned long long edge = first ? __shfl(kv, (int)src, W) : __shfl(kv_next, (int)src, W);
    const uint32_t bit = (wave_lane & ~(W - 1)) + src;          // that lane's position in the wave
    link[b] = (((first ? ym : ym_next) >> bit) & 1ull) != 0;
    const uint32_t u = (uint32_t)(edge >> 32), v = (uint32_t)(edge & 0xffffffffu);
    if constexpr (FUSE) {
      partner[b] = u == shared_node ? v : u;
      const float* rb = a.fuse.phi_vec + ((have[b] ? e_raw : (uint64_t)st.edge_begin) + 1) * K;  // node e + 1
#pragma unroll
      for (int j = 0; j < KPT; ++j) {
        const uint32_t k = l + j * L;
        const float xb = rb[k < K ? k : K - 1];
        pa[b][j] = pa_s[j];
        pb[b][j] = k < K ? xb : 0.0f;
      }
    } else {
      const float* ra = rpm_row(a.pi, u);
      const float* rb = rpm_row(a.pi, v);
#pragma unroll
      for (int j = 0; j < KPT; ++j) {
        const uint32_t k = l + j * L;
        const uint32_t ck = k < K ? k : K - 1;  // unconditional loads: a column beyond K shadows column K-1
        const float xa = ra[ck], xb = rb[ck];
        pa[b][j] = k < K ? xa : 0.0f;
        pb[b][j] = xb;
      }
    }
  };
  auto consume = [&](int b) {
    const bool y = link[b];
    if constexpr (FUSE) {  // update_pi of the partner: normalise its phi_vec row, store it (once: this is its edge)
      float partial = 0.0f;
#pragma unroll
      for (int j = 0; j < KPT; ++j) partial += pb[b][j];
      const float sum = Grp::sum(partial, aux, phase);
      float* dst = rpm_row(a.pi, partner[b]);
#pragma unroll
      for (int j = 0; j < KPT; ++j) {
        const uint32_t k = l + j * L;
        pb[b][j] = pb[b][j] / sum;
        if (have[b] && k < K) dst[k] = pb[b][j];
      }
      if (have[b] && l == 0) a.fuse.phi_sum[partner[b]] = sum;
    }
    float scratch = 0.0f, ppart = 0.0f, lo = 1.0f;
    float probs[KPT];
#pragma unroll
    for (int j = 0; j < KPT; ++j) {  // CALC_PROBS, beta.cc:145-160
      const float f = pa[b][j] * pb[b][j];
      scratch += f;
      probs[j] = y ? bk[j] * f : omb[j] * f;
      ppart += probs[j];
      const float m = fabsf(probs[j]);
      lo = fminf(lo, m == 0.0f ? 1.0f : m);  // an exact zero divides exactly on either path
    }
    const float pi_sum = Grp::sum(scratch, aux, phase);  // beta.cc:209-213
    float probs_sum = Grp::sum(ppart, aux, phase);       // beta.cc:214-217
    const float w = y ? EPS : (1.0f - EPS);
    const float prob_0 = w * (1.0f - pi_sum);
    probs_sum += prob_0;
    if (have[b]) {
      // CALC_GRADS, beta.cc:161-171.  probs[j] / probs_sum with the reciprocal refined once per edge
      // (ammsb_dev.h "exact division"): exact when every numerator is 0 or at least 2^-100 in
      // magnitude and the divisor is in range; otherwise the plain IEEE divide.
      if (lo >= kProbsLo && in_range(probs_sum, kPsumLo, kPsumHi)) {
        const float rps = refined_rcp(probs_sum);
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
          const float f = div_with_rcp(probs[j], probs_sum, rps);
          acc0[j] += f * (y ? noo[j] : d0n[j]);
          acc1[j] += f * (y ? d1l[j] : noo[j]);
        }
      } else {
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
          const float f = probs[j] / probs_sum;
          acc0[j] += f * (y ? noo[j] : d0n[j]);
          acc1[j] += f * (y ? d1l[j] : noo[j]);
        }
      }
    }
  };

  fetch(0, 0);
  fetch(1, 1);
  for (uint32_t t = 0; t < trips; t += 3) {
    // trips t+2 .. t+4 stay inside [tb, tb + 2W): advance the key window when t leaves the first batch
    if (t + 2 >= tb + 2 * W - 3) {
      kv = kv_next;
      ym = ym_next;
      tb += W;
      kv_next = load_keys(tb + W, &ym_next);
    }
    fetch(2, t + 2);
    consume(0);
    fetch(0, t + 3);
    consume(1);
    fetch(1, t + 4);
    consume(2);
  }

  if (live) {
    float* out = a.partials + (uint64_t)gs * 2 * K;
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      const uint32_t k = l + j * L;
      if (k < K) *reinterpret_cast<float2*>(out + 2 * k) = make_float2(acc0[j], acc1[j]);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// LDS-streamed form for L = 64 and K = 64 * KPT (KPT in {4, 8, 16}), the shape of ammsb_phi.hip's
// update_phi_lds_kernel: one wave per slot, so both WG_SUMs are pure cross-lane reductions (no LDS
// exchange, no barrier), and the pi rows of the next three edges arrive by LDS-DMA in a four-slot ring while
// the current edge is being reduced.  probs[] overwrites the row in place.  Same arithmetic and
// the same operation order as the register kernel at L = 64.

typedef __attribute__((address_space(3))) void beta_lds_void_t;
typedef const __attribute__((address_space(1))) void beta_glb_void_t;

// FUSE (W == 1, descriptor form, node-stratified mini-batch: edge e = (nodes[0], nodes[e + 1]) in either order): the
// kernel is also update_pi (phi.cc:177-197).  The ring is filled from phi_vec rows instead of pi rows; a row is
// normalised as it is consumed -- the lane's columns added in ascending order, WG_SUM over the 64 lanes, one IEEE
// division per column: update_pi_kernel<64, KPT>'s arithmetic -- and written to pi (+ phi_sum), so every pi row of
// the mini-batch is written exactly once (the shared node's by slot 0) and what the gradient multiplies are the
// values the separate update_pi would have stored: bit-identical, one launch and one pass over the rows less.
// VL = 32 (W == 1): the reference work-group size is 32 (its default, main.cc:64); the slot keeps its whole wave, the
// WG_SUM chains / trees follow the 32 virtual lanes (VLane<32>, ammsb_dev.h).
#ifndef AMMSB_BETA_VGPRS
#define AMMSB_BETA_VGPRS 256
#endif
template <int KPT, int W, bool FUSE, int VL>
__device__ __forceinline__ void beta_grads_lds_body(const BetaArgs& a) {
  static_assert(!FUSE || W == 1, "the fused form is one wave per slot");
  static_assert(VL == 64 || W == 1, "virtual half-wave lanes only for one-wave slots");
  using VLn = VLane<VL>;
  // L = 64 W lanes per slot: wave wv owns columns 64 wv + ln + L j (the slicing of update_phi_lds_kernel); every
  // wave streams its own slice of the rows, the two WG_SUMs of an edge share one LDS exchange and one barrier.
  constexpr int L = 64 * W;
  constexpr int KW = 64 * KPT;
  constexpr int K = L * KPT, HP = KPT / 2, PIECES = KPT / 4;
  constexpr uint32_t D = 4;  // ring depth: edge t is reduced while the rows of t+1 .. t+3 are in flight
  constexpr int ST = 2 * HP + 1;  // (fused form) store instructions of one trip: KPT columns of the pi row + phi_sum
  extern __shared__ __align__(16) char smem[];  // per wave [D][KW] floats: row slice of the second end point, then probs
  __shared__ float xsum[W > 1 ? 4 * L : 1];     // two partials per lane, double buffered
  const int tid = threadIdx.x, wv = W == 1 ? 0 : tid >> 6, ln = W == 1 ? tid : tid & 63;
  char* wave_smem = smem + wv * (D * KW * sizeof(float));
  float* ring = reinterpret_cast<float*>(wave_smem);
  const BetaStep st = beta_step(a);
  beta_stamp<FUSE>(a);
  const uint32_t gs = blockIdx.x;  // partial-row slot; the grid is exactly P blocks (at least P with a descriptor)
  if (gs >= st.P) return;          // block-uniform
  const float EPS = a.epsilon;
  BETA_TRACE(0);
  BETA_BLK(0);

  f32x2 bk[HP], d0n[HP], d1l[HP], noo[HP];
  auto load_coef = [&]() {  // the lane's per-column constants (theta_coef above): one 16-byte load per column
#pragma unroll
    for (int p = 0; p < HP; ++p) {
      const float4 c0 = a.coef[tid + 2 * L * p], c1 = a.coef[tid + 2 * L * p + L];
      bk[p] = f32x2{c0.x, c1.x};
      d0n[p] = f32x2{c0.y, c1.y};
      d1l[p] = f32x2{c0.z, c1.z};
      noo[p] = f32x2{c0.w, c1.w};
    }
  };
  if constexpr (FUSE || W > 1) load_coef();  // (the plain one-wave form requests its first rows first, below)
  f32x2 acc0[HP], acc1[HP];
#pragma unroll
  for (int p = 0; p < HP; ++p) acc0[p] = acc1[p] = f32x2{0.0f, 0.0f};

  // edges of this slot: e(t) = edge_begin + gs + t * P, t < trips
  const uint32_t n_edges = st.edge_end - st.edge_begin;
  const uint32_t trips = gs < n_edges ? (n_edges - gs + st.P - 1) / st.P : 0;  // block-uniform
  int phase = 0;

  // keys and link bits: lane i of every wave holds trip tb + i (and tb + 64 + i in the second window), probed one
  // whole window ahead -- see the register kernel
  auto load_key = [&](uint32_t tb) -> unsigned long long {  // the key of trip tb + ln
    const bool ok = tb + ln < trips;
    const uint64_t e = (uint64_t)st.edge_begin + gs + (uint64_t)(tb + ln) * st.P;
    return a.edges[ok ? e : st.edge_begin];
  };
  auto probe_key = [&](unsigned long long edge) -> unsigned long long {  // link bits of a window (one per lane)
    const uint32_t u = (uint32_t)(edge >> 32), v = (uint32_t)(edge & 0xffffffffu);
    return __ballot(set_has(a.set, make_edge(u, v)));
  };
  auto load_keys = [&](uint32_t tb, unsigned long long* ymask) -> unsigned long long {
    const unsigned long long edge = load_key(tb);
    *ymask = probe_key(edge);
    return edge;
  };
  uint32_t tb = 0;
  unsigned long long ym = 0, ym_next = 0;
  BETA_TRACE(1);
  unsigned long long kv = load_key(0), kv_next = 0;
  // The plain one-wave form puts NOTHING between a slot's first keys and its first row requests: the D rows the ring
  // holds are asked for as soon as the keys are there (the second end points come out of the key window by readlane),
  // then the per-column constants and the first end point's row, and only then the edge-set probes -- whose round trip
  // the rows' flight now covers.  (Round 3 had three IEEE divisions per column and both windows' probes in front of the
  // first request: 20 000 of a wave's 131 500 cycles.)
  f32x2 pa[HP];
  uint32_t cur_u = 0xffffffffu;
#pragma unroll
  for (int p = 0; p < HP; ++p) pa[p] = f32x2{0.0f, 0.0f};
  uint32_t issued = 0;  // (plain one-wave form) rows requested so far
  if constexpr (!FUSE && W == 1) {
#pragma unroll
    for (uint32_t k = 0; k < D; ++k) {
      if (k < trips) {
        const uint32_t v = (uint32_t)(wave_lane_u64(kv, k) & 0xffffffffu);
        const float* rb = rpm_row(a.pi, v) + 4 * tid;
        char* dst = wave_smem + k * (KW * sizeof(float));
#pragma unroll
        for (int p = 0; p < PIECES; ++p)
          __builtin_amdgcn_global_load_lds((beta_glb_void_t*)(rb + 4 * L * p), (beta_lds_void_t*)(dst + 1024 * p), 16, 0, 0);
        ++issued;
      }
    }
    load_coef();
    if (trips > 0) {  // the first end point of the slot's first edge (shared by every edge of a node-strategy mini-batch)
      cur_u = (uint32_t)(wave_lane_u64(kv, 0) >> 32);
      const float* ra = rpm_row(a.pi, cur_u);
#pragma unroll
      for (int p = 0; p < HP; ++p) pa[p] = f32x2{ra[tid + 2 * L * p], ra[tid + 2 * L * p + L]};
    }
  }
  ym = probe_key(kv);
  if (trips > 64u) kv_next = load_keys(64, &ym_next);  // (block-uniform; a slot of at most 64 edges never looks there)
  if constexpr (!FUSE && W == 1) {
    // everything the prologue loaded is waited for HERE, not at the loop's first use (a load left pending across the
    // loop's back edge makes hipcc's wait-count pass drain the row ring on every trip, see below)
#pragma unroll
    for (int p = 0; p < HP; ++p)
      asm volatile("" : "+v"(pa[p].x), "+v"(pa[p].y), "+v"(bk[p].x), "+v"(bk[p].y), "+v"(d0n[p].x), "+v"(d0n[p].y), "+v"(d1l[p].x),
                   "+v"(d1l[p].y), "+v"(noo[p].x), "+v"(noo[p].y));
  }
  BETA_TRACE(2);

  auto key_of = [&](uint32_t t, bool* y) -> unsigned long long {
    const uint32_t rel = t - tb;  // 0 .. 127 by construction
    const bool first = rel < 64u;
    const uint32_t src = first ? rel : rel - 64u;
    const unsigned long long edge = wave_lane_u64(first ? kv : kv_next, src);  // (t, tb are wave-uniform)
    *y = (((first ? ym : ym_next) >> src) & 1ull) != 0;
    return edge;
  };
  // the row of the second end point goes through the ring; the first end point is shared by every edge of
  // a mini-batch of the node strategies (sample.cc:249-303), so its row stays in registers until it changes
  auto request = [&](uint32_t t) {
    bool y;
    const unsigned long long edge = key_of(t, &y);
    const uint32_t v = __builtin_amdgcn_readfirstlane((uint32_t)(edge & 0xffffffffu));
    const float* rb;
    if constexpr (FUSE) rb = a.fuse.phi_vec + ((uint64_t)gs + (uint64_t)t * st.P + 1) * K + 4 * tid;  // node e + 1
    else rb = rpm_row(a.pi, v) + (W == 1 ? 4 * tid : L * (ln >> 4) + 64 * wv + 4 * (ln & 15));
    char* dst = wave_smem + (t % D) * (KW * sizeof(float));
#pragma unroll
    for (int p = 0; p < PIECES; ++p)
      __builtin_amdgcn_global_load_lds((beta_glb_void_t*)(rb + 4 * L * p), (beta_lds_void_t*)(dst + 1024 * p), 16, 0, 0);
  };

  // WG_SUM over L lanes (sum.cc:20-29) of two values at once
  auto group_sum2 = [&](float& v0, float& v1) {
    if constexpr (W == 1) {
      v0 = VLn::tree(v0);
      v1 = VLn::tree(v1);
    } else {
      float* x = xsum + phase * 2 * L;
      phase ^= 1;
      x[tid] = v0;
      x[L + tid] = v1;
      __syncthreads();
      float p0[W], p1[W];
#pragma unroll
      for (int i = 0; i < W; ++i) {
        p0[i] = x[64 * i + ln];
        p1[i] = x[L + 64 * i + ln];
      }
#pragma unroll
      for (int st = W / 2; st >= 1; st >>= 1) {
#pragma unroll
        for (int i = 0; i < st; ++i) {
          p0[i] += p0[i + st];
          p1[i] += p1[i + st];
        }
      }
      v0 = Group<64>::wave_tree64(p0[0]);
      v1 = Group<64>::wave_tree64(p1[0]);
    }
  };

  uint32_t shared_node = 0;
  if constexpr (FUSE) {  // the shared end point: node 0, normalised here by every slot, stored by slot 0
    shared_node = a.fuse.nodes[0];
    const float* src = a.fuse.phi_vec;
    float partial = 0.0f;
#pragma unroll
    for (int p = 0; p < HP; ++p) {
      pa[p] = f32x2{src[tid + 2 * L * p], src[tid + 2 * L * p + L]};
      VLn::chain(partial, pa[p].x);
      VLn::chain(partial, pa[p].y);
    }
    const float sum = VLn::tree(partial);
    float* dst = rpm_row(a.pi, shared_node);
#pragma unroll
    for (int p = 0; p < HP; ++p) {
      pa[p] = f32x2{pa[p].x / sum, pa[p].y / sum};
      if (gs == 0) {
        dst[tid + 2 * L * p] = pa[p].x;
        dst[tid + 2 * L * p + L] = pa[p].y;
      }
    }
    if (gs == 0 && tid == 0) a.fuse.phi_sum[shared_node] = sum;
  }

  // the link flag's picks among the per-column constants; mode 1 / 2 = the flag is known to be false / true (uniform
  // mini-batches: no select is emitted), mode 0 = decided at run time
  // (`one`: 1.0f behind an empty asm, refreshed per step by the callers -- written as a literal, 1 - beta_k is loop
  // invariant and hipcc keeps all KPT of them in registers across the loop: 16 more VGPRs at K = 1024, which drops
  // the kernel from two waves per SIMD to one)
  float one = 1.0f;
  auto sel_b = [&](int p, bool y, auto mode) -> f32x2 {
    if constexpr (decltype(mode)::value == 1) return one - bk[p];
    else if constexpr (decltype(mode)::value == 2) return bk[p];
    else return y ? bk[p] : one - bk[p];
  };
  auto sel_0 = [&](int p, bool y, auto mode) -> f32x2 {
    if constexpr (decltype(mode)::value == 1) return d0n[p];
    else if constexpr (decltype(mode)::value == 2) return noo[p];
    else return y ? noo[p] : d0n[p];
  };
  auto sel_1 = [&](int p, bool y, auto mode) -> f32x2 {
    if constexpr (decltype(mode)::value == 1) return noo[p];
    else if constexpr (decltype(mode)::value == 2) return d1l[p];
    else return y ? d1l[p] : noo[p];
  };
  auto sel_w = [&](bool y, auto mode) -> float {
    if constexpr (decltype(mode)::value == 1) return 1.0f - EPS;
    else if constexpr (decltype(mode)::value == 2) return EPS;
    else return y ? EPS : (1.0f - EPS);
  };
  if constexpr (!FUSE && W == 1) {
    // TWO edges per step where the slot's consecutive edges share their first end point (every edge of a node-strategy
    // mini-batch does): the chains of edges t and t + 1 -- probs, two WG_SUMs each, reciprocal, division -- are issued
    // together (a one-edge trip is a dependent chain of ~3 900 cycles for 301 vector instructions, in-kernel stamps),
    // their four sums advance as one transposed four-row chain / tree (VLane::chain_rows<4> / tree_rows<4>), and the
    // accumulators take edge t before edge t + 1: bit-identical to the one-edge loop.  The ring (D = 4 rows) is kept
    // full: rows t .. t + 3 are in it or on their way whenever rows < t have been consumed.
    auto fill = [&](uint32_t t) {
#pragma unroll
      for (uint32_t k = 0; k < D; ++k) {
        if (issued < trips && issued < t + D) {
          if (issued >= tb + 128) {  // the look-ahead leaves the two key windows: slide them (t >= tb + 64 here)
            kv = kv_next;
            ym = ym_next;
            tb += 64;
            kv_next = load_keys(tb + 64, &ym_next);
          }
          request(issued);
          ++issued;
        }
      }
    };
    auto wait_rows = [&](uint32_t behind) {  // rows up to the ones consumed now have landed; `behind` rows may still fly
      if (behind >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PIECES) : "memory");
      else if (behind == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PIECES) : "memory");
      else if (behind == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 * PIECES) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    uint32_t t = 0;
    while (t < trips) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slots of rows < t have been read for the last time
      BETA_TRACE(8 + 4 * (t / 2));
      fill(t);
      bool y0, y1 = false;
      const unsigned long long e0 = key_of(t, &y0);
      y0 = __builtin_amdgcn_readfirstlane((int)y0) != 0;
      const uint32_t u0 = __builtin_amdgcn_readfirstlane((uint32_t)(e0 >> 32));
      bool pair = t + 1 < trips;
      if (pair) {
        const unsigned long long e1 = key_of(t + 1, &y1);
        y1 = __builtin_amdgcn_readfirstlane((int)y1) != 0;
        // (a pair shares its first end point AND its link flag -- every edge of a node-strategy mini-batch does; an odd
        // one out is reduced alone)
        pair = __builtin_amdgcn_readfirstlane((uint32_t)(e1 >> 32)) == u0 && y1 == y0;
      }
      BETA_TRACE(8 + 4 * (t / 2) + 1);
      wait_rows(issued - (t + (pair ? 2u : 1u)));
      BETA_TRACE(8 + 4 * (t / 2) + 2);
      if (u0 != cur_u) {
        const float* ra = rpm_row(a.pi, u0);
#pragma unroll
        for (int p = 0; p < HP; ++p) pa[p] = f32x2{ra[tid + 2 * L * p], ra[tid + 2 * L * p + L]};
        // the loads are waited for HERE, inside the branch: left pending, hipcc's wait-count pass puts a vmcnt(0) in
        // front of pa's first use after the join -- on every trip, whether the branch was taken or not -- which
        // drains the rows in flight (the one-edge loop had exactly that until round 3: its ring never ran ahead)
#pragma unroll
        for (int p = 0; p < HP; ++p) asm volatile("" : "+v"(pa[p].x), "+v"(pa[p].y));
        cur_u = u0;
      }
      const float* row0 = ring + (t % D) * KW;
      if (pair) {
        // every edge of a node-strategy mini-batch has the same link flag (a link batch is all links, a non-link
        // batch all non-links, sample.cc:249-293): the two uniform cases get their own straight-line copy of the
        // step, with the per-column selects on y -- six v_cndmask per column pair and edge -- folded away
        auto pair_step = [&](auto mode) {
        asm volatile("" : "+v"(one));
        const float* row1 = ring + ((t + 1) % D) * KW;
        float sums[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // pi_a . pi_b and probs of edge t, of edge t + 1
        float lo0 = 1.0f, lo1 = 1.0f;
        f32x2 pr0[HP], pr1[HP];
#pragma unroll
        for (int p = 0; p < HP; ++p) {
          const f32x2 b0 = f32x2{row0[ln + 128 * p], row0[ln + 128 * p + 64]};
          const f32x2 b1 = f32x2{row1[ln + 128 * p], row1[ln + 128 * p + 64]};
          const f32x2 f0 = pa[p] * b0, f1 = pa[p] * b1;
          pr0[p] = sel_b(p, y0, mode) * f0;
          pr1[p] = sel_b(p, y1, mode) * f1;
          const float vx[4] = {f0.x, pr0[p].x, f1.x, pr1[p].x}, vy[4] = {f0.y, pr0[p].y, f1.y, pr1[p].y};
          VLn::template chain_rows<4>(sums, vx);
          VLn::template chain_rows<4>(sums, vy);
          const float m00 = fabsf(pr0[p].x), m01 = fabsf(pr0[p].y), m10 = fabsf(pr1[p].x), m11 = fabsf(pr1[p].y);
          lo0 = fminf(fminf(lo0, m00 == 0.0f ? 1.0f : m00), m01 == 0.0f ? 1.0f : m01);  // an exact zero divides exactly
          lo1 = fminf(fminf(lo1, m10 == 0.0f ? 1.0f : m10), m11 == 0.0f ? 1.0f : m11);
        }
        float tot[4];
        VLn::template tree_rows<4>(sums, tot);  // beta.cc:209-217, both edges
        BETA_TRACE(8 + 4 * (t / 2) + 3);
        const float w0 = sel_w(y0, mode), w1 = sel_w(y1, mode);
        const float ps0 = tot[1] + w0 * (1.0f - tot[0]), ps1 = tot[3] + w1 * (1.0f - tot[2]);
        // CALC_GRADS, beta.cc:161-171: edge t, then edge t + 1
        const bool fast0 = lo0 >= kProbsLo && in_range(ps0, kPsumLo, kPsumHi);
        const bool fast1 = lo1 >= kProbsLo && in_range(ps1, kPsumLo, kPsumHi);
        if (fast0 && fast1) {
          const float r0 = exact_rcp(ps0), r1 = exact_rcp(ps1);
          const f32x2 s0 = f32x2{ps0, ps0}, s1 = f32x2{ps1, ps1}, q0 = f32x2{r0, r0}, q1 = f32x2{r1, r1};
#pragma unroll
          for (int p = 0; p < HP; ++p) {
            const f32x2 g0 = div_exact3(pr0[p], s0, q0), g1 = div_exact3(pr1[p], s1, q1);
            acc0[p] += g0 * sel_0(p, y0, mode);
            acc1[p] += g0 * sel_1(p, y0, mode);
            acc0[p] += g1 * sel_0(p, y1, mode);
            acc1[p] += g1 * sel_1(p, y1, mode);
          }
        } else {
#pragma unroll
          for (int p = 0; p < HP; ++p) {
            const f32x2 g0 = f32x2{pr0[p].x / ps0, pr0[p].y / ps0};
            acc0[p] += g0 * sel_0(p, y0, mode);
            acc1[p] += g0 * sel_1(p, y0, mode);
            const f32x2 g1 = f32x2{pr1[p].x / ps1, pr1[p].y / ps1};
            acc0[p] += g1 * sel_0(p, y1, mode);
            acc1[p] += g1 * sel_1(p, y1, mode);
          }
        }
        };
        if (y0) pair_step(std::integral_constant<int, 2>{});
        else pair_step(std::integral_constant<int, 1>{});
        t += 2;
      } else {
        float sums[2] = {0.0f, 0.0f};
        float lo = 1.0f;
        f32x2 prr[HP];
#pragma unroll
        for (int p = 0; p < HP; ++p) {
          const f32x2 pb = f32x2{row0[ln + 128 * p], row0[ln + 128 * p + 64]};
          const f32x2 f = pa[p] * pb;
          const f32x2 pr = (y0 ? bk[p] : 1.0f - bk[p]) * f;
          prr[p] = pr;
          const float vx[2] = {f.x, pr.x}, vy[2] = {f.y, pr.y};
          VLn::template chain_rows<2>(sums, vx);
          VLn::template chain_rows<2>(sums, vy);
          const float m0 = fabsf(pr.x), m1 = fabsf(pr.y);
          lo = fminf(fminf(lo, m0 == 0.0f ? 1.0f : m0), m1 == 0.0f ? 1.0f : m1);
        }
        float tot[2];
        VLn::template tree_rows<2>(sums, tot);
        const float w = y0 ? EPS : (1.0f - EPS);
        const float probs_sum = tot[1] + w * (1.0f - tot[0]);
        if (lo >= kProbsLo && in_range(probs_sum, kPsumLo, kPsumHi)) {
          const float rps = exact_rcp(probs_sum);
          const f32x2 psum2 = f32x2{probs_sum, probs_sum}, rps2 = f32x2{rps, rps};
#pragma unroll
          for (int p = 0; p < HP; ++p) {
            const f32x2 f = div_exact3(prr[p], psum2, rps2);
            acc0[p] += f * (y0 ? noo[p] : d0n[p]);
            acc1[p] += f * (y0 ? d1l[p] : noo[p]);
          }
        } else {
#pragma unroll
          for (int p = 0; p < HP; ++p) {
            const f32x2 f = f32x2{prr[p].x / probs_sum, prr[p].y / probs_sum};
            acc0[p] += f * (y0 ? noo[p] : d0n[p]);
            acc1[p] += f * (y0 ? d1l[p] : noo[p]);
          }
        }
        t += 1;
      }
    }
  } else {
  for (uint32_t t = 0; t < D - 1 && t < trips; ++t) request(t);
  BETA_TRACE(3);
  for (uint32_t t = 0; t < trips; ++t) {
    float* row_b = ring + (t % D) * KW;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // slot (t - 1) % D has been read for the last time
    BETA_TRACE(8 + 4 * t);
    if (t + D - 1 < trips) {
      if (t + D - 1 >= tb + 128) {  // the look-ahead leaves the two key windows: slide them (t >= tb + 64 here)
        kv = kv_next;
        ym = ym_next;
        tb += 64;
        kv_next = load_keys(tb + 64, &ym_next);
      }
      request(t + D - 1);
      // row t landed, t+1 .. t+3 in flight.  vmcnt counts stores too and retires in issue order (gfx9 family): the
      // fused form's pi stores of the previous trip (ST instructions, issued after request(t + D - 2)) are younger
      // than row t and must not be waited for
      if (FUSE && t > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * PIECES + ST) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * PIECES) : "memory");
    } else {
      const uint32_t ahead = trips - 1 - t;  // 0 .. D-2 rows still in flight behind row t
      if (FUSE && t > 0) {
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PIECES + ST) : "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES + ST) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ST) : "memory");
      } else {
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PIECES) : "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    BETA_TRACE(8 + 4 * t + 1);
    bool y;
    const unsigned long long edge = key_of(t, &y);
    y = __builtin_amdgcn_readfirstlane((int)y) != 0;
    const uint32_t u = __builtin_amdgcn_readfirstlane((uint32_t)(edge >> 32));
    f32x2 pbn[FUSE ? HP : 1];
    if constexpr (FUSE) {
      // update_pi of the partner (the end point that is not the shared node): normalise its phi_vec row, store it
      const uint32_t vv = __builtin_amdgcn_readfirstlane((uint32_t)(edge & 0xffffffffu));
      const uint32_t partner = u == shared_node ? vv : u;
      float partial = 0.0f;
#pragma unroll
      for (int p = 0; p < HP; ++p) {
        pbn[p] = f32x2{row_b[ln + 128 * p], row_b[ln + 128 * p + 64]};
        VLn::chain(partial, pbn[p].x);
        VLn::chain(partial, pbn[p].y);
      }
      const float sum = VLn::tree(partial);
      float* dst = rpm_row(a.pi, partner);
#pragma unroll
      for (int p = 0; p < HP; ++p) pbn[p] = f32x2{pbn[p].x / sum, pbn[p].y / sum};
      // Non-temporal stores (AMMSB_BETA_PI_NT=1, off by default): the 268 MB of rows a C3 launch writes are not read
      // again before the next update_phi (the gradient takes them from registers), and left in the caches they are
      // written back UNDER that launch -- in one process, over one pi, update_phi runs 2-6 % longer after this kernel
      // than after separate update_pi + gradient launches.  With the hint it recovers half of that and this kernel
      // loses as much: whole step 1.7745 (plain) / 1.7825 (nt) / 1.8187 ms (separate launches) on one box
      // (tools/phi_in_sequence.py, profiles/r04_phi_in_sequence.txt).  (wave-uniform branch)
      if (a.pi_nt) {
#pragma unroll
        for (int p = 0; p < HP; ++p) {
          __builtin_nontemporal_store(pbn[p].x, dst + tid + 2 * L * p);
          __builtin_nontemporal_store(pbn[p].y, dst + tid + 2 * L * p + L);
        }
      } else {
#pragma unroll
        for (int p = 0; p < HP; ++p) {
          dst[tid + 2 * L * p] = pbn[p].x;
          dst[tid + 2 * L * p + L] = pbn[p].y;
        }
      }
      if (tid == 0) a.fuse.phi_sum[partner] = sum;
    } else if (u != cur_u) {
      const float* ra = rpm_row(a.pi, u);
#pragma unroll
      for (int p = 0; p < HP; ++p) pa[p] = f32x2{ra[tid + 2 * L * p], ra[tid + 2 * L * p + L]};
#pragma unroll
      for (int p = 0; p < HP; ++p) asm volatile("" : "+v"(pa[p].x), "+v"(pa[p].y));  // (waited for inside the branch, see above)
      cur_u = u;
    }

    // (the link flag is wave-uniform: each of its two values gets its own copy of the trip, without the per-column
    // selects on y)
    auto trip = [&](auto mode) {
    asm volatile("" : "+v"(one));
    // CALC_PROBS, beta.cc:145-160
    // (probs[] stays in registers between the two passes: the write-back into the ring slot and its re-read cost a
    // wave ~15 % of a trip -- in-kernel stamps, tools/beta_trace.sh -- and 16 registers do not change the occupancy)
    // the two sums of an edge (pi_a . pi_b and the probs, beta.cc:209-217) advance together: one wave per slot takes
    // VLane's transposed two-row form (at wg 32 one swap per column pair for both chains; one tree for both sums)
    float sums[2] = {0.0f, 0.0f};
    float lo = 1.0f;
    f32x2 prr[HP];
#pragma unroll
    for (int p = 0; p < HP; ++p) {
      f32x2 pb;
      if constexpr (FUSE) pb = pbn[p];
      else pb = f32x2{row_b[ln + 128 * p], row_b[ln + 128 * p + 64]};
      const f32x2 f = pa[p] * pb;
      const f32x2 pr = sel_b(p, y, mode) * f;
      prr[p] = pr;
      if constexpr (W == 1) {
        const float vx[2] = {f.x, pr.x}, vy[2] = {f.y, pr.y};
        VLn::template chain_rows<2>(sums, vx);
        VLn::template chain_rows<2>(sums, vy);
      } else {
        sums[0] += f.x;
        sums[0] += f.y;
        sums[1] += pr.x;
        sums[1] += pr.y;
      }
      const float m0 = fabsf(pr.x), m1 = fabsf(pr.y);
      lo = fminf(fminf(lo, m0 == 0.0f ? 1.0f : m0), m1 == 0.0f ? 1.0f : m1);  // an exact zero divides exactly
    }
    float pi_sum = sums[0], probs_sum = sums[1];
    BETA_TRACE(8 + 4 * t + 2);
    if constexpr (W == 1) {
      float out2[2];
      VLn::template tree_rows<2>(sums, out2);
      pi_sum = out2[0];
      probs_sum = out2[1];
    } else {
      group_sum2(pi_sum, probs_sum);  // beta.cc:209-217
    }
    BETA_TRACE(8 + 4 * t + 3);
    const float w = sel_w(y, mode);
    const float prob_0 = w * (1.0f - pi_sum);
    probs_sum += prob_0;

    // CALC_GRADS, beta.cc:161-171
    if (lo >= kProbsLo && in_range(probs_sum, kPsumLo, kPsumHi)) {
      const float rps = exact_rcp(probs_sum);
      const f32x2 psum2 = f32x2{probs_sum, probs_sum}, rps2 = f32x2{rps, rps};
#pragma unroll
      for (int p = 0; p < HP; ++p) {
        const f32x2 f = div_exact3(prr[p], psum2, rps2);
        acc0[p] += f * sel_0(p, y, mode);
        acc1[p] += f * sel_1(p, y, mode);
      }
    } else {
#pragma unroll
      for (int p = 0; p < HP; ++p) {
        const f32x2 f = f32x2{prr[p].x / probs_sum, prr[p].y / probs_sum};
        acc0[p] += f * sel_0(p, y, mode);
        acc1[p] += f * sel_1(p, y, mode);
      }
    }
    };
    if (y) trip(std::integral_constant<int, 2>{});
    else trip(std::integral_constant<int, 1>{});
  }

  }
  BETA_TRACE(4);
  float* out = a.partials + (uint64_t)gs * 2 * K;
#pragma unroll
  for (int p = 0; p < HP; ++p) {
    *reinterpret_cast<float2*>(out + 2 * (tid + 2 * L * p)) = make_float2(acc0[p].x, acc1[p].x);
    *reinterpret_cast<float2*>(out + 2 * (tid + 2 * L * p + L)) = make_float2(acc0[p].y, acc1[p].y);
  }
  BETA_BLK(1);
}

template <int KPT, int W, bool FUSE = false, int VL = 64>
__global__ __launch_bounds__(64 * W) void beta_grads_lds_kernel(const BetaArgs a) {
  beta_grads_lds_body<KPT, W, FUSE, VL>(a);
}
// K = 1024 on one wave per slot (the C3 shape): the launch is 2 048 slots = 8 waves per CU = two per SIMD, and it runs
// beside the sampling chain of the mini-batch two steps ahead.  The register budget is said out loud (two waves per
// SIMD); what matters beyond it is to stay well below 256: a build that used all of them (a third, run-time-selected
// copy of the pair step kept every per-column constant alive) left a SIMD that also held one small wave of another
// kernel room for only ONE of these waves, and the launch took a second round inside the loop (0.093 -> 0.118 ms)
// while the same kernel launched alone did not change.  206 now.
template <>
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(AMMSB_BETA_VGPRS))) void beta_grads_lds_kernel<16, 1, false, 64>(const BetaArgs a) {
  beta_grads_lds_body<16, 1, false, 64>(a);
}
template <>
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(AMMSB_BETA_VGPRS))) void beta_grads_lds_kernel<16, 1, false, 32>(const BetaArgs a) {
  beta_grads_lds_body<16, 1, false, 32>(a);
}

template <int KPT, int W, int VL = 64>
int launch_grads_lds(ammsb_ctx* ctx, const BetaArgs& a, hipStream_t s) {
  const size_t lds = (size_t)W * 4 * sizeof(float) * 64 * KPT;
  static const std::string name = ammsb_kname("beta_grads_lds_kernel<%d, %d, false, %d>", KPT, W, VL);
  static const std::string name_fused = ammsb_kname("beta_grads_lds_kernel<%d, %d, true, %d>", KPT, W, VL);
  ctx->kernel_name[AMMSB_KN_GRADS] = (a.fuse.phi_vec ? name_fused : name).c_str();
  if constexpr (W == 1) {
    if (a.fuse.phi_vec) {
      beta_grads_lds_kernel<KPT, 1, true, VL><<<a.P, 64, lds, s>>>(a);
      AMMSB_LAUNCH_CHECK(ctx);
      return AMMSB_OK;
    }
  }
  beta_grads_lds_kernel<KPT, W, false, VL><<<a.P, 64 * W, lds, s>>>(a);
  AMMSB_LAUNCH_CHECK(ctx);
  return AMMSB_OK;
}

// sum_grads: grads[c] = sum over the P partial rows in a fixed order.  16 columns x 16 row-lanes per
// block (2K/16 blocks, so the whole chip takes part): row-lane r adds rows r, r+16, ... ascending,
// then the 16 lane sums are added by the halving tree r += r+8, +4, +2, +1.
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* partials, uint32_t P, uint32_t cols,
                                                            float* out, const ammsb_step_desc* desc) {
  __shared__ float red[16][17];
  if (ammsb_desc_skip(desc)) return;
  if (desc) P = desc->n_edges < P ? desc->n_edges : P;  // the gradient kernel's slot count (beta_step)
  const uint32_t cl = threadIdx.x & 15, r = threadIdx.x >> 4;
  const uint32_t c = blockIdx.x * 16 + cl;
  float s = 0.0f;
  if (c < cols)
    for (uint32_t p = r; p < P; p += 16) s += partials[(uint64_t)p * cols + c];
  red[r][cl] = s;
  __syncthreads();
  for (uint32_t h = 8; h > 0; h >>= 1) {
    if (r < h) red[r][cl] += red[r + h][cl];
    __syncthreads();
  }
  if (r == 0 && c < cols) out[c] = red[0][cl];
}

// The same reduction for cols % 8 == 0 with wide loads: a block owns 8 columns (two float4 lanes) and
// 128 row-lanes; row-lane r adds rows r, r + 128, ... ascending (all loads independent, in flight
// together), then the halving tree r += r + 64, ..., + 1.  2K/8 blocks.
__global__ __launch_bounds__(256) void sum_partials8_kernel(const float* partials, uint32_t P, uint32_t cols,
                                                             float* out, const ammsb_step_desc* desc) {
  __shared__ float4 red[128][2];
  if (ammsb_desc_skip(desc)) return;
  if (desc) P = desc->n_edges < P ? desc->n_edges : P;  // the gradient kernel's slot count (beta_step)
  const uint32_t h = threadIdx.x & 1, r = threadIdx.x >> 1;
  const uint32_t c = blockIdx.x * 8 + 4 * h;
  float4 s = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  for (uint32_t p = r; p < P; p += 128) {
    const float4 v = *reinterpret_cast<const float4*>(partials + (uint64_t)p * cols + c);
    s.x += v.x;
    s.y += v.y;
    s.z += v.z;
    s.w += v.w;
  }
  red[r][h] = s;
  __syncthreads();
  for (uint32_t half = 64; half > 0; half >>= 1) {
    if (r < half) {
      float4 x = red[r][h];
      const float4 y = red[r + half][h];
      x.x += y.x;
      x.y += y.y;
      x.z += y.z;
      x.w += y.w;
      red[r][h] = x;
    }
    __syncthreads();
  }
  if (r == 0) *reinterpret_cast<float4*>(out + c) = red[0][h];
}

// update_theta (beta.cc:51-82) + beta = pair-normalised theta (beta.cc:376-383; Normalizer slice 2,
// wg 1: lsum = (0 + t0) + t1) for component k: stream k draws r0 for theta[k,0], then r1 for theta[k,1].
// coef / theta_sum (descriptor loop; null on the eager path): the stepped theta's gradient constants for the NEXT
// iteration's gradient kernel (theta_coef above) -- nobody else changes theta between this step and that launch.
__device__ __forceinline__ void theta_step(uint32_t k, float g0, float g1, float* theta, float* beta, ammsb_seed* seeds,
                                           float eps_t, float scale, float eta0, float eta1, uint32_t noise_on,
                                           const ZigTables* zig, float4* coef = nullptr, float* theta_sum = nullptr) {
  ammsb_seed rs = seeds[k];
  const float half = eps_t / 2.0f;
  float th[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const float r = noise_on ? rng_normal(rs, zig) : 1.0f;
    const float g = c == 0 ? g0 : g1;
    const float t = theta[2 * k + c];
    const float eta = c == 0 ? eta0 : eta1;
    const float ep = eps_t * t;
    const float f = sqrtf(ep);
    const float sg = scale * g;
    float in = eta - t;
    in = in + sg;
    const float drift = half * in;
    const float aa = t + drift;
    const float bb = f * r;
    const float v = fabsf(aa + bb);
    th[c] = v > 1e-24f ? v : 1e-24f;
  }
  seeds[k] = rs;
  theta[2 * k] = th[0];
  theta[2 * k + 1] = th[1];
  float lsum = 0.0f;
  lsum += th[0];
  lsum += th[1];
  beta[2 * k] = th[0] / lsum;
  const float b1 = th[1] / lsum;
  beta[2 * k + 1] = b1;
  if (coef) theta_coef(k, th[0], th[1], b1, coef, theta_sum);
}

// captured graph: hand the next two descriptors of the ring to the graph that runs next, advance the cursor
__device__ __forceinline__ void step_advance(const ammsb_step_advance& adv, const ammsb_step_desc* desc) {
  if (!adv.ring) return;
  if (ammsb_desc_skip(desc)) {  // this step was skipped (a wait gave up earlier): so is the next one; count nothing
    adv.cur_out->n_nodes = 0;
    adv.cur_out->n_edges = 0;
    return;
  }
  unsigned long long* stamp = adv.stamps ? adv.stamps + AMMSB_STAMP_SLOTS * (desc->step % AMMSB_STAMP_CAP) : nullptr;
  const uint32_t c = *adv.cursor;
  *adv.cur_out = adv.ring[c + 1];
  *adv.nxt_out = adv.ring[c + adv.nxt_offset];
  *adv.cursor = c + 1;
  if (adv.main_seq) {
    // releases the sampler chain that overwrites the buffer set this step has read and reads the descriptor above
    __threadfence();
    const uint32_t done = *adv.main_seq + 1u;
    __hip_atomic_store(adv.main_seq, done, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if (stamp) stamp[AMMSB_STAMP_RELEASED] = wall_clock64();
    if (adv.avail) {  // hold this kernel (one lane of one block) until the next step's mini-batch has been sampled
      const unsigned long long t0 = wall_clock64();
      while ((int)(__hip_atomic_load(adv.avail, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - done) < 1) {
        // gives up after max_ticks, or at once when another wait already has (the sampling chain skips from then on):
        // the next step must not run on a mini-batch that is not there -- it gets a skip descriptor, and hands it on
        const bool gave_up = wall_clock64() - t0 > adv.max_ticks;
        if (gave_up || __hip_atomic_load(adv.timeouts, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
          if (gave_up) atomicAdd(adv.timeouts, 1u);
          adv.cur_out->n_nodes = 0;
          adv.cur_out->n_edges = 0;
          break;
        }
        __builtin_amdgcn_s_sleep(8);
      }
    }
  }
  if (stamp) {
    if (!adv.main_seq) stamp[AMMSB_STAMP_RELEASED] = wall_clock64();
    stamp[AMMSB_STAMP_NEXT] = wall_clock64();
  }
}

__global__ __launch_bounds__(64) void update_theta_kernel(float* theta, float* beta, const float* grads,
                                                           ammsb_seed* seeds, uint32_t K, float eps_t, float scale,
                                                           float eta0, float eta1, uint32_t noise_on,
                                                           const ammsb_step_desc* desc, const ammsb_step_advance adv,
                                                           float4* coef, float* theta_sum) {
  __shared__ ZigTables zig;
  zig_load(&zig);
  __syncthreads();
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (desc) {  // captured graph: this step's scalars
    eps_t = desc->eps_t;
    scale = desc->scale;
    if (ammsb_desc_skip(desc)) {  // (uniform) a skipped step: theta and its streams stay as they are
      if (k == 0) step_advance(adv, desc);
      return;
    }
    note_stamp(adv.stamps, desc, AMMSB_STAMP_THETA);
    if (k == 0) step_advance(adv, desc);
  }
  if (k >= K) return;
  theta_step(k, grads[2 * k], grads[2 * k + 1], theta, beta, seeds, eps_t, scale, eta0, eta1, noise_on, &zig, coef, theta_sum);
}

// sum_grads + update_theta in one launch (captured graph): block b reduces columns 8b .. 8b+7 of the partial rows
// exactly as sum_partials8_kernel does (same row-lane assignment, same tree), stores them to grads_out
// (BetaUpdater::GetGrads()), and its first four threads then run the theta step of components 4b .. 4b+3.
__global__ __launch_bounds__(256) void sum_update_theta_kernel(const float* partials, uint32_t P, uint32_t cols,
                                                                float* grads_out, float* theta, float* beta,
                                                                ammsb_seed* seeds, float eta0, float eta1,
                                                                uint32_t noise_on, const ammsb_step_desc* desc,
                                                                const ammsb_step_advance adv, float4* coef,
                                                                float* theta_sum) {
  __shared__ float4 red[128][2];
  __shared__ ZigTables zig;
  if (ammsb_desc_skip(desc)) {  // (uniform) a skipped step: gradient, theta, beta and the streams stay as they are
    if (blockIdx.x == 0 && threadIdx.x == 64) step_advance(adv, desc);
    return;
  }
  zig_load(&zig);
  note_stamp(adv.stamps, desc, AMMSB_STAMP_THETA);
  P = desc->n_edges < P ? desc->n_edges : P;  // the gradient kernel's slot count (beta_step)
  const uint32_t h = threadIdx.x & 1, r = threadIdx.x >> 1;
  const uint32_t c = blockIdx.x * 8 + 4 * h;
  float4 s = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  // (the loads of four partial rows are issued together; the adds keep the row order p = r, r + 128, ...)
  uint32_t p = r;
  for (; p + 384 < P; p += 512) {
    float4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const float4*>(partials + (uint64_t)(p + 128 * i) * cols + c);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s.x += v[i].x;
      s.y += v[i].y;
      s.z += v[i].z;
      s.w += v[i].w;
    }
  }
  for (; p < P; p += 128) {
    const float4 v = *reinterpret_cast<const float4*>(partials + (uint64_t)p * cols + c);
    s.x += v.x;
    s.y += v.y;
    s.z += v.z;
    s.w += v.w;
  }
  red[r][h] = s;
  __syncthreads();
  for (uint32_t half = 64; half > 0; half >>= 1) {
    if (r < half) {
      float4 x = red[r][h];
      const float4 y = red[r + half][h];
      x.x += y.x;
      x.y += y.y;
      x.z += y.z;
      x.w += y.w;
      red[r][h] = x;
    }
    __syncthreads();
  }
  if (r == 0) *reinterpret_cast<float4*>(grads_out + c) = red[0][h];
  // the hand-over (three dependent round trips, then the poll for the next mini-batch) runs in the block's SECOND wave,
  // beside the theta step of the first: neither waits for the other
  if (blockIdx.x == 0 && threadIdx.x == 64) step_advance(adv, desc);
  if (threadIdx.x < 4) {
    const uint32_t k = blockIdx.x * 4 + threadIdx.x;  // columns 2k, 2k+1 = words (t & 1) * 2, +1 of red[0][t >> 1]
    const float4 g = red[0][threadIdx.x >> 1];
    const float g0 = (threadIdx.x & 1) ? g.z : g.x, g1 = (threadIdx.x & 1) ? g.w : g.y;
    theta_step(k, g0, g1, theta, beta, seeds, desc->eps_t, desc->scale, eta0, eta1, noise_on, &zig, coef, theta_sum);
  }
}

__global__ void beta_from_theta_kernel(const float* theta, float* beta, uint32_t K) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  const float t0 = theta[2 * k], t1 = theta[2 * k + 1];
  float lsum = 0.0f;
  lsum += t0;
  lsum += t1;
  beta[2 * k] = t0 / lsum;
  beta[2 * k + 1] = t1 / lsum;
}

// ---------------------------------------------------------------------------------------------------------
// Generic form: any K <= 1024 * (blockDim.x / 64), any power-of-two reference work-group size L <= blockDim.x / 2.
//
// The reference's per-work-group kernel loops K_PER_THREAD = ceil(K / L) columns per work-item generically
// (beta.cc:145-233) and its default beta_wg_size is 32 (main.cc:64): K = 1024 means 32 columns, K = 4096 means 128
// columns per work-item -- more than a lane's registers hold next to the 2 K / L accumulators.  As in
// update_phi_gen_kernel (ammsb_phi.hip) the elementwise work of a slot is spread over all T = blockDim.x threads
// (thread t owns columns t + T i) independently of L, and the two WG_SUMs of an edge (beta.cc:209-217) are emulated
// lane by lane with the reference's association order (vgroup_sum<2>, ammsb_dev.h).  The slot -> edge assignment, the
// per-column accumulation order and every operation are those of beta_grads_kernel<L, KPT>: the partial rows are
// bit-identical to it wherever both run.
template <int CPT>
__global__ __launch_bounds__(512) void beta_grads_gen_kernel(const BetaArgs a, uint32_t L, uint32_t lgL) {
  extern __shared__ __align__(16) char smem[];  // [K] f, [K] probs, [2 L] lane partials, [4] sums, [64] keys, [64] links
  const uint32_t K = a.K, T = blockDim.x, t = threadIdx.x;
  float* s_f = reinterpret_cast<float*>(smem);
  float* s_p = s_f + K;
  float* s_aux = s_p + K;
  float* s_res = s_aux + 2 * L;
  unsigned long long* s_key = reinterpret_cast<unsigned long long*>(s_res + 4);
  uint32_t* s_link = reinterpret_cast<uint32_t*>(s_key + 64);
  const BetaStep st = beta_step(a);
  beta_stamp<false>(a);
  const uint32_t gs = blockIdx.x;  // partial-row slot
  if (gs >= st.P) return;          // block-uniform
  const float EPS = a.epsilon;

  auto col = [&](int j) -> uint32_t { return t + (uint32_t)j * T; };
  auto has = [&](int j) -> bool { return t + (uint32_t)j * T < K; };

  float bk[CPT], d0n[CPT], d1l[CPT], noo[CPT];
#pragma unroll
  for (int j = 0; j < CPT; ++j) {
    const uint32_t k = col(j);
    if (k < K) {
      const float4 c = a.coef[k];  // (theta_coef above)
      bk[j] = c.x;
      d0n[j] = c.y;
      d1l[j] = c.z;
      noo[j] = c.w;
    } else {
      bk[j] = d0n[j] = d1l[j] = noo[j] = 0.0f;
    }
  }
  float acc0[CPT], acc1[CPT];
#pragma unroll
  for (int j = 0; j < CPT; ++j) acc0[j] = acc1[j] = 0.0f;

  // edges of this slot: e(r) = edge_begin + gs + r * P, r < trips
  const uint32_t n_edges = st.edge_end - st.edge_begin;
  const uint32_t trips = gs < n_edges ? (n_edges - gs + st.P - 1) / st.P : 0;  // block-uniform
  int phase = 0;
  float pa[CPT], pb[CPT], na[CPT], nb[CPT];
  auto load_rows = [&](float (&da)[CPT], float (&db)[CPT], uint32_t slot) {
    const unsigned long long edge = s_key[slot];
    const uint32_t u = __builtin_amdgcn_readfirstlane((uint32_t)(edge >> 32));
    const uint32_t v = __builtin_amdgcn_readfirstlane((uint32_t)(edge & 0xffffffffu));
    const float* ra = rpm_row(a.pi, u);
    const float* rb = rpm_row(a.pi, v);
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      const uint32_t k = col(j), ck = k < K ? k : K - 1;
      da[j] = ra[ck];
      db[j] = rb[ck];
    }
  };

  for (uint32_t tb = 0; tb < trips; tb += 64) {
    // keys and link bits of the next 64 trips: one probe per thread of the first wave
    __syncthreads();
    if (t < 64) {
      const bool ok = tb + t < trips;
      const uint64_t e = (uint64_t)st.edge_begin + gs + (uint64_t)(tb + t) * st.P;
      const unsigned long long edge = a.edges[ok ? e : (uint64_t)st.edge_begin + gs];
      const uint32_t u = (uint32_t)(edge >> 32), v = (uint32_t)(edge & 0xffffffffu);
      s_key[t] = edge;
      s_link[t] = set_has(a.set, make_edge(u, v)) ? 1u : 0u;
    }
    __syncthreads();
    const uint32_t cnt = trips - tb < 64u ? trips - tb : 64u;
    load_rows(pa, pb, 0);
    for (uint32_t r = 0; r < cnt; ++r) {
      load_rows(na, nb, r + 1 < cnt ? r + 1 : r);  // unconditional: the last trip of a batch re-requests its own rows
      const bool y = __builtin_amdgcn_readfirstlane(s_link[r]) != 0;
      float probs[CPT];
      float lo = 1.0f;
#pragma unroll
      for (int j = 0; j < CPT; ++j) {  // CALC_PROBS, beta.cc:145-160
        const float f = (has(j) ? pa[j] : 0.0f) * pb[j];
        probs[j] = y ? bk[j] * f : (1.0f - bk[j]) * f;
        if (has(j)) {
          s_f[col(j)] = f;
          s_p[col(j)] = probs[j];
        }
        const float m = fabsf(probs[j]);
        lo = fminf(lo, m == 0.0f ? 1.0f : m);
      }
      __syncthreads();
      const float* const vv[2] = {s_f, s_p};
      float sums[2];
      vgroup_sum<2>(vv, K, L, lgL, s_aux, s_res, phase, sums);  // beta.cc:209-217
      const float pi_sum = sums[0];
      float probs_sum = sums[1];
      const float w = y ? EPS : (1.0f - EPS);
      const float prob_0 = w * (1.0f - pi_sum);
      probs_sum += prob_0;
      // CALC_GRADS, beta.cc:161-171
      if (lo >= kProbsLo && in_range(probs_sum, kPsumLo, kPsumHi)) {
        const float rps = refined_rcp(probs_sum);
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
          const float f = div_with_rcp(probs[j], probs_sum, rps);
          acc0[j] += f * (y ? noo[j] : d0n[j]);
          acc1[j] += f * (y ? d1l[j] : noo[j]);
        }
      } else {
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
          const float f = probs[j] / probs_sum;
          acc0[j] += f * (y ? noo[j] : d0n[j]);
          acc1[j] += f * (y ? d1l[j] : noo[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < CPT; ++j) {
        pa[j] = na[j];
        pb[j] = nb[j];
      }
    }
  }

  float* out = a.partials + (uint64_t)gs * 2 * K;
#pragma unroll
  for (int j = 0; j < CPT; ++j)
    if (has(j)) *reinterpret_cast<float2*>(out + 2 * col(j)) = make_float2(acc0[j], acc1[j]);
}

constexpr uint32_t kGenMaxK = 8192;  // 512 threads x 16 columns
inline int gen_cpt(uint64_t K) { return K <= 4096 ? 8 : 16; }  // columns per thread (as in ammsb_phi.hip)

// threads per block: enough for the columns and two virtual lanes' worth of chains; 0 if the shape does not fit
inline uint32_t gen_threads(uint64_t K, uint32_t L) {
  if (K > kGenMaxK) return 0;
  const uint32_t per_wave = 64u * (uint32_t)gen_cpt(K);
  uint32_t T = 64u * (uint32_t)((K + per_wave - 1) / per_wave);
  if (T < 2 * L) T = 2 * L;
  if (T < 64) T = 64;
  return T <= 512 ? T : 0;
}

int launch_grads_gen(ammsb_ctx* ctx, const BetaArgs& a, uint32_t wg, hipStream_t s) {
  const uint32_t T = gen_threads(a.K, wg);
  if (!T) return AMMSB_ERANGE;
  const size_t lds = sizeof(float) * (2 * (size_t)a.K + 2 * wg + 4) + 64 * (sizeof(unsigned long long) + sizeof(uint32_t));
  ctx->kernel_name[AMMSB_KN_GRADS] = gen_cpt(a.K) == 8 ? "beta_grads_gen_kernel<8>" : "beta_grads_gen_kernel<16>";
  if (gen_cpt(a.K) == 8) beta_grads_gen_kernel<8><<<a.P, T, lds, s>>>(a, wg, ilog2_u32(wg));
  else beta_grads_gen_kernel<16><<<a.P, T, lds, s>>>(a, wg, ilog2_u32(wg));
  AMMSB_LAUNCH_CHECK(ctx);
  return AMMSB_OK;
}

template <int L, int KPT>
int launch_grads(ammsb_ctx* ctx, const BetaArgs& a, hipStream_t s) {
  using Grp = Group<L>;
  const uint32_t blocks = (a.P + Grp::PER_BLOCK - 1) / Grp::PER_BLOCK;
  static const std::string name = ammsb_kname("beta_grads_kernel<%d, %d, false, false>", L, KPT);
  static const std::string name_fused = ammsb_kname("beta_grads_kernel<%d, %d, true, false>", L, KPT);
  static const std::string name_fused1 = ammsb_kname("beta_grads_kernel<%d, %d, true, true>", L, KPT);
  ctx->kernel_name[AMMSB_KN_GRADS] = (a.fuse.phi_vec ? name_fused : name).c_str();
  if constexpr ((L == 32 || L == 64) && KPT <= 2) {  // the short-row shapes that take the fusion (beta_fuse_shape)
    if (a.fuse.phi_vec) {
      if (a.pi.num_blocks == 1) {  // (the launch-latency shapes, C1: the single-block instantiation pays there)
        ctx->kernel_name[AMMSB_KN_GRADS] = name_fused1.c_str();
        beta_grads_kernel<L, KPT, true, true><<<blocks, Grp::BLOCK, 0, s>>>(a);
      } else {
        beta_grads_kernel<L, KPT, true><<<blocks, Grp::BLOCK, 0, s>>>(a);
      }
      AMMSB_LAUNCH_CHECK(ctx);
      return AMMSB_OK;
    }
  }
  beta_grads_kernel<L, KPT><<<blocks, Grp::BLOCK, 0, s>>>(a);
  AMMSB_LAUNCH_CHECK(ctx);
  return AMMSB_OK;
}

inline int pick_kpt(uint64_t K, uint32_t L) {
  const uint64_t need = (K + L - 1) / L;
  for (int c : {1, 2, 4, 8, 16})
    if ((uint64_t)c >= need) return c;
  return 0;
}

}  // namespace

#define AMMSB_DISPATCH_KPT16(kpt, ...)                                \
  switch (kpt) {                                                      \
    case 1: { constexpr int KPT_ = 1; __VA_ARGS__; } break;           \
    case 2: { constexpr int KPT_ = 2; __VA_ARGS__; } break;           \
    case 4: { constexpr int KPT_ = 4; __VA_ARGS__; } break;           \
    case 8: { constexpr int KPT_ = 8; __VA_ARGS__; } break;           \
    case 16: { constexpr int KPT_ = 16; __VA_ARGS__; } break;         \
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

// shapes that take the fused form of beta_grads_lds_kernel<KPT, 1>
static bool beta_fuse_shape(ammsb_ctx* ctx, uint32_t wg) {
  const uint32_t K = (uint32_t)ctx->params.K;
  static const bool off = [] {
    const char* f = getenv("AMMSB_BETA_FORM");
    const char* g = getenv("AMMSB_LOOP_FUSE_PI");
    return (f && (f[0] == 'r' || f[0] == 'g')) || (g && atoi(g) == 0);
  }();
  // K = 1024 too since round 4 (AMMSB_LOOP_FUSE_PI=1 keeps it to K <= 512, the earlier default): with the gradient
  // constants from the per-theta table and the per-flag copies of the trip, the fused launch takes 0.111-0.114 ms at C3
  // against 0.084 (update_pi) + 0.089 (gradient) separately -- one launch and one pass over 268 MB of rows less; the
  // non-link step 1.905 -> 1.86-1.875 ms although update_phi's stamped time grows by 0.02 ms (the shorter main chain
  // moves more of the concurrent sampling chain under it); same-box alternation, gpurun_out/r04/call10.log.
  static const bool k1024 = !(getenv("AMMSB_LOOP_FUSE_PI") && atoi(getenv("AMMSB_LOOP_FUSE_PI")) == 1);
  if (off) return false;
  if ((wg == 64 || wg == 32) && (K == 256 || K == 512 || (K == 1024 && k1024))) return true;  // beta_grads_lds_kernel<KPT, 1, true, wg>
  return (wg == 32 || wg == 64) && K <= 2 * wg;                                   // beta_grads_kernel<L, 1 | 2, true>
}

bool ammsb_beta_can_fuse_pi(ammsb_ctx* ctx, uint32_t phi_wg, uint32_t beta_wg) {
  // update_pi's WG_SUM is over phi_wg lanes with update_pi_kernel<phi_wg, KPT>'s column ownership: the fused form
  // reproduces it when the gradient uses the same work-group size
  return ctx && phi_wg == beta_wg && beta_fuse_shape(ctx, beta_wg);
}

#ifndef AMMSB_BETA_PI_NT_DEFAULT
#define AMMSB_BETA_PI_NT_DEFAULT 0
#endif
static int g_beta_pi_nt = -1;  // in-process A/B (tools/phi_in_sequence.py): -1 = follow AMMSB_BETA_PI_NT
extern "C" int ammsb_debug_beta_pi_nt(int on) {
  g_beta_pi_nt = on;
  return AMMSB_OK;
}

static int beta_grads_common(ammsb_ctx* ctx, const float* theta, const float* beta, const ammsb_rpm* pi,
                             const ammsb_set* training_set, const uint64_t* edges, uint32_t n_edges,
                             uint32_t edge_begin, uint32_t edge_end, uint32_t wg, float* grads_out,
                             const ammsb_step_desc* desc, bool sum_rows, uint32_t* slots_out, void* stream,
                             const ammsb_pi_fusion* fuse = nullptr, unsigned long long* stamps = nullptr,
                             bool coef_ready = false) {
  AMMSB_CHECK_ARG(ctx, ctx && theta && beta && pi && training_set && edges && grads_out, "null argument");
  AMMSB_CHECK_ARG(ctx, pi->num_blocks >= 1 && pi->num_blocks <= AMMSB_RPM_MAX_BLOCKS && pi->rows_in_block > 0,
                  "bad pi descriptor");
  AMMSB_CHECK_ARG(ctx, pi->num_cols == ctx->params.K, "pi cols != K");
  AMMSB_CHECK_ARG(ctx, training_set->slots && training_set->num_bins > 0 && training_set->prime_idx < 4,
                  "bad set descriptor");
  AMMSB_CHECK_ARG(ctx, is_pow2(wg) && wg >= 16 && wg <= 1024, "beta wg must be a power of two in [16, 1024]");
  if (edge_end > n_edges) edge_end = n_edges;
  hipStream_t s = as_stream(stream);
  const uint32_t K = (uint32_t)ctx->params.K;
  if (edge_begin >= edge_end) {  // an empty shard contributes zero
    AMMSB_HIP(ctx, hipMemsetAsync(grads_out, 0, sizeof(float) * 2 * K, s));
    return AMMSB_OK;
  }
  const int kpt = pick_kpt(K, wg);
  // AMMSB_BETA_FORM=g: the generic kernel wherever it fits (tests compare it with the specialised ones)
  static const bool force_gen = [] {
    const char* f = getenv("AMMSB_BETA_FORM");
    return f && f[0] == 'g';
  }();
  const bool generic = kpt == 0 || (force_gen && gen_threads(K, wg) != 0 && !(fuse && fuse->phi_vec));
  if (generic && gen_threads(K, wg) == 0) {
    snprintf(ctx->err, sizeof ctx->err, "ammsb_beta_grads: K=%u at wg=%u: more than 16 columns per work-item needs K <= %u",
             K, wg, kGenMaxK);
    return AMMSB_ERANGE;
  }
  BetaArgs a;
  a.theta = theta;
  a.beta = beta;
  a.pi = *pi;
  a.set = dev_set(*training_set);
  a.edges = edges;
  a.partials = ctx->grad_partials;
  a.coef = ctx->theta_coef;
  // the per-column constants of this theta (theta_coef): computed here, unless the caller -- the descriptor loop, whose
  // theta step writes them and whose run start calls ammsb_theta_coef_d -- vouches for the table
  if (!coef_ready) {
    theta_coef_kernel<<<(K + 255) / 256, 256, 0, s>>>(theta, beta, ctx->theta_coef, ctx->theta_sum, K);
    AMMSB_LAUNCH_CHECK(ctx);
  }
  a.edge_begin = edge_begin;
  a.edge_end = edge_end;
  a.K = K;
  a.epsilon = ctx->params.epsilon;
  a.desc = desc;
  a.stamps = desc ? stamps : nullptr;
  a.fuse = ammsb_pi_fusion{nullptr, nullptr, nullptr, nullptr};
  {
    static const int nt_env = [] {  // AMMSB_BETA_PI_NT=0|1 (A/B runs)
      const char* f = getenv("AMMSB_BETA_PI_NT");
      return f ? atoi(f) : AMMSB_BETA_PI_NT_DEFAULT;
    }();
    const uint64_t pi_bytes = pi->num_rows * pi->num_cols * sizeof(float);
    // (a pi that fits the 256 MB Infinity Cache is re-read from there: keep its rows cached, as update_phi's hint does)
    a.pi_nt = (g_beta_pi_nt >= 0 ? g_beta_pi_nt : nt_env) != 0 && pi_bytes > (256ull << 20) ? 1u : 0u;
  }
  if (fuse && fuse->phi_vec) {
    // (the slots of a fused launch write every pi row of the mini-batch exactly once: only over the whole batch)
    AMMSB_CHECK_ARG(ctx, fuse->phi_sum && fuse->nodes && beta_fuse_shape(ctx, wg) && (desc || (edge_begin == 0 && edge_end == n_edges)),
                    "update_pi fusion needs the whole mini-batch and a shape the fused kernels take");
    a.fuse = *fuse;
  }
  const uint32_t span = edge_end - edge_begin;
  // enough slots to fill an MI355X (256 CUs, ~8 waves each), never more than there are edges.  A function of
  // (span, wg) only -- not of the CU count visible to this process -- so that the summation order of the
  // gradient, and with it theta and beta, is the same under every partition mode of the device.
  uint32_t want = 256u * 8u * 64u / (wg < 64 ? 64u : wg) * (wg < 64 ? 64u / wg : 1u);
  if (const char* ov = getenv("AMMSB_BETA_SLOTS")) want = (uint32_t)atoi(ov);  // tuning override
  if (want < 64) want = 64;
  if (want > ctx->max_partials) want = ctx->max_partials;
  a.P = span < want ? span : want;
  static const bool force_reg = [] {
    const char* f = getenv("AMMSB_BETA_FORM");
    return f && f[0] == 'r';
  }();
  // LDS-streamed kernels: K == wg * kpt exactly; one wave per slot for wg 64, wg / 64 waves with 16 columns per
  // lane for longer rows (K = 4096: wg 256)
  bool launched = false;
  int rc = AMMSB_OK;
  if (wg == 32 && !force_reg && !force_gen && pi->num_cols % 4 == 0 && (K == 256 || K == 512 || K == 1024)) {
    // the reference's default work-group size on the LDS-streamed one-wave-per-slot kernels (VLane<32>)
    launched = true;
    if (K == 256) rc = launch_grads_lds<4, 1, 32>(ctx, a, s);
    else if (K == 512) rc = launch_grads_lds<8, 1, 32>(ctx, a, s);
    else rc = launch_grads_lds<16, 1, 32>(ctx, a, s);
  } else if (generic) {
    launched = true;
    rc = launch_grads_gen(ctx, a, wg, s);
  } else if (!force_reg && K == wg * (uint32_t)kpt && pi->num_cols % 4 == 0) {
    launched = true;
    if (wg == 64 && kpt == 4) rc = launch_grads_lds<4, 1>(ctx, a, s);
    else if (wg == 64 && kpt == 8) rc = launch_grads_lds<8, 1>(ctx, a, s);
    else if (wg == 64 && kpt == 16) rc = launch_grads_lds<16, 1>(ctx, a, s);
    else if (wg == 128 && kpt == 16) rc = launch_grads_lds<16, 2>(ctx, a, s);
    else if (wg == 256 && kpt == 16) rc = launch_grads_lds<16, 4>(ctx, a, s);
    else launched = false;
  }
  if (rc) return rc;
  if (!launched) {
    AMMSB_DISPATCH_HOT_L(wg, AMMSB_DISPATCH_KPT16(kpt, {
                           int rc2 = launch_grads<L_, KPT_>(ctx, a, s);
                           if (rc2) return rc2;
                         }));
  }
  if (slots_out) *slots_out = a.P;
  if (!sum_rows) {
    AMMSB_LAUNCH_CHECK(ctx);
    return AMMSB_OK;
  }
  if ((2 * K) % 8 == 0 && (reinterpret_cast<uintptr_t>(grads_out) & 15) == 0)
    sum_partials8_kernel<<<2 * K / 8, 256, 0, s>>>(a.partials, a.P, 2 * K, grads_out, desc);
  else
    sum_partials_kernel<<<(2 * K + 15) / 16, 256, 0, s>>>(a.partials, a.P, 2 * K, grads_out, desc);
  AMMSB_LAUNCH_CHECK(ctx);
  return AMMSB_OK;
}

extern "C" int ammsb_beta_grads(ammsb_ctx* ctx, const float* theta, const float* beta, const ammsb_rpm* pi,
                                const ammsb_set* training_set, const uint64_t* edges, uint32_t n_edges,
                                uint32_t edge_begin, uint32_t edge_end, uint32_t wg, float* grads_out, void* stream) {
  return beta_grads_common(ctx, theta, beta, pi, training_set, edges, n_edges, edge_begin, edge_end, wg, grads_out,
                           nullptr, true, nullptr, stream);
}

extern "C" int ammsb_can_fuse_pi_beta(ammsb_ctx* ctx, uint32_t phi_wg, uint32_t beta_wg) {
  return ammsb_beta_can_fuse_pi(ctx, phi_wg, beta_wg) ? 1 : 0;
}

// ammsb_update_pi over nodes[0 .. n_edges] followed by ammsb_beta_grads over the whole mini-batch, as ONE launch (the
// descriptor loop's fused kernel, here with immediate sizes): for the callers outside the loop -- the multi-GPU
// schedule, where every rank then holds the whole gradient and no collective is needed for it.
extern "C" int ammsb_update_pi_beta_grads(ammsb_ctx* ctx, const float* theta, const float* beta, const ammsb_rpm* pi,
                                          float* phi_sum, const float* phi_vec, const uint32_t* nodes,
                                          const ammsb_set* training_set, const uint64_t* edges, uint32_t n_edges,
                                          uint32_t wg, float* grads_out, void* stream) {
  AMMSB_CHECK_ARG(ctx, ctx && phi_sum && phi_vec && nodes && n_edges > 0, "null argument / empty mini-batch");
  if (!beta_fuse_shape(ctx, wg)) {
    snprintf(ctx->err, sizeof ctx->err, "ammsb_update_pi_beta_grads: K=%u at wg=%u is not a shape the fused kernels take "
             "(ammsb_can_fuse_pi_beta)", (unsigned)ctx->params.K, wg);
    return AMMSB_EINVAL;
  }
  const ammsb_pi_fusion fuse = {phi_vec, phi_sum, nodes, nullptr};
  return beta_grads_common(ctx, theta, beta, pi, training_set, edges, n_edges, 0, n_edges, wg, grads_out, nullptr, true,
                           nullptr, stream, &fuse);
}

int ammsb_beta_grads_d(ammsb_ctx* ctx, const float* theta, const float* beta, const ammsb_rpm* pi,
                       const ammsb_set* training_set, const uint64_t* edges, uint32_t n_edges_cap, uint32_t wg,
                       float* grads_out, const ammsb_step_desc* desc, void* stream) {
  AMMSB_CHECK_ARG(ctx, ctx && desc && n_edges_cap > 0, "null descriptor / empty capacity");
  return beta_grads_common(ctx, theta, beta, pi, training_set, edges, n_edges_cap, 0, n_edges_cap, wg, grads_out, desc,
                           true, nullptr, stream);
}

// Run start of the descriptor loop: the gradient constants of the theta the run begins with (every later step's come
// from the theta step in front of it).
int ammsb_theta_coef_d(ammsb_ctx* ctx, const float* theta, const float* beta, void* stream) {
  AMMSB_CHECK_ARG(ctx, ctx && theta && beta, "null argument");
  const uint32_t K = (uint32_t)ctx->params.K;
  theta_coef_kernel<<<(K + 255) / 256, 256, 0, as_stream(stream)>>>(theta, beta, ctx->theta_coef, ctx->theta_sum, K);
  AMMSB_LAUNCH_CHECK(ctx);
  return AMMSB_OK;
}

// The gradient and the theta/beta step of one captured iteration: partial rows, then ONE kernel that sums them into
// grads_out and steps theta (falls back to the separate sum and step kernels for shapes the fused one does not take).
int ammsb_beta_step_d(ammsb_ctx* ctx, float* theta, float* beta, const ammsb_rpm* pi, const ammsb_set* training_set,
                      const uint64_t* edges, uint32_t n_edges_cap, uint32_t wg, float* grads_out, ammsb_seed* seeds,
                      uint32_t flags, const ammsb_step_desc* desc, const ammsb_step_advance* adv,
                      const ammsb_pi_fusion* fuse, unsigned long long* stamps, void* stream) {
  AMMSB_CHECK_ARG(ctx, ctx && desc && adv && seeds && n_edges_cap > 0, "null argument / empty capacity");
  const ammsb_params& p = ctx->params;
  const uint32_t K = (uint32_t)p.K;
  const bool fused = (2 * K) % 8 == 0 && (reinterpret_cast<uintptr_t>(grads_out) & 15) == 0;
  uint32_t slots = 0;
  int rc = beta_grads_common(ctx, theta, beta, pi, training_set, edges, n_edges_cap, 0, n_edges_cap, wg, grads_out, desc,
                             !fused, &slots, stream, fuse, stamps, /*coef_ready=*/true);
  if (rc != AMMSB_OK) return rc;
  if (!fused) return ammsb_update_theta_d(ctx, theta, beta, grads_out, seeds, flags, desc, adv, stream);
  sum_update_theta_kernel<<<2 * K / 8, 256, 0, as_stream(stream)>>>(ctx->grad_partials, slots, 2 * K, grads_out, theta, beta,
                                                                      seeds, p.eta0, p.eta1,
                                                                      (flags & AMMSB_NOISE_OFF) ? 0u : 1u, desc, *adv,
                                                                      ctx->theta_coef, ctx->theta_sum);
  AMMSB_LAUNCH_CHECK(ctx);
  return AMMSB_OK;
}

// Multi-GPU: the R per-rank gradient vectors (all-gathered into [R, cols]) are added in rank order, every rank
// running the same kernel over the same bytes: out[c] = (...((in[0][c] + in[1][c]) + in[2][c]) ...).
__global__ void sum_rows_kernel(const float* in, uint32_t rows, uint32_t cols, float* out) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  float s = in[c];
  for (uint32_t r = 1; r < rows; ++r) s += in[(uint64_t)r * cols + c];
  out[c] = s;
}

extern "C" int ammsb_sum_rows_f32(ammsb_ctx* ctx, const float* in, uint32_t rows, uint32_t cols, float* out,
                                  void* stream) {
  AMMSB_CHECK_ARG(ctx, ctx && in && out && rows > 0 && cols > 0, "bad argument");
  sum_rows_kernel<<<(cols + 255) / 256, 256, 0, as_stream(stream)>>>(in, rows, cols, out);
  AMMSB_LAUNCH_CHECK(ctx);
  return AMMSB_OK;
}

extern "C" int ammsb_theta_sum(ammsb_ctx* ctx, float* out, void* stream) {
  AMMSB_CHECK_ARG(ctx, ctx && out, "null argument");
  AMMSB_HIP(ctx, hipMemcpyAsync(out, ctx->theta_sum, sizeof(float) * ctx->params.K, hipMemcpyDeviceToDevice,
                                as_stream(stream)));
  return AMMSB_OK;
}

extern "C" int ammsb_update_theta(ammsb_ctx* ctx, float* theta, float* beta, const float* grads, uint32_t step_count,
                                  float scale, ammsb_seed* seeds, uint32_t flags, void* stream) {
  AMMSB_CHECK_ARG(ctx, ctx && theta && beta && grads && seeds, "null argument");
  const ammsb_params& p = ctx->params;
  const uint32_t K = (uint32_t)p.K;
  const ammsb_step_advance none = {nullptr, nullptr, nullptr, nullptr, 0u, nullptr, nullptr, nullptr, 0ull, nullptr};
  update_theta_kernel<<<(K + 63) / 64, 64, 0, as_stream(stream)>>>(theta, beta, grads, seeds, K,
                                                                     ammsb_eps_t(&p, step_count), scale, p.eta0,
                                                                     p.eta1, (flags & AMMSB_NOISE_OFF) ? 0u : 1u,
                                                                     nullptr, none, nullptr, nullptr);
  AMMSB_LAUNCH_CHECK(ctx);
  return AMMSB_OK;
}

int ammsb_update_theta_d(ammsb_ctx* ctx, float* theta, float* beta, const float* grads, ammsb_seed* seeds,
                         uint32_t flags, const ammsb_step_desc* desc, const ammsb_step_advance* adv, void* stream) {
  AMMSB_CHECK_ARG(ctx, ctx && theta && beta && grads && seeds && desc, "null argument");
  const ammsb_params& p = ctx->params;
  const uint32_t K = (uint32_t)p.K;
  const ammsb_step_advance none = {nullptr, nullptr, nullptr, nullptr, 0u, nullptr, nullptr, nullptr, 0ull, nullptr};
  update_theta_kernel<<<(K + 63) / 64, 64, 0, as_stream(stream)>>>(theta, beta, grads, seeds, K, 0.0f, 0.0f, p.eta0,
                                                                     p.eta1, (flags & AMMSB_NOISE_OFF) ? 0u : 1u, desc,
                                                                     adv ? *adv : none, ctx->theta_coef, ctx->theta_sum);
  AMMSB_LAUNCH_CHECK(ctx);
  return AMMSB_OK;
}

extern "C" int ammsb_beta_from_theta(ammsb_ctx* ctx, const float* theta, float* beta, void* stream) {
  AMMSB_CHECK_ARG(ctx, ctx && theta && beta, "null argument");
  const uint32_t K = (uint32_t)ctx->params.K;
  beta_from_theta_kernel<<<(K + 255) / 256, 256, 0, as_stream(stream)>>>(theta, beta, K);
  AMMSB_LAUNCH_CHECK(ctx);
  return AMMSB_OK;
}
