// update_phi / update_pi for gfx950.
//
// Replaces PhiUpdater::operator() (mcmc/phi.cc:728-763) and the work-group kernels
// update_phi / update_phi_for_nodeWG (phi.cc:214-302) and update_pi (phi.cc:177-197).
//
// Mapping.  The reference's OpenCL work-group of L = phi_wg_size work-items becomes a "virtual
// group" (ammsb_dev.h Group<L>): lane l owns columns k = l, l+L, ... of the K-vector -- the same
// ownership the reference uses, so each lane's WG_SUM partial, its RNG stream (seeds[g*L+l]) and
// its draw order (ascending k) are the reference's.  Everything a lane owns lives in registers:
// pi_a, grads, probs and DEPTH neighbour rows in flight (KPT = ceil(K/L) floats each).
//
// HBM traffic per mini-batch node: (n+1) pi rows read (random 4K-byte rows at K=1024) + one
// phi_vec row written + n ids + <= 2n 32-byte cuckoo bins: 4K(n+2) + 68n + 8 bytes.  Rows are
// streamed with lane-contiguous 256-byte wave loads; DEPTH rows per wave are kept in flight so that
// ~12 waves/CU cover the HBM latency.  The n cuckoo probes of a node are issued by n lanes at once
// (one latency instead of n dependent ones, as the reference's lane-0-style lookup would cost).
#pragma clang fp contract(off)

#include <stdlib.h>

#include "ammsb_ctx.h"
#include "ammsb_dev.h"
#include "ammsb_step.h"

#include <type_traits>

using namespace ammsb;

#ifdef AMMSB_PHI_TRACE
// development aid (tools/phi_trace.sh): shader-clock stamps of block 0's first node, read back with ammsb_debug_trace
__device__ unsigned long long g_phi_trace[256];
#define PHI_TRACE(slot) do { if (blockIdx.x == 0 && threadIdx.x == 0 && (slot) < 256) g_phi_trace[(slot)] = __builtin_readcyclecounter(); } while (0)
extern "C" int ammsb_debug_trace(unsigned long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phi_trace), sizeof(unsigned long long) * (n < 256 ? n : 256)) == hipSuccess ? 0 : -2;
}
#define AMMSB_PHI_BLK_CAP 65536
// per-block occupancy record: [block][0..3] = shader clock at start / end, 100 MHz wall clock at start / end,
// [4] = HW_ID | XCC_ID << 32 (which CU / SIMD / wave slot the block ran on)
__device__ unsigned long long g_phi_blk[AMMSB_PHI_BLK_CAP * 5];
#define PHI_BLK(end) do { if (threadIdx.x == 0 && blockIdx.x < AMMSB_PHI_BLK_CAP) { \
    g_phi_blk[blockIdx.x * 5 + (end)] = __builtin_readcyclecounter(); \
    g_phi_blk[blockIdx.x * 5 + 2 + (end)] = __builtin_amdgcn_s_memrealtime(); \
    if (!(end)) g_phi_blk[blockIdx.x * 5 + 4] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | \
                                               ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32); } } while (0)
extern "C" int ammsb_debug_blocks(unsigned long long* out, int n_blocks) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phi_blk), sizeof(unsigned long long) * 5 * (n_blocks < AMMSB_PHI_BLK_CAP ? n_blocks : AMMSB_PHI_BLK_CAP)) == hipSuccess ? 0 : -2;
}
#else
#define PHI_TRACE(slot) do { } while (0)
#define PHI_BLK(end) do { } while (0)
#endif

#ifndef AMMSB_PHI_LDS3_DEFAULT
#define AMMSB_PHI_LDS3_DEFAULT false
#endif

namespace {

struct PhiArgs {
  const float* beta;
  ammsb_rpm pi;
  const float* phi_sum;
  DevSet set;  // (ammsb_dev.h: the descriptor + the modulo magic)
  const uint32_t* nodes;
  const uint32_t* neighbors;
  ammsb_seed* seeds;
  float* phi_vec;
  uint32_t n_nodes, G, group_begin, group_end;
  uint32_t K, n;
  float eps_t, alpha, epsilon, Nn;
  uint32_t noise_on;
  uint32_t rows_nt;  // neighbour rows requested with the non-temporal hint (a pi that does not fit the last-level cache)
  const ammsb_step_desc* desc;  // non-null (captured graph): n_nodes and eps_t come from here, all groups run
  unsigned long long* stamps;   // optional (with desc): block 0 notes the device time at which it starts
};

// (time stamps: note_stamp in ammsb_step.h -- block 0 of update_phi notes when it starts, block 0 of the kernel after
// it when IT starts)

// the per-iteration scalars, from the descriptor when there is one (block-uniform scalar loads)
struct PhiStep {
  uint32_t n_nodes, G, group_end;
  float eps_t;
};
__device__ __forceinline__ PhiStep phi_step(const PhiArgs& a) {
  PhiStep st = {a.n_nodes, a.G, a.group_end, a.eps_t};
  if (a.desc) {
    st.n_nodes = a.desc->n_nodes;
    st.G = st.n_nodes < AMMSB_MAX_GROUPS ? st.n_nodes : AMMSB_MAX_GROUPS;
    st.group_end = st.G;
    st.eps_t = a.desc->eps_t;
  }
  return st;
}

// FULL: K == L * KPT, so no column guard is needed anywhere.  Loads are always unconditional (row
// and column indices are clamped instead of predicated): a predicated load turns into a branch plus a
// full vmcnt(0) drain per element, which serialises the row stream.
template <int L, int KPT, int DEPTH, bool FULL, bool ONE = false>
__global__ __launch_bounds__(Group<L>::BLOCK) void update_phi_kernel(const PhiArgs a) {
  if constexpr (ONE) __builtin_assume(a.pi.num_blocks == 1);  // (see update_phi_lds2_kernel; short rows at wg 32 / 64 only)
  using Grp = Group<L>;
  extern __shared__ uint32_t s_nb_all[];  // [PER_BLOCK][n]: neighbour id | link bit << 31
  __shared__ ZigTables zig;
  __shared__ float aux[Grp::AUX];

  const int l = Grp::lane();
  const int slot = Grp::slot();
  uint32_t* s_nb = s_nb_all + slot * a.n;
  const PhiStep st = phi_step(a);
  if (st.n_nodes == 0) return;  // (uniform) a skipped step
  note_stamp(a.stamps, a.desc, AMMSB_STAMP_PHI);
  const uint32_t g = a.group_begin + blockIdx.x * Grp::PER_BLOCK + slot;
  const bool live = g < st.group_end;
  const uint32_t K = a.K, n = a.n;
  const float EPS = a.epsilon;

  if (a.noise_on) zig_load(&zig);

  // column owned by (lane, j), its clamped form for addressing, and whether it exists
  auto col = [&](int j) -> uint32_t { return l + j * L; };
  auto ccol = [&](int j) -> uint32_t {
    const uint32_t k = l + j * L;
    return FULL ? k : (k < K ? k : K - 1);
  };
  auto has = [&](int j) -> bool { return FULL || (uint32_t)(l + j * L) < K; };

  // per-lane constants: f = beta_k - EPSILON for a link, its exact negation for a non-link
  float bf[KPT];
  bool beta_safe = true;  // every beta_k this lane owns lies in [EPSILON, 1 - 2^-20] (exact-division fast path)
#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    const float b = a.beta[2 * ccol(j) + 1];
    bf[j] = b - EPS;
    beta_safe = beta_safe && in_range(b, EPS, kBetaHi);
  }

  ammsb_seed rs = {0, 0};
  if (live && a.noise_on) rs = a.seeds[(uint64_t)g * L + l];  // rand->base_[GET_GLOBAL_ID()], phi.cc:291

  const uint32_t trips = (st.n_nodes + st.G - 1) / st.G;  // uniform over the block
  int phase = 0;
  for (uint32_t t = 0; t < trips; ++t) {
    const uint64_t i_raw = (uint64_t)g + (uint64_t)t * st.G;  // node index handled by this group
    const bool on = live && i_raw < st.n_nodes;
    if constexpr (Grp::PER_BLOCK == 1) {
      if (!on) continue;  // block-uniform: every thread skips the barriers below together
    }
    const uint64_t i = on ? i_raw : 0;  // idle sub-wave groups shadow node 0 and store nothing
    const uint32_t node = a.nodes[i];

    // ---- stage neighbour ids and the n link bits (one cuckoo probe per lane)
    __syncthreads();
    for (uint32_t q = l; q < n; q += L) {
      const uint32_t nb = a.neighbors[i * n + q];
      const bool y = set_has(a.set, make_edge(node, nb));
      s_nb[q] = nb | (y ? 0x80000000u : 0u);
    }
    __syncthreads();

    const float phi_sum = a.phi_sum[node];
    const float inv_phi_sum = 1.0f / phi_sum;
    const float* row_a = rpm_row(a.pi, node);
    // den[j] = pi_a[j] * phi_sum is the divisor of every neighbour's second division: refine its
    // reciprocal once per node (see ammsb_dev.h "exact division")
    float pi_a[KPT], grads[KPT], rden[KPT];  // den itself is recomputed where needed (one multiply)
    bool node_safe = beta_safe && in_range(phi_sum, kPhiSumLo, kPhiSumHi);
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      const float x = row_a[ccol(j)];
      pi_a[j] = has(j) ? x : 0.0f;
      grads[j] = 0.0f;
      const float den = pi_a[j] * phi_sum;
      rden[j] = refined_rcp(den);
      node_safe = node_safe && (in_range(den, kDenLo, kDenHi) || !has(j));
    }

    float buf[DEPTH][KPT];
    auto load_row = [&](float (&dst)[KPT], uint32_t q) {
      uint32_t w = s_nb[q] & 0x7fffffffu;
      if constexpr (L >= 64) w = __builtin_amdgcn_readfirstlane(w);  // uniform per wave: scalar row base
      const float* row = rpm_row(a.pi, w);
#pragma unroll
      for (int j = 0; j < KPT; ++j) dst[j] = row[ccol(j)];
    };
    auto consume = [&](float (&pin)[KPT], uint32_t q) {
      bool y = (s_nb[q] >> 31) != 0;
      if constexpr (L >= 64) y = __builtin_amdgcn_readfirstlane((int)y) != 0;  // wave-uniform branch below
      const float e = y ? EPS : 1.0f - EPS;
      float partial = 0.0f;
      float lo = 1.0f;  // smallest |probs| of this lane (columns that exist)
      // phi.cc:241-253; pin[] is overwritten with probs[].  pin * (EPS - beta) + e == e - pin * (beta - EPS)
      // bit for bit, so the non-link case subtracts instead of selecting a negated factor.
      if (y) {
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
          float tt = pin[j] * bf[j];
          tt = tt + e;
          pin[j] = pi_a[j] * tt;
        }
      } else {
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
          const float tt0 = pin[j] * bf[j];
          const float tt = e - tt0;
          pin[j] = pi_a[j] * tt;
        }
      }
#pragma unroll
      for (int j = 0; j < KPT; ++j) {
        partial += pin[j];  // 0 for a column beyond K (pi_a = 0)
        lo = fminf(lo, has(j) ? fabsf(pin[j]) : 1.0f);
      }
      const float probs_sum = Grp::sum(partial, aux, phase);  // phi.cc:254-257
      // phi.cc:259-263: grads += (probs / probs_sum) / (pi * phi_sum) - 1 / phi_sum
      if (node_safe && lo >= kProbsLo && in_range(probs_sum, kPsumLo, kPsumHi)) {
        const float rps = refined_rcp(probs_sum);
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
          float qv = div_with_rcp(pin[j], probs_sum, rps);
          qv = div_with_rcp(qv, pi_a[j] * phi_sum, rden[j]);
          grads[j] += qv - inv_phi_sum;
        }
      } else {
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
          float qv = pin[j] / probs_sum;
          qv = qv / (pi_a[j] * phi_sum);
          grads[j] += qv - inv_phi_sum;
        }
      }
    };

    // Short rows (at most two columns per lane): what a row costs is not its bytes but the chain of dependent
    // instructions behind them -- two WG_SUM trees, a reciprocal, two divisions -- so FOUR rows are taken through
    // that chain together (independent chains interleave), the next four already requested.  Per row the arithmetic
    // is consume()'s (the non-link sign as an exact multiplication by -1, as in update_phi_lds2_kernel) and the
    // gradient accumulates row by row in order: bit-identical.
    bool batched = false;
    if constexpr (KPT <= 2 && DEPTH == 4) {
      if (n % 4 == 0) {
        batched = true;
        float nxt[4][KPT];
        auto consume4 = [&](float (&pin)[4][KPT], uint32_t q0) {
          float ee[4], sg[4], part[4], lo[4], psum[4];
          bool fast[4], all_fast = true;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool y = (s_nb[q0 + r] >> 31) != 0;
            ee[r] = y ? EPS : 1.0f - EPS;
            sg[r] = y ? 1.0f : -1.0f;
            part[r] = 0.0f;
            lo[r] = 1.0f;
#pragma unroll
            for (int j = 0; j < KPT; ++j) {
              const float tt = (pin[r][j] * bf[j]) * sg[r] + ee[r];
              pin[r][j] = pi_a[j] * tt;
              part[r] += pin[r][j];
              lo[r] = fminf(lo[r], has(j) ? fabsf(pin[r][j]) : 1.0f);
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            psum[r] = Grp::sum(part[r], aux, phase);
            fast[r] = node_safe && lo[r] >= kProbsLo && in_range(psum[r], kPsumLo, kPsumHi);
            all_fast = all_fast && fast[r];
          }
          if (all_fast) {
            float rps[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) rps[r] = refined_rcp(psum[r]);
#pragma unroll
            for (int j = 0; j < KPT; ++j) {
              float qv[4];
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                qv[r] = div_with_rcp(pin[r][j], psum[r], rps[r]);
                qv[r] = div_with_rcp(qv[r], pi_a[j] * phi_sum, rden[j]);
              }
#pragma unroll
              for (int r = 0; r < 4; ++r) grads[j] += qv[r] - inv_phi_sum;
            }
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if (fast[r]) {
                const float rps = refined_rcp(psum[r]);
#pragma unroll
                for (int j = 0; j < KPT; ++j) {
                  float qv = div_with_rcp(pin[r][j], psum[r], rps);
                  qv = div_with_rcp(qv, pi_a[j] * phi_sum, rden[j]);
                  grads[j] += qv - inv_phi_sum;
                }
              } else {
#pragma unroll
                for (int j = 0; j < KPT; ++j) {
                  float qv = pin[r][j] / psum[r];
                  qv = qv / (pi_a[j] * phi_sum);
                  grads[j] += qv - inv_phi_sum;
                }
              }
            }
          }
        };
        if (KPT == 1 && n <= 32) {
          // one column per lane, at most 32 neighbours (the reference's default, main.cc:58): ALL rows are requested before the first is
          // consumed -- 32 KPT registers -- so a node costs one row round trip instead of one per group of four
          // (C1: a row group's arithmetic is shorter than the latency of the next group's rows, the loop was a chain
          // of eight round trips).  Row indices are clamped, not predicated (see FULL above).
          float all[8][4][KPT];
#pragma unroll
          for (int gq = 0; gq < 8; ++gq)
#pragma unroll
            for (int d = 0; d < 4; ++d) {
              const uint32_t q = (uint32_t)(4 * gq + d);
              load_row(all[gq][d], q < n ? q : n - 1);
            }
#pragma unroll
          for (int gq = 0; gq < 8; ++gq)
            if ((uint32_t)(4 * gq) < n) consume4(all[gq], 4 * gq);
        } else {
#pragma unroll
        for (int d = 0; d < 4; ++d) load_row(buf[d], d);
        for (uint32_t q0 = 0; q0 < n; q0 += 4) {
          if (q0 + 4 < n) {
#pragma unroll
            for (int d = 0; d < 4; ++d) load_row(nxt[d], q0 + 4 + d);
          }
          consume4(buf, q0);
          if (q0 + 4 < n) {
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
              for (int j = 0; j < KPT; ++j) buf[d][j] = nxt[d][j];
          }
        }
        }
      }
    }

    // Software pipeline, DEPTH-1 rows ahead.  The steady state is straight-line code (no branch around
    // a load): hipcc's wait-count insertion falls back to vmcnt(0) -- draining the rows just requested
    // -- when paths with different numbers of outstanding loads meet.  n = G*DEPTH + r rows: the first
    // G-1 groups request and consume DEPTH rows each, the last group requests only its last row, the r
    // left-over rows are handled one at a time.
    const uint32_t groups = batched ? 0 : n / DEPTH;
    if (groups > 0) {
#pragma unroll
      for (int d = 0; d < DEPTH - 1; ++d) load_row(buf[d], d);
      for (uint32_t gq = 0; gq + 1 < groups; ++gq) {
        const uint32_t q0 = gq * DEPTH;
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
          load_row(buf[(d + DEPTH - 1) % DEPTH], q0 + d + DEPTH - 1);
          consume(buf[d], q0 + d);
        }
      }
      const uint32_t q0 = (groups - 1) * DEPTH;
      load_row(buf[DEPTH - 1], q0 + DEPTH - 1);
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) consume(buf[d], q0 + d);
    }
    for (uint32_t q = batched ? n : groups * DEPTH; q < n; ++q) {
      load_row(buf[0], q);
      consume(buf[0], q);
    }

    // ---- SGLD step, phi.cc:265-274; lane l draws for k = l, l+L, ... in ascending order
    if (on) {
      float* out = a.phi_vec + i * K;
      const float half = st.eps_t / 2;
#pragma unroll
      for (int j = 0; j < KPT; ++j) {
        if (has(j)) {
          const float noise = a.noise_on ? rng_normal(rs, &zig) : 1.0f;
          const float phi_k = pi_a[j] * phi_sum;
          const float ng = a.Nn * grads[j];
          float in = a.alpha - phi_k;
          in = in + ng;
          const float drift = half * in;
          const float aa = phi_k + drift;
          const float ep = st.eps_t * phi_k;
          const float sq = sqrtf(ep);
          const float bb = sq * noise;
          const float v = fabsf(aa + bb);
          out[col(j)] = v > 1e-24f ? v : 1e-24f;
        }
      }
    }
  }
  if (live && a.noise_on) a.seeds[(uint64_t)g * L + l] = rs;
}

// ---------------------------------------------------------------------------------------------
// LDS-streamed form of update_phi for L = 64 and K = 64 * KPT (KPT in {4, 8, 16, 32}).
//
// The register-pipelined kernel above keeps DEPTH neighbour rows per wave in VGPRs; with the
// persistent per-column state that costs ~250 VGPRs at K = 1024, i.e. two waves per SIMD, and a wave
// that is computing cannot cover its partner's memory stalls: measured VALU busy 45 %, HBM 51 %.
// Here neighbour rows never enter VGPRs wholesale: each wave owns a two-slot LDS ring that is filled
// by LDS-DMA (global_load_lds_dwordx4: 4 x 1 KiB pieces per 4 KiB row, per-lane source address,
// wave-uniform LDS destination) one row ahead, reads a row with ds_read2st64_b32 (lane l gets columns
// l + 64 j -- the reference's lane ownership, hence its WG_SUM order), writes probs[] back in place
// and re-reads it for the gradient pass.  ~100 VGPRs -> 4 waves per SIMD, 16 rows in flight per CU.
// Arithmetic, operation order and RNG consumption are those of the register kernel, bit for bit.

// minimum waves per SIMD the register allocator must allow for the short-row kernels (K = 256 / 512): their
// per-row work is latency-bound, so they want more resident waves than the 4 KiB-row kernel (A/B: tools/gpu_exp.sh)
#ifndef AMMSB_PHI_WPE4
#define AMMSB_PHI_WPE4 3
#endif
#ifndef AMMSB_PHI_WPE8
#define AMMSB_PHI_WPE8 3
#endif
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

// D = ring depth (power of two): rows q+1 .. q+D-1 are in flight while row q is reduced.  Two slots are right for
// 4 KiB rows (K = 1024: LDS, not latency, limits the waves per CU); short rows (K = 256: one 1 KiB piece per
// row) need more rows in flight per wave to keep enough bytes on their way to each CU.
// NB = nodes per block (only with W == 1): NB independent one-wave nodes share the block's ziggurat tables, nothing
// else -- 2 x (3 x 4 KiB + 128 B) + 1.5 KiB = 26 368 B puts six such blocks = TWELVE waves on a CU where the one-node
// block (13 956 B) fits eleven times.  The waves never wait for each other after the table load (wave-local LDS
// ordering instead of block barriers: their node counts may differ).
// VL = 32 (only with W == 1): the reference work-group size is 32, not 64 -- the node still gets the whole wave and
// every lane its 64-strided columns; the WG_SUM chain / tree and the stream-to-column map follow the 32 virtual lanes
// (VLane<32>, ammsb_dev.h).  K = 1024 at the reference's default phi_wg_size (main.cc:61) takes this form.
template <int KPT, int W, int D = 2, int NB = 1, int VL = 64, bool ONE = false>
__global__ __launch_bounds__(64 * W * NB) __attribute__((amdgpu_waves_per_eu(KPT <= 4 ? AMMSB_PHI_WPE4 : KPT <= 8 ? AMMSB_PHI_WPE8 : KPT <= 16 ? 3 : 2))) void update_phi_lds_kernel(const PhiArgs a) {
  if constexpr (ONE) __builtin_assume(a.pi.num_blocks == 1);  // (see update_phi_lds2_kernel)
  static_assert(NB == 1 || W == 1, "several nodes per block only for one-wave nodes");
  static_assert(VL == 64 || W == 1, "virtual half-wave lanes only for one-wave nodes");
  using VLn = VLane<VL>;
  constexpr int LV = VL == 64 ? 64 * W : VL;  // the reference work-group size: streams per node
  constexpr int KV = KPT * VLn::PER;          // columns (= normals) per virtual lane
  // L = 64 W lanes per node: wave wv owns columns 64 wv + ln + L j, i.e. KPT chunks of 64 consecutive floats per
  // row.  Each wave runs the single-wave pipeline on its own slice (own ring, own waits); the only cross-wave
  // step is the WG_SUM of a neighbour's probs: one LDS exchange and one barrier per row.
  constexpr int L = 64 * W;
  constexpr int KW = 64 * KPT;     // floats of a row that one wave handles
  constexpr int K = L * KPT;
  constexpr int PIECES = KPT / 4;  // 1 KiB LDS-DMA pieces per wave and row (4 chunks of 256 B each)
  static_assert(D >= 2 && (D & (D - 1)) == 0 && (D - 1) * PIECES <= 63, "ring depth");
  extern __shared__ __align__(16) char smem[];  // per wave: [D][KW] ring, [KW] normals; then [n] u32 (id | link bit)
  __shared__ ZigTables zig;
  __shared__ float xsum[W > 1 ? 2 * L : 1];  // double-buffered lane partials of the cross-wave sum
  // tid = thread within its node's group of 64 W lanes; nb = which of the block's NB nodes
  const int tid = NB == 1 ? (int)threadIdx.x : (int)(threadIdx.x & 63), nb = NB == 1 ? 0 : (int)(threadIdx.x >> 6);
  const int wv = W == 1 ? 0 : tid >> 6, ln = W == 1 ? tid : tid & 63;
  const uint32_t n = a.n;
  char* node_smem = smem;
  if constexpr (NB > 1) node_smem += nb * (W * (D + 1) * KW * sizeof(float) + ((n * sizeof(uint32_t) + 15) & ~15u));
  char* wave_smem = node_smem + wv * ((D + 1) * KW * sizeof(float));
  float* ring = reinterpret_cast<float*>(wave_smem);
  float* s_noise = ring + D * KW;
  uint32_t* s_nb = reinterpret_cast<uint32_t*>(node_smem + W * (D + 1) * KW * sizeof(float));
  // orders this node's LDS traffic: a block barrier for one node per block; with several independent one-wave nodes a
  // wave-local wait (one wave's LDS operations complete in order)
  auto group_sync = [&]() {
    if constexpr (NB == 1) {
      __syncthreads();
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    }
  };

  const PhiStep st = phi_step(a);
  note_stamp(a.stamps, a.desc, AMMSB_STAMP_PHI);
  if constexpr (NB > 1) {  // every wave of the block helps loading the shared tables before any of them may leave
    if (a.noise_on) zig_load(&zig);
    __syncthreads();
  }
  const uint32_t g = a.group_begin + blockIdx.x * NB + nb;
  if (g >= st.group_end) return;  // uniform per node group
  PHI_BLK(0);
  const float EPS = a.epsilon;
  if constexpr (NB == 1) {
    if (a.noise_on) zig_load(&zig);
  }

  constexpr int HP = KPT / 2;  // column pairs per lane: pair p = columns tid + L (2p), tid + L (2p + 1)
  f32x2 bf[HP];
  bool beta_safe = true;
#pragma unroll
  for (int p = 0; p < HP; ++p) {
    const float b0 = a.beta[2 * (tid + 2 * L * p) + 1];
    const float b1 = a.beta[2 * (tid + 2 * L * p + L) + 1];
    bf[p] = f32x2{b0 - EPS, b1 - EPS};
    beta_safe = beta_safe && in_range(b0, EPS, kBetaHi) && in_range(b1, EPS, kBetaHi);
  }
  ammsb_seed rs = {0, 0};
  if (a.noise_on) rs = a.seeds[(uint64_t)g * LV + VLn::vlane(tid)];  // (VL = 32: both halves carry the same stream)
  // the virtual lane's normal number j scales the noise slot of physical column j / PER in the lane that owns it
  auto draw_normal = [&](uint32_t j) {
    if constexpr (VL == 64) {
      s_noise[ln + 64 * j] = s_noise[ln + 64 * j] * rng_normal(rs, &zig);
    } else {
      const float z = rng_normal(rs, &zig);
      if (VLn::keeps(tid, j)) s_noise[ln + 64 * (j / VLn::PER)] = s_noise[ln + 64 * (j / VLn::PER)] * z;
    }
  };

  // request this wave's slice of neighbour row q into ring slot `slot`: piece t carries chunks j = 4t .. 4t+3
  // (16 lanes x 16 B each), so the slice lands as [j][64] and lane ln reads column tid + L j at [j * 64 + ln]
  auto request = [&](uint32_t q, uint32_t slot) {
    // aux 2 = nt (non-temporal): a neighbour row is read once per launch; keeping it out of the caches' way
    // measured -4 % per launch (same-box A/B, five alternations)
    const uint32_t nbr = __builtin_amdgcn_readfirstlane(s_nb[q] & 0x7fffffffu);
    const float* src = rpm_row(a.pi, nbr) + (W == 1 ? 4 * tid : L * (ln >> 4) + 64 * wv + 4 * (ln & 15));
    char* dst = wave_smem + slot * (KW * sizeof(float));
    if (a.rows_nt) {
#pragma unroll
      for (int t = 0; t < PIECES; ++t)
        __builtin_amdgcn_global_load_lds((glb_void_t*)(src + 4 * L * t), (lds_void_t*)(dst + 1024 * t), 16, 0, 2);
    } else {
#pragma unroll
      for (int t = 0; t < PIECES; ++t)
        __builtin_amdgcn_global_load_lds((glb_void_t*)(src + 4 * L * t), (lds_void_t*)(dst + 1024 * t), 16, 0, 0);
    }
  };

  // WG_SUM over L lanes (sum.cc:20-29): levels L/2 .. 64 fold wave i + s onto wave i, then the in-wave tree
  int phase = 0;
  auto group_sum = [&](float v) -> float {
    if constexpr (W == 1) {
      return VLn::tree(v);
    } else {
      float* x = xsum + phase * L;
      phase ^= 1;
      x[tid] = v;
      __syncthreads();
      float part[W];
#pragma unroll
      for (int i = 0; i < W; ++i) part[i] = x[64 * i + ln];
#pragma unroll
      for (int st = W / 2; st >= 1; st >>= 1) {
#pragma unroll
        for (int i = 0; i < st; ++i) part[i] += part[i + st];
      }
      return Group<64>::wave_tree64(part[0]);
    }
  };

  for (uint64_t i = g; i < st.n_nodes; i += st.G) {
    const uint32_t node = a.nodes[i];
    group_sync();  // orders the LDS traffic of consecutive nodes
    for (uint32_t q = tid; q < n; q += L) {
      const uint32_t nbq = a.neighbors[i * n + q];
      const bool y = set_has(a.set, make_edge(node, nbq));
      s_nb[q] = nbq | (y ? 0x80000000u : 0u);
    }
    group_sync();

    const float phi_sum = a.phi_sum[node];
    const float inv_phi_sum = 1.0f / phi_sum;
    const float* row_a = rpm_row(a.pi, node);
    f32x2 pi_a[HP], grads[HP], rden[HP];
    bool node_safe = beta_safe && in_range(phi_sum, kPhiSumLo, kPhiSumHi);
#pragma unroll
    for (int p = 0; p < HP; ++p)
      pi_a[p] = f32x2{__builtin_nontemporal_load(row_a + tid + 2 * L * p), __builtin_nontemporal_load(row_a + tid + 2 * L * p + L)};
#pragma unroll
    for (uint32_t r = 0; r < (uint32_t)(D - 1); ++r)
      if (r < n) request(r, r);  // the first rows' flight overlaps the per-node set-up below
#pragma unroll
    for (int p = 0; p < HP; ++p) {
      grads[p] = f32x2{0.0f, 0.0f};
      const f32x2 den = pi_a[p] * phi_sum;
      rden[p] = f32x2{exact_rcp(den.x), exact_rcp(den.y)};
      node_safe = node_safe && in_range(den.x, kDenLo, kDenHi) && in_range(den.y, kDenLo, kDenHi);
      // sqrt(eps_t * phi_k) of the SGLD step does not depend on the gradient: computed here, under the first
      // row's latency, and parked in the noise slot, where the loop multiplies the normal in
      const f32x2 ep = den * st.eps_t;
      s_noise[ln + 128 * p] = sqrtf(ep.x);
      s_noise[ln + 128 * p + 64] = sqrtf(ep.y);
    }

    for (uint32_t q = 0; q < n; ++q) {
      const uint32_t slot = q & (D - 1);
      float* row = ring + slot * KW;
      // every LDS read of the slot of row q-1 has been consumed; refill it with row q+D-1
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (q + (D - 1) < n) {
        request(q + (D - 1), (q + (D - 1)) & (D - 1));
        // one of the lane's KPT normals per iteration, drawn while row q is still on its way (stream order is
        // the ascending column order of the SGLD step below)
        if (a.noise_on && q < (uint32_t)KV) draw_normal(q);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * PIECES) : "memory");  // row q landed, D-1 rows in flight
      } else {
        if constexpr (D > 2) {  // (D == 2 keeps the round-1 instruction stream: the last row draws no normal)
          if (a.noise_on && q < (uint32_t)KV) draw_normal(q);
        }
        // the tail: rows q+1 .. n-1 (fewer than D-1) are still in flight
        const uint32_t rem = n - 1 - q;
        if constexpr (D == 2) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if constexpr (D == 4) {
          if (rem == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PIECES) : "memory");
          else if (rem == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 * PIECES) : "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
          if (rem >= 6) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * PIECES) : "memory");
          else if (rem == 5) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * PIECES) : "memory");
          else if (rem == 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * PIECES) : "memory");
          else if (rem == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PIECES) : "memory");
          else if (rem == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PIECES) : "memory");
          else if (rem == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 * PIECES) : "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
      }
      bool y = (s_nb[q] >> 31) != 0;
      y = __builtin_amdgcn_readfirstlane((int)y) != 0;
      const float e = y ? EPS : 1.0f - EPS;

      // pass 1 (phi.cc:241-253): probs[] in place of the row, lane partial in ascending column order.
      // pin * (EPS - beta) + e == e - pin * (beta - EPS) bit for bit; the two wave-uniform cases are separate
      // code so that the sign rides on the packed add's source modifier.
      float partial = 0.0f, lo = 1.0f;
      auto pass1 = [&](auto link) {
#pragma unroll
        for (int p = 0; p < HP; ++p) {
          const f32x2 pin = f32x2{row[ln + 128 * p], row[ln + 128 * p + 64]};
          const f32x2 tt0 = pin * bf[p];
          const f32x2 tt = decltype(link)::value ? tt0 + e : e - tt0;
          const f32x2 pr = pi_a[p] * tt;
          row[ln + 128 * p] = pr.x;
          row[ln + 128 * p + 64] = pr.y;
          VLn::chain(partial, pr.x);
          VLn::chain(partial, pr.y);
          lo = fminf(fminf(lo, fabsf(pr.x)), fabsf(pr.y));
        }
      };
      if (y) pass1(std::true_type{});
      else pass1(std::false_type{});
      const float probs_sum = group_sum(partial);  // phi.cc:254-257

      // pass 2 (phi.cc:259-263): grads += (probs / probs_sum) / (pi * phi_sum) - 1 / phi_sum
      if (node_safe && lo >= kProbsLo && in_range(probs_sum, kPsumLo, kPsumHi)) {
        const float rps = exact_rcp(probs_sum);
        float ps = phi_sum;
        asm volatile("" : "+v"(ps));  // keeps pi_a * phi_sum from being hoisted into KPT more registers
        const f32x2 psum2 = f32x2{probs_sum, probs_sum}, rps2 = f32x2{rps, rps};
#pragma unroll
        for (int p = 0; p < HP; ++p) {
          const f32x2 pr = f32x2{row[ln + 128 * p], row[ln + 128 * p + 64]};
          f32x2 qv = div_exact3(pr, psum2, rps2);
          qv = div_exact3(qv, pi_a[p] * ps, rden[p]);
          grads[p] += qv - inv_phi_sum;
        }
      } else {
#pragma unroll
        for (int p = 0; p < HP; ++p) {
          const f32x2 den = pi_a[p] * phi_sum;
          float q0 = row[ln + 128 * p] / probs_sum;
          float q1 = row[ln + 128 * p + 64] / probs_sum;
          q0 = q0 / den.x;
          q1 = q1 / den.y;
          grads[p] += f32x2{q0 - inv_phi_sum, q1 - inv_phi_sum};
        }
      }
    }

    // normals the loop did not get to (n - 1 < KPT): one rolled loop, a single copy of the ziggurat code
    if (a.noise_on) {
#pragma unroll 1
      for (uint32_t j = D > 2 ? n : (n > 0 ? n - 1 : 0); j < (uint32_t)KV; ++j) draw_normal(j);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // SGLD step, phi.cc:265-274
    float* out = a.phi_vec + i * K;
    const float half = st.eps_t / 2;
#pragma unroll
    for (int p = 0; p < HP; ++p) {
      const f32x2 bb = f32x2{s_noise[ln + 128 * p], s_noise[ln + 128 * p + 64]};  // sqrt(eps_t * phi_k) * noise
      const f32x2 phi_k = pi_a[p] * phi_sum;
      const f32x2 ng = grads[p] * a.Nn;
      f32x2 in = a.alpha - phi_k;
      in = in + ng;
      const f32x2 drift = in * half;
      const f32x2 aa = phi_k + drift;
      const f32x2 s2 = aa + bb;
      const float v0 = fabsf(s2.x), v1 = fabsf(s2.y);
      __builtin_nontemporal_store(v0 > 1e-24f ? v0 : 1e-24f, out + tid + 2 * L * p);
      __builtin_nontemporal_store(v1 > 1e-24f ? v1 : 1e-24f, out + tid + 2 * L * p + L);
    }
  }
  if (a.noise_on && tid < LV) a.seeds[(uint64_t)g * LV + tid] = rs;
  PHI_BLK(1);
}

// ---------------------------------------------------------------------------------------------------------
// K = 1024 (KPT = 16), one wave per node, THREE ring slots in the LDS of two slots + the noise buffer (round 4).
//
// update_phi_lds_kernel<16, 1, 2> keeps per wave two 4 KiB ring slots and a 4 KiB buffer of sqrt(eps_t phi_k) * normal_k:
// eleven one-wave blocks per CU (LDS), ONE row in flight per wave while another is reduced -- 44 KB in flight per CU
// against a loaded HBM latency of ~2 us, and SQ counters that show the waves parked on s_waitcnt 58 % of their life.
// A third slot at the same LDS footprint needs the noise buffer gone.  Here the lane's normals go where the node's
// output row will go: draw j is stored (raw) to phi_vec[i][ln + 64 j], and after the last neighbour row the ring fetches
// that 4 KiB row back like one more row -- it has exactly a pi row's layout -- so the noise costs no LDS, no registers,
// and its latency is covered like a row's.  sqrt(eps_t phi_k) is evaluated in the SGLD step instead of the prologue and
// multiplied with the normal there: the same product (a * b == b * a), bit-identical results, same stream order.
// Two rows in flight per wave at eleven waves per CU.  Needs n >= KV + 2 (the loop draws one normal per row, the noise
// row is requested two rows before the end); the launcher falls back to the two-slot kernel otherwise.
template <int KPT, int VL = 64, bool ONE = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3))) void update_phi_lds3_kernel(const PhiArgs a) {
  if constexpr (ONE) __builtin_assume(a.pi.num_blocks == 1);
  using VLn = VLane<VL>;
  constexpr int D = 3;
  constexpr int KV = KPT * VLn::PER;  // normals per virtual lane
  constexpr int L = 64, KW = 64 * KPT, K = L * KPT, PIECES = KPT / 4, HP = KPT / 2;
  extern __shared__ __align__(16) char smem[];  // [D][KW] ring, then [n] u32 (id | link bit)
  __shared__ ZigTables zig;
  const int tid = threadIdx.x, ln = tid;
  const uint32_t n = a.n;
  float* ring = reinterpret_cast<float*>(smem);
  uint32_t* s_nb = reinterpret_cast<uint32_t*>(smem + D * KW * sizeof(float));

  const PhiStep st = phi_step(a);
  note_stamp(a.stamps, a.desc, AMMSB_STAMP_PHI);
  const uint32_t g = a.group_begin + blockIdx.x;
  if (g >= st.group_end) return;  // block-uniform
  PHI_BLK(0);
  const float EPS = a.epsilon;
  if (a.noise_on) zig_load(&zig);

  f32x2 bf[HP];
  bool beta_safe = true;
#pragma unroll
  for (int p = 0; p < HP; ++p) {
    const float b0 = a.beta[2 * (tid + 2 * L * p) + 1];
    const float b1 = a.beta[2 * (tid + 2 * L * p + L) + 1];
    bf[p] = f32x2{b0 - EPS, b1 - EPS};
    beta_safe = beta_safe && in_range(b0, EPS, kBetaHi) && in_range(b1, EPS, kBetaHi);
  }
  ammsb_seed rs = {0, 0};
  if (a.noise_on) rs = a.seeds[(uint64_t)g * VL + VLn::vlane(tid)];

  // 4 KiB at `src_row` into ring slot `slot`: piece t carries chunks j = 4t .. 4t+3 (16 lanes x 16 B each), so the row
  // lands as [j][64] and lane ln reads column ln + 64 j at [j * 64 + ln]
  auto fetch = [&](const float* src_row, uint32_t slot, bool nt) {
    const float* src = src_row + 4 * tid;
    char* dst = smem + slot * (KW * sizeof(float));
    if (nt) {
#pragma unroll
      for (int t = 0; t < PIECES; ++t)
        __builtin_amdgcn_global_load_lds((glb_void_t*)(src + 4 * L * t), (lds_void_t*)(dst + 1024 * t), 16, 0, 2);
    } else {
#pragma unroll
      for (int t = 0; t < PIECES; ++t)
        __builtin_amdgcn_global_load_lds((glb_void_t*)(src + 4 * L * t), (lds_void_t*)(dst + 1024 * t), 16, 0, 0);
    }
  };

  for (uint64_t i = g; i < st.n_nodes; i += st.G) {
    const uint32_t node = a.nodes[i];
    float* out = a.phi_vec + i * K;
    __syncthreads();  // orders the LDS traffic of consecutive nodes
    for (uint32_t q = tid; q < n; q += L) {
      const uint32_t nbq = a.neighbors[i * n + q];
      const bool y = set_has(a.set, make_edge(node, nbq));
      s_nb[q] = nbq | (y ? 0x80000000u : 0u);
    }
    __syncthreads();
    // row r of the node's stream: neighbour r for r < n, the node's noise row (its output row, holding the raw normals
    // the loop has stored there) for r == n
    auto request = [&](uint32_t r, uint32_t slot) {
      if (r < n) {
        const uint32_t nbr = __builtin_amdgcn_readfirstlane(s_nb[r] & 0x7fffffffu);
        fetch(rpm_row(a.pi, nbr), slot, a.rows_nt != 0);
      } else {
        // (the row's address is loop-invariant: computed here, behind an empty asm, and not hoisted into four more
        // 64-bit address registers that live across the row loop -- the kernel sits exactly at its register budget, and
        // a spilled address is reloaded by a scratch load whose wait drains the ring every trip)
        const float* o = out;
        asm volatile("" : "+s"(o));
        fetch(o, slot, false);
      }
    };

    const float phi_sum = a.phi_sum[node];
    const float inv_phi_sum = 1.0f / phi_sum;
    const float* row_a = rpm_row(a.pi, node);
    f32x2 pi_a[HP], grads[HP], rden[HP];
    bool node_safe = beta_safe && in_range(phi_sum, kPhiSumLo, kPhiSumHi);
#pragma unroll
    for (int p = 0; p < HP; ++p)
      pi_a[p] = f32x2{__builtin_nontemporal_load(row_a + tid + 2 * L * p), __builtin_nontemporal_load(row_a + tid + 2 * L * p + L)};
    request(0, 0);  // (n >= 2 here)
    request(1, 1);
#pragma unroll
    for (int p = 0; p < HP; ++p) {
      grads[p] = f32x2{0.0f, 0.0f};
      const f32x2 den = pi_a[p] * phi_sum;
      rden[p] = f32x2{exact_rcp(den.x), exact_rcp(den.y)};
      node_safe = node_safe && in_range(den.x, kDenLo, kDenHi) && in_range(den.y, kDenLo, kDenHi);
    }

    uint32_t slot = 0;  // q % 3
    for (uint32_t q = 0; q < n; ++q) {
      float* row = ring + slot * KW;
      const uint32_t free_slot = slot == 0 ? 2u : slot - 1u;  // (q + 2) % 3 == (q - 1) % 3: row q-1's, read for the last time
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      // rows q+1 and q+2 in flight while row q is reduced; the stream has n + 1 rows when there is noise to fetch
      const uint32_t last = a.noise_on ? n : n - 1;
      if (q + 2 <= last) {
        request(q + 2, free_slot);
        // one of the lane's normals per row, drawn while the rows are on their way and parked (raw) in the output row
        if (a.noise_on && q < (uint32_t)KV) {
          const float z = rng_normal(rs, &zig);
          float* o = out;
          asm volatile("" : "+s"(o));  // (as in request(): no hoisted per-lane address)
          if (VLn::keeps(tid, q)) o[ln + 64 * (q / VLn::PER)] = z;
          // (a store is one more vector-memory operation, younger than the three rows: rows q+1, q+2 and it may be pending)
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PIECES + 1) : "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PIECES) : "memory");
        }
      } else if (q + 1 <= last) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      bool y = (s_nb[q] >> 31) != 0;
      y = __builtin_amdgcn_readfirstlane((int)y) != 0;
      const float e = y ? EPS : 1.0f - EPS;

      // pass 1 (phi.cc:241-253), as update_phi_lds_kernel
      float partial = 0.0f, lo = 1.0f;
      auto pass1 = [&](auto link) {
#pragma unroll
        for (int p = 0; p < HP; ++p) {
          const f32x2 pin = f32x2{row[ln + 128 * p], row[ln + 128 * p + 64]};
          const f32x2 tt0 = pin * bf[p];
          const f32x2 tt = decltype(link)::value ? tt0 + e : e - tt0;
          const f32x2 pr = pi_a[p] * tt;
          row[ln + 128 * p] = pr.x;
          row[ln + 128 * p + 64] = pr.y;
          VLn::chain(partial, pr.x);
          VLn::chain(partial, pr.y);
          lo = fminf(fminf(lo, fabsf(pr.x)), fabsf(pr.y));
        }
      };
      if (y) pass1(std::true_type{});
      else pass1(std::false_type{});
      const float probs_sum = VLn::tree(partial);  // phi.cc:254-257

      // pass 2 (phi.cc:259-263)
      if (node_safe && lo >= kProbsLo && in_range(probs_sum, kPsumLo, kPsumHi)) {
        const float rps = exact_rcp(probs_sum);
        float ps = phi_sum;
        asm volatile("" : "+v"(ps));  // keeps pi_a * phi_sum from being hoisted into KPT more registers
        const f32x2 psum2 = f32x2{probs_sum, probs_sum}, rps2 = f32x2{rps, rps};
#pragma unroll
        for (int p = 0; p < HP; ++p) {
          const f32x2 pr = f32x2{row[ln + 128 * p], row[ln + 128 * p + 64]};
          f32x2 qv = div_exact3(pr, psum2, rps2);
          qv = div_exact3(qv, pi_a[p] * ps, rden[p]);
          grads[p] += qv - inv_phi_sum;
        }
      } else {
#pragma unroll
        for (int p = 0; p < HP; ++p) {
          const f32x2 den = pi_a[p] * phi_sum;
          float q0 = row[ln + 128 * p] / probs_sum;
          float q1 = row[ln + 128 * p + 64] / probs_sum;
          q0 = q0 / den.x;
          q1 = q1 / den.y;
          grads[p] += f32x2{q0 - inv_phi_sum, q1 - inv_phi_sum};
        }
      }
      slot = slot == 2 ? 0u : slot + 1u;
    }
    // the noise row (row n of the stream) sits in slot n % 3 == `slot`; it was requested two rows ago and may still be
    // on its way (the last row's wait left it pending)
    const float* zrow = ring + slot * KW;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

    // SGLD step, phi.cc:265-274
    const float half = st.eps_t / 2;
#pragma unroll
    for (int p = 0; p < HP; ++p) {
      asm volatile("" ::: "memory");  // one column pair at a time: keeps the 2 HP LDS reads of the noise row from being hoisted
      const f32x2 phi_k = pi_a[p] * phi_sum;
      const f32x2 ep = phi_k * st.eps_t;
      f32x2 bb = f32x2{sqrtf(ep.x), sqrtf(ep.y)};
      if (a.noise_on) bb = bb * f32x2{zrow[ln + 128 * p], zrow[ln + 128 * p + 64]};  // sqrt(eps_t * phi_k) * noise
      const f32x2 ng = grads[p] * a.Nn;
      f32x2 in = a.alpha - phi_k;
      in = in + ng;
      const f32x2 drift = in * half;
      const f32x2 aa = phi_k + drift;
      const f32x2 s2 = aa + bb;
      const float v0 = fabsf(s2.x), v1 = fabsf(s2.y);
      __builtin_nontemporal_store(v0 > 1e-24f ? v0 : 1e-24f, out + tid + 2 * L * p);
      __builtin_nontemporal_store(v1 > 1e-24f ? v1 : 1e-24f, out + tid + 2 * L * p + L);
    }
  }
  if (a.noise_on && tid < VL) a.seeds[(uint64_t)g * VL + tid] = rs;
  PHI_BLK(1);
}

// ---------------------------------------------------------------------------------------------------------
// Short rows (K = 256 / 512: KPT = 4 / 8, one wave per node): TWO neighbour rows per loop iteration.
//
// (U rows per iteration, U = 2 or 4; the text below says two.)
// With 1 KiB rows the single-row loop above is latency-bound, not bandwidth-bound: an iteration is a dependent
// chain LDS read -> probs -> wave tree sum -> divisions -> LDS, about a microsecond, whatever the ring depth, and
// there are not enough nodes in a mini-batch (8193 at C2) to hide it with more waves.  Here the chains of rows q and
// q + 1 are issued together as one straight-line block (no branch between them: the link / non-link sign is a
// multiplication by +-1, which is exact, instead of two code copies), so the scheduler interleaves two independent
// chains per wave.  Every value is computed by the same operations in the same order as in the single-row kernel:
// results are bit-identical (same tests).  Needs an even n; odd n takes the single-row kernel.
// VL = 32: the reference work-group size 32 in the same one-wave-per-node layout (see update_phi_lds_kernel).
// ONE: pi is a single block (every configuration of interest on a 288 GB device; the launcher checks).  Told to the
// compiler as an assumption, which folds the block look-up -- a 32-bit division, a table read and the registers that
// keep them alive -- out of every rpm_row() of the kernel: -3 % per launch at K = 256, -10 % at K = 32 (same-box A/B).
template <int KPT, int D, int U, int VL = 64, bool ONE = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(KPT == 4 ? 4 : 3))) void update_phi_lds2_kernel(const PhiArgs a) {
  if constexpr (ONE) __builtin_assume(a.pi.num_blocks == 1);
  constexpr int L = 64, KW = 64 * KPT, K = L * KPT, PIECES = KPT / 4, HP = KPT / 2;
  using VLn = VLane<VL>;
  constexpr int KV = KPT * VLn::PER;  // columns (= normals) per virtual lane
  // probs[] of the U rows in flight stay in registers when that is at most 16 of them per lane (K = 256: 4 rows x 4
  // columns, K = 512: 2 x 8): no write-back into the ring slot and re-read between the two passes
  constexpr bool REGP = U * HP <= 8;
  // the lane's KPT noise factors sqrt(eps_t phi_k) * normal stay in registers when the first row group draws all of
  // them (KPT == U): no [KW] LDS slot -- at K = 256 that is the kilobyte that keeps a CU from holding 16 of these
  // one-wave blocks (8 KiB ring + 1.5 KiB ziggurat tables + 128 B neighbour ids each)
  constexpr bool REGN = KPT == U;
  static_assert(D > U && (D & (D - 1)) == 0 && (U == 2 || U == 4) && (D - U) * PIECES <= 63, "ring depth / rows per step");
  extern __shared__ __align__(16) char smem[];  // [D][KW] ring, [KW] normals, [n] u32 (id | link bit)
  __shared__ ZigTables zig;
  const int tid = threadIdx.x, ln = tid;
  float* ring = reinterpret_cast<float*>(smem);
  float* s_noise = ring + D * KW;  // (unused with REGN)
  uint32_t* s_nb = reinterpret_cast<uint32_t*>(smem + (D + (REGN ? 0 : 1)) * KW * sizeof(float));
  float nz[REGN ? KPT : 1];

  const PhiStep st = phi_step(a);
  note_stamp(a.stamps, a.desc, AMMSB_STAMP_PHI);
  const uint32_t g0 = a.group_begin + blockIdx.x;
  if (g0 >= st.group_end) return;  // block-uniform
  PHI_BLK(0);
  const uint32_t n = a.n;
  const float EPS = a.epsilon;
  if (a.noise_on) zig_load(&zig);

  f32x2 bf[HP];
  bool beta_safe = true;
#pragma unroll
  for (int p = 0; p < HP; ++p) {
    const float b0 = a.beta[2 * (tid + 2 * L * p) + 1];
    const float b1 = a.beta[2 * (tid + 2 * L * p + L) + 1];
    bf[p] = f32x2{b0 - EPS, b1 - EPS};
    beta_safe = beta_safe && in_range(b0, EPS, kBetaHi) && in_range(b1, EPS, kBetaHi);
  }
  ammsb_seed rs = {0, 0};

  auto request = [&](uint32_t q, uint32_t slot) {
    const uint32_t nbr = __builtin_amdgcn_readfirstlane(s_nb[q] & 0x7fffffffu);
    const float* src = rpm_row(a.pi, nbr) + 4 * tid;
    char* dst = smem + slot * (KW * sizeof(float));
    if (a.rows_nt) {
#pragma unroll
      for (int t = 0; t < PIECES; ++t)
        __builtin_amdgcn_global_load_lds((glb_void_t*)(src + 4 * L * t), (lds_void_t*)(dst + 1024 * t), 16, 0, 2);
    } else {
#pragma unroll
      for (int t = 0; t < PIECES; ++t)
        __builtin_amdgcn_global_load_lds((glb_void_t*)(src + 4 * L * t), (lds_void_t*)(dst + 1024 * t), 16, 0, 0);
    }
  };

  PHI_TRACE(0);
  // (gridDim.x < groups only with AMMSB_PHI_PERSIST, see launch_phi_lds2: block b then takes groups b, b + gridDim.x, ...;
  // virtual group g keeps stream g L + l and its nodes whichever block runs it)
  for (uint32_t g = g0; g < st.group_end; g += gridDim.x) {
  if (a.noise_on) rs = a.seeds[(uint64_t)g * VL + VLn::vlane(tid)];
  for (uint64_t i = g; i < st.n_nodes; i += st.G) {
    const uint32_t node = a.nodes[i];
    __syncthreads();  // orders the LDS traffic of consecutive nodes
    // the neighbour ids first: the first rows are requested as soon as they are known, and the edge-set probes of
    // the same neighbours (another dependent round trip) fly together with those rows instead of ahead of them
    for (uint32_t q = tid; q < n; q += L) s_nb[q] = a.neighbors[i * n + q];
    __syncthreads();
    PHI_TRACE(1);

    const float phi_sum = a.phi_sum[node];
    const float inv_phi_sum = 1.0f / phi_sum;
    const float* row_a = rpm_row(a.pi, node);
    f32x2 pi_a[HP], grads[HP], rden[HP];
    bool node_safe = beta_safe && in_range(phi_sum, kPhiSumLo, kPhiSumHi);
#pragma unroll
    for (int p = 0; p < HP; ++p)
      pi_a[p] = f32x2{__builtin_nontemporal_load(row_a + tid + 2 * L * p), __builtin_nontemporal_load(row_a + tid + 2 * L * p + L)};
#pragma unroll
    for (uint32_t r = 0; r < (uint32_t)(D - U); ++r)
      if (r < n) request(r, r);  // rows 0 .. D-U-1 fly during the per-node set-up
    // The probes' loads are issued here and consumed after the lane's normals have been drawn (REGN: all KV of them):
    // the ziggurat -- ~1000 cycles a draw for a wave, measured with in-kernel stamps (tools/phi_trace.sh) -- used to
    // sit in the first row iteration, behind rows that had long landed; now it runs under the probes' round trip.
    // (n <= 64 here: one probe per lane at most, the loop form is kept for the general case below)
    const bool one_probe = n <= (uint32_t)L;
    const uint32_t my_nb = one_probe && (uint32_t)tid < n ? s_nb[tid] : 0u;
    SetProbe my_probe = {};
    if (one_probe && (uint32_t)tid < n) my_probe = set_probe(a.set, make_edge(node, my_nb));
    float zraw[REGN ? KV : 1];
    if constexpr (REGN) {
      if (a.noise_on) {
#pragma unroll
        for (int j = 0; j < KV; ++j) zraw[j] = rng_normal(rs, &zig);
      }
    }
    if (one_probe) {
      if ((uint32_t)tid < n && set_hit(my_probe, make_edge(node, my_nb))) s_nb[tid] = my_nb | 0x80000000u;
    } else {
      for (uint32_t q = tid; q < n; q += L) {
        const uint32_t nb = s_nb[q];
        if (set_has(a.set, make_edge(node, nb))) s_nb[q] = nb | 0x80000000u;  // (loads return in order: after the rows above)
      }
    }
    __syncthreads();
    PHI_TRACE(2);
#pragma unroll
    for (int p = 0; p < HP; ++p) {
      grads[p] = f32x2{0.0f, 0.0f};
      const f32x2 den = pi_a[p] * phi_sum;
      rden[p] = f32x2{exact_rcp(den.x), exact_rcp(den.y)};
      node_safe = node_safe && in_range(den.x, kDenLo, kDenHi) && in_range(den.y, kDenLo, kDenHi);
      const f32x2 ep = den * st.eps_t;
      if constexpr (REGN) {
        nz[2 * p] = sqrtf(ep.x);
        nz[2 * p + 1] = sqrtf(ep.y);
      } else {
        s_noise[ln + 128 * p] = sqrtf(ep.x);
        s_noise[ln + 128 * p + 64] = sqrtf(ep.y);
      }
    }

    PHI_TRACE(3);
    for (uint32_t q = 0; q < n; q += U) {  // n is a multiple of U (dispatch)
      float* row[U];
#pragma unroll
      for (int r = 0; r < U; ++r) row[r] = ring + ((q + r) & (D - 1)) * KW;
      // every LDS read of rows q-U .. q-1 has been consumed; their slots take rows q+D-U .. q+D-1
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      PHI_TRACE(8 + 4 * (q / U));
      if (q + (D - U) < n) {  // n a multiple of U: all U rows exist
#pragma unroll
        for (int r = 0; r < U; ++r) request(q + (D - U) + r, (q + (D - U) + r) & (D - 1));
      }
      // U of the lane's KPT normals per iteration, drawn while the rows are on their way (ascending column order)
      if (a.noise_on) {
        if constexpr (REGN) {
          if (q == 0) {  // (drawn in the prologue, under the probes' latency; stream order = ascending j as before)
#pragma unroll
            for (int j = 0; j < KV; ++j)
              if (VLn::keeps(tid, (uint32_t)j)) nz[j / VLn::PER] = nz[j / VLn::PER] * zraw[j];
          }
        } else {
#pragma unroll
          for (int r = 0; r < U; ++r)
            if (q + r < (uint32_t)KV) {
              const float z = rng_normal(rs, &zig);
              const uint32_t c = (q + r) / VLn::PER;
              if (VLn::keeps(tid, q + r)) s_noise[ln + 64 * c] = s_noise[ln + 64 * c] * z;
            }
        }
      }
      {  // rows q .. q+U-1 landed; rows q+U .. min(q+D-1, n-1) may still be in flight (a multiple of U of them)
        const uint32_t rem = n - U - q < (uint32_t)(D - U) ? n - U - q : (uint32_t)(D - U);
        if (rem >= 6) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * PIECES) : "memory");
        else if (rem >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * PIECES) : "memory");
        else if (rem >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PIECES) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      PHI_TRACE(8 + 4 * (q / U) + 1);
      float ee[U], sg[U], part[U], lo[U], psum[U];
      f32x2 prr[REGP ? U : 1][REGP ? HP : 1];
      bool fast[U], all_fast = true;
#pragma unroll
      for (int r = 0; r < U; ++r) {
        const bool y = (__builtin_amdgcn_readfirstlane(s_nb[q + r]) >> 31) != 0;
        ee[r] = y ? EPS : 1.0f - EPS;
        // pin * (EPS - beta) + e == e - pin * (beta - EPS) == e + (-(pin * (beta - EPS))): the negation is exact, so
        // multiplying by -1 for a non-link and adding gives the single-row kernel's value bit for bit
        sg[r] = y ? 1.0f : -1.0f;
        part[r] = 0.0f;
        lo[r] = 1.0f;
      }
      // pass 1 of all U rows (phi.cc:241-253): probs[] in place of the row, lane partials in ascending column order
#pragma unroll
      for (int p = 0; p < HP; ++p) {
        float vx[U], vy[U];
#pragma unroll
        for (int r = 0; r < U; ++r) {
          const f32x2 pin = f32x2{row[r][ln + 128 * p], row[r][ln + 128 * p + 64]};
          const f32x2 tt = (pin * bf[p]) * sg[r] + ee[r];
          const f32x2 pr = pi_a[p] * tt;
          if constexpr (REGP) {
            prr[r][p] = pr;
          } else {
            row[r][ln + 128 * p] = pr.x;
            row[r][ln + 128 * p + 64] = pr.y;
          }
          vx[r] = pr.x;
          vy[r] = pr.y;
          lo[r] = fminf(fminf(lo[r], fabsf(pr.x)), fabsf(pr.y));
        }
        VLn::template chain_rows<U>(part, vx);  // all U rows' chains advance by the column pair (ammsb_dev.h)
        VLn::template chain_rows<U>(part, vy);
      }
      float prcp[U];
      VLn::template tree_rows_rcp<U>(part, psum, prcp);  // phi.cc:254-257, and RN(1 / probs_sum) of all U rows in one division
#pragma unroll
      for (int r = 0; r < U; ++r) {
        fast[r] = node_safe && lo[r] >= kProbsLo && in_range(psum[r], kPsumLo, kPsumHi);
        all_fast = all_fast && fast[r];
      }
      // pass 2 (phi.cc:259-263): grads += (probs / probs_sum) / (pi * phi_sum) - 1 / phi_sum, rows in order
      if (all_fast) {
        float ps = phi_sum;
        asm volatile("" : "+v"(ps));  // keeps pi_a * phi_sum from being hoisted into KPT more registers
        f32x2 s2[U], r2[U];
#pragma unroll
        for (int r = 0; r < U; ++r) {
          s2[r] = f32x2{psum[r], psum[r]};
          r2[r] = f32x2{prcp[r], prcp[r]};
        }
#pragma unroll
        for (int p = 0; p < HP; ++p) {
          const f32x2 den = pi_a[p] * ps;
          f32x2 v[U];
#pragma unroll
          for (int r = 0; r < U; ++r) {
            f32x2 pr;
            if constexpr (REGP) pr = prr[r][p];
            else pr = f32x2{row[r][ln + 128 * p], row[r][ln + 128 * p + 64]};
            v[r] = div_exact3(div_exact3(pr, s2[r], r2[r]), den, rden[p]);
          }
#pragma unroll
          for (int r = 0; r < U; ++r) grads[p] += v[r] - inv_phi_sum;
        }
      } else {
#pragma unroll
        for (int r = 0; r < U; ++r) {
          const float probs_sum = psum[r];
          if (fast[r]) {
            const float rps = prcp[r];
            float ps = phi_sum;
            asm volatile("" : "+v"(ps));
            const f32x2 psum2 = f32x2{probs_sum, probs_sum}, rps2 = f32x2{rps, rps};
#pragma unroll
            for (int p = 0; p < HP; ++p) {
              f32x2 pr;
              if constexpr (REGP) pr = prr[r][p];
              else pr = f32x2{row[r][ln + 128 * p], row[r][ln + 128 * p + 64]};
              f32x2 qv = div_exact3(pr, psum2, rps2);
              qv = div_exact3(qv, pi_a[p] * ps, rden[p]);
              grads[p] += qv - inv_phi_sum;
            }
          } else {
#pragma unroll
            for (int p = 0; p < HP; ++p) {
              const f32x2 den = pi_a[p] * phi_sum;
              float v0, v1;
              if constexpr (REGP) {
                v0 = prr[r][p].x / probs_sum;
                v1 = prr[r][p].y / probs_sum;
              } else {
                v0 = row[r][ln + 128 * p] / probs_sum;
                v1 = row[r][ln + 128 * p + 64] / probs_sum;
              }
              v0 = v0 / den.x;
              v1 = v1 / den.y;
              grads[p] += f32x2{v0 - inv_phi_sum, v1 - inv_phi_sum};
            }
          }
        }
      }
    }

    PHI_TRACE(4);
    // normals the loop did not get to (n < KPT): one rolled loop, a single copy of the ziggurat code
    if constexpr (!REGN) {  // (REGN: n >= U == KPT, the first row group drew them all)
      if (a.noise_on) {
#pragma unroll 1
        for (uint32_t j = n; j < (uint32_t)KV; ++j) {
          const float z = rng_normal(rs, &zig);
          if (VLn::keeps(tid, j)) s_noise[ln + 64 * (j / VLn::PER)] = s_noise[ln + 64 * (j / VLn::PER)] * z;
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // SGLD step, phi.cc:265-274
    float* out = a.phi_vec + i * K;
    const float half = st.eps_t / 2;
#pragma unroll
    for (int p = 0; p < HP; ++p) {
      f32x2 bb;
      if constexpr (REGN) bb = f32x2{nz[2 * p], nz[2 * p + 1]};
      else bb = f32x2{s_noise[ln + 128 * p], s_noise[ln + 128 * p + 64]};
      const f32x2 phi_k = pi_a[p] * phi_sum;
      const f32x2 ng = grads[p] * a.Nn;
      f32x2 in = a.alpha - phi_k;
      in = in + ng;
      const f32x2 drift = in * half;
      const f32x2 aa = phi_k + drift;
      const f32x2 s2 = aa + bb;
      const float v0 = fabsf(s2.x), v1 = fabsf(s2.y);
      if (a.rows_nt) {  // (a small pi: the row stays in the cache for update_pi, the next kernel)
        __builtin_nontemporal_store(v0 > 1e-24f ? v0 : 1e-24f, out + tid + 2 * L * p);
        __builtin_nontemporal_store(v1 > 1e-24f ? v1 : 1e-24f, out + tid + 2 * L * p + L);
      } else {
        out[tid + 2 * L * p] = v0 > 1e-24f ? v0 : 1e-24f;
        out[tid + 2 * L * p + L] = v1 > 1e-24f ? v1 : 1e-24f;
      }
    }
    PHI_TRACE(5);
  }
  if (a.noise_on && tid < VL) a.seeds[(uint64_t)g * VL + tid] = rs;
  }
  PHI_BLK(1);
}

// ---------------------------------------------------------------------------------------------------------
// Short rows, TWO NODES PER WAVE (K = 32 KPT: 256 / 512; reference work-group size VL = 32 or 64; n % U == 0).
//
// At K = 256 a node is 32 KB of rows and its wave spends most of its instructions on what does not scale with the row:
// the WG_SUM tree, the reciprocal, the range checks, the ring bookkeeping, the waits (SQ counters, DESIGN.md 4.7: 88
// vector instructions per neighbour row per wave, 43 % of a wave's life parked on s_waitcnt).  Here each HALF of the
// wave owns a node: physical lane l of half h holds columns l + 32 j of node h's rows, so every instruction of the
// tree / reciprocal / bookkeeping serves two nodes, and a CU holds twice the nodes per resident wave.
//   VL = 32: physical lane l IS the reference's work-item l (columns l + 32 j): its lane partial is the plain chain.
//   VL = 64: physical lane l carries the reference's work-items l (even j: columns l + 64 i) and l + 32 (odd j): two
//            lane partials, whose sum is the first level of the reference's tree (aux[l] += aux[l + 32], sum.cc:23-29),
//            and two RNG streams.
// The five remaining tree levels run on both halves at once (v_permlane16_swap + DPP row shifts stay inside 32-lane
// halves / 16-lane rows).  A neighbour row pair travels as PIECES LDS-DMA instructions of 512 B per half.  Link flags,
// neighbour ids and the fast / slow division choice are per half, i.e. per-lane values here, not wave-uniform ones.
// Same operations in the same order on every value as update_phi_lds2_kernel / update_phi_kernel: bit-identical.
template <int KPT, int D, int U, int VL>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(KPT <= 8 ? 4 : 3))) void update_phi_pair_kernel(const PhiArgs a) {
  constexpr int K = 32 * KPT, HP = KPT / 2, PIECES = KPT / 4, SLOT = 2 * K;  // SLOT: floats of one ring slot (both halves)
  constexpr int NS = VL / 32;  // reference work-items (lane partials, RNG streams) per physical lane
  static_assert((VL == 32 || VL == 64) && D > U && (D & (D - 1)) == 0 && (U == 2 || U == 4) && (D - U) * PIECES <= 63, "shape");
  extern __shared__ __align__(16) char smem[];  // [D][PIECES][2 halves][128] ring, [2][K] noise, [2][n] u32 (id | link bit)
  __shared__ ZigTables zig;
  const int tid = threadIdx.x, h = tid >> 5, l = tid & 31;
  float* ring = reinterpret_cast<float*>(smem);
  float* s_nz = ring + D * SLOT + h * K + l;  // this lane's noise factors: column l + 32 j at [32 j]
  const uint32_t n = a.n;
  uint32_t* s_nb = reinterpret_cast<uint32_t*>(smem + (size_t)(D + 1) * SLOT * sizeof(float)) + h * n;  // this half's list
  // column l + 32 j of this half's row in a slot: piece j / 4, then [half][128]
  auto lds_col = [&](int j) -> int { return (j >> 2) * 256 + h * 128 + l + 32 * (j & 3); };

  const PhiStep st = phi_step(a);
  if (st.n_nodes == 0) return;  // (uniform) a skipped step
  note_stamp(a.stamps, a.desc, AMMSB_STAMP_PHI);
  const uint32_t g = a.group_begin + blockIdx.x * 2 + h;  // this half's virtual group
  const bool live = g < st.group_end;
  if (!__builtin_amdgcn_readfirstlane((int)(a.group_begin + blockIdx.x * 2 < st.group_end))) return;  // neither half
  const float EPS = a.epsilon;
  if (a.noise_on) zig_load(&zig);

  f32x2 bf[HP];
  bool beta_safe = true;
#pragma unroll
  for (int p = 0; p < HP; ++p) {
    const float b0 = a.beta[2 * (l + 64 * p) + 1];
    const float b1 = a.beta[2 * (l + 64 * p + 32) + 1];
    bf[p] = f32x2{b0 - EPS, b1 - EPS};
    beta_safe = beta_safe && in_range(b0, EPS, kBetaHi) && in_range(b1, EPS, kBetaHi);
  }
  ammsb_seed rs[NS];
#pragma unroll
  for (int v = 0; v < NS; ++v) {
    rs[v] = ammsb_seed{0, 0};
    if (live && a.noise_on) rs[v] = a.seeds[(uint64_t)g * VL + l + 32 * v];
  }

  // the halving tree over one half's 32 lane values (levels 16 .. 1), both halves at once; result in every lane
  auto tree32 = [&](float v) -> float {
    typedef unsigned int u2 __attribute__((ext_vector_type(2)));
    const u2 t = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(t[0]) + __uint_as_float(t[1]);
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x108, 0xf, 0xf, true));  // row_shl:8
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x104, 0xf, 0xf, true));  // row_shl:4
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x102, 0xf, 0xf, true));  // row_shl:2
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x101, 0xf, 0xf, true));  // row_shl:1
    const float s0 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 0));
    const float s1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 32));
    return h ? s1 : s0;
  };

  // row q of each half's neighbour list into ring slot `slot` (512 B per half and piece)
  // a row base per half from two scalar address computations (a per-lane rpm_row would carry a 64-bit division)
  auto half_row = [&](uint32_t id) -> const float* {
    const float* r0 = rpm_row(a.pi, __builtin_amdgcn_readlane(id, 0));
    const float* r1 = rpm_row(a.pi, __builtin_amdgcn_readlane(id, 32));
    return h ? r1 : r0;
  };
  auto request = [&](uint32_t q, uint32_t slot) {
    const float* src = half_row(s_nb[q] & 0x7fffffffu) + 4 * l;
    char* dst = smem + (size_t)slot * (SLOT * sizeof(float));
#pragma unroll
    for (int t = 0; t < PIECES; ++t)
      __builtin_amdgcn_global_load_lds((glb_void_t*)(src + 128 * t), (lds_void_t*)(dst + 1024 * t), 16, 0, 2);
  };

  const uint32_t trips = (st.n_nodes + st.G - 1) / st.G;  // wave-uniform
  for (uint32_t t = 0; t < trips; ++t) {
    const uint64_t i_raw = (uint64_t)g + (uint64_t)t * st.G;
    const bool on = live && i_raw < st.n_nodes;
    if (!__builtin_amdgcn_readfirstlane((int)(__ballot(on) != 0ull))) continue;  // no half has a node in this trip
    const uint64_t i = on ? i_raw : 0;  // an idle half shadows node 0 and stores nothing
    const uint32_t node = a.nodes[i];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (one wave: its LDS operations complete in order)
    __builtin_amdgcn_wave_barrier();
    for (uint32_t q = l; q < n; q += 32) s_nb[q] = a.neighbors[i * n + q];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();

    const float phi_sum = a.phi_sum[node];
    const float inv_phi_sum = 1.0f / phi_sum;
    const float* row_a = half_row(node);
    f32x2 pi_a[HP], grads[HP], rden[HP];
    bool node_safe = beta_safe && in_range(phi_sum, kPhiSumLo, kPhiSumHi);
#pragma unroll
    for (int p = 0; p < HP; ++p)
      pi_a[p] = f32x2{__builtin_nontemporal_load(row_a + l + 64 * p), __builtin_nontemporal_load(row_a + l + 64 * p + 32)};
#pragma unroll
    for (uint32_t r = 0; r < (uint32_t)(D - U); ++r)
      if (r < n) request(r, r);  // rows 0 .. D-U-1 fly during the per-node set-up
    // the edge-set probes of the same neighbours fly together with those rows (loads return in order: after them)
    for (uint32_t q = l; q < n; q += 32) {
      const uint32_t nb = s_nb[q];
      if (set_has(a.set, make_edge(node, nb))) s_nb[q] = nb | 0x80000000u;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int p = 0; p < HP; ++p) {
      grads[p] = f32x2{0.0f, 0.0f};
      const f32x2 den = pi_a[p] * phi_sum;
      rden[p] = f32x2{exact_rcp(den.x), exact_rcp(den.y)};
      node_safe = node_safe && in_range(den.x, kDenLo, kDenHi) && in_range(den.y, kDenLo, kDenHi);
      // sqrt(eps_t * phi_k) of the SGLD step, parked in the lane's noise slots (LDS: not live across the row loop)
      const f32x2 ep = den * st.eps_t;
      s_nz[64 * p] = sqrtf(ep.x);
      s_nz[64 * p + 32] = sqrtf(ep.y);
    }
    // the lane's normals, while the first rows are on their way: stream v draws for its columns in ascending order
    // (VL = 32: one stream, columns j = 0, 1, 2, ...; VL = 64: stream 0 even j = .x of every pair, stream 1 odd j = .y).
    // One rolled loop: a single copy of the ziggurat code.
    if (a.noise_on && on) {  // (an idle half's streams do not advance)
#pragma unroll 1
      for (int p = 0; p < HP; ++p) {
        const float z0 = rng_normal(rs[0], &zig);
        const float z1 = rng_normal(rs[NS - 1], &zig);
        s_nz[64 * p] = s_nz[64 * p] * z0;
        s_nz[64 * p + 32] = s_nz[64 * p + 32] * z1;
      }
    }

    for (uint32_t q = 0; q < n; q += U) {  // n is a multiple of U (dispatch)
      float* row[U];
#pragma unroll
      for (int r = 0; r < U; ++r) row[r] = ring + ((q + r) & (D - 1)) * SLOT;
      // every LDS read of rows q-U .. q-1 has been consumed; their slots take rows q+D-U .. q+D-1
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (q + (D - U) < n) {
#pragma unroll
        for (int r = 0; r < U; ++r) request(q + (D - U) + r, (q + (D - U) + r) & (D - 1));
      }
      {  // rows q .. q+U-1 landed; rows q+U .. min(q+D-1, n-1) may still be in flight (a multiple of U of them)
        const uint32_t rem = n - U - q < (uint32_t)(D - U) ? n - U - q : (uint32_t)(D - U);
        if (rem >= 6) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * PIECES) : "memory");
        else if (rem >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * PIECES) : "memory");
        else if (rem >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PIECES) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      float ee[U], sg[U], lo[U], psum[U];
      float part[U][NS];
      f32x2 prr[U][HP];
      bool fast[U], all_fast = true;
#pragma unroll
      for (int r = 0; r < U; ++r) {
        const bool y = (s_nb[q + r] >> 31) != 0;  // this half's flag
        ee[r] = y ? EPS : 1.0f - EPS;
        sg[r] = y ? 1.0f : -1.0f;  // e - pin (beta - EPS) == e + (-(pin (beta - EPS))): the negation is exact
        lo[r] = 1.0f;
#pragma unroll
        for (int v = 0; v < NS; ++v) part[r][v] = 0.0f;
      }
      // pass 1 of all U rows (phi.cc:241-253): probs[], lane partials in ascending column order
#pragma unroll
      for (int p = 0; p < HP; ++p) {
#pragma unroll
        for (int r = 0; r < U; ++r) {
          const f32x2 pin = f32x2{row[r][lds_col(2 * p)], row[r][lds_col(2 * p + 1)]};
          const f32x2 tt = (pin * bf[p]) * sg[r] + ee[r];
          const f32x2 pr = pi_a[p] * tt;
          prr[r][p] = pr;
          part[r][0] += pr.x;       // VL = 32: one chain over j = 2p, 2p + 1; VL = 64: work-item l takes the even j,
          part[r][NS - 1] += pr.y;  // work-item l + 32 the odd ones
          lo[r] = fminf(fminf(lo[r], fabsf(pr.x)), fabsf(pr.y));
        }
      }
#pragma unroll
      for (int r = 0; r < U; ++r) {
        // VL = 64: aux[l] += aux[l + 32] is the tree's first level (sum.cc:23-29), inside the lane here
        const float lane_sum = NS == 1 ? part[r][0] : part[r][0] + part[r][NS - 1];
        psum[r] = tree32(lane_sum);  // phi.cc:254-257
        fast[r] = node_safe && lo[r] >= kProbsLo && in_range(psum[r], kPsumLo, kPsumHi);
        all_fast = all_fast && fast[r];
      }
      // pass 2 (phi.cc:259-263): grads += (probs / probs_sum) / (pi * phi_sum) - 1 / phi_sum, rows in order
      if (all_fast) {
        float ps = phi_sum;
        asm volatile("" : "+v"(ps));  // keeps pi_a * phi_sum from being hoisted into KPT more registers
        f32x2 s2[U], r2[U];
#pragma unroll
        for (int r = 0; r < U; ++r) {
          const float rc = exact_rcp(psum[r]);
          s2[r] = f32x2{psum[r], psum[r]};
          r2[r] = f32x2{rc, rc};
        }
#pragma unroll
        for (int p = 0; p < HP; ++p) {
          const f32x2 den = pi_a[p] * ps;
          f32x2 v[U];
#pragma unroll
          for (int r = 0; r < U; ++r) v[r] = div_exact3(div_exact3(prr[r][p], s2[r], r2[r]), den, rden[p]);
#pragma unroll
          for (int r = 0; r < U; ++r) grads[p] += v[r] - inv_phi_sum;
        }
      } else {
#pragma unroll
        for (int r = 0; r < U; ++r) {
          const float probs_sum = psum[r];
          if (fast[r]) {
            const float rps = exact_rcp(probs_sum);
            const f32x2 psum2 = f32x2{probs_sum, probs_sum}, rps2 = f32x2{rps, rps};
#pragma unroll
            for (int p = 0; p < HP; ++p) {
              f32x2 qv = div_exact3(prr[r][p], psum2, rps2);
              qv = div_exact3(qv, pi_a[p] * phi_sum, rden[p]);
              grads[p] += qv - inv_phi_sum;
            }
          } else {
#pragma unroll
            for (int p = 0; p < HP; ++p) {
              const f32x2 den = pi_a[p] * phi_sum;
              float v0 = prr[r][p].x / probs_sum, v1 = prr[r][p].y / probs_sum;
              v0 = v0 / den.x;
              v1 = v1 / den.y;
              grads[p] += f32x2{v0 - inv_phi_sum, v1 - inv_phi_sum};
            }
          }
        }
      }
    }

    // SGLD step, phi.cc:265-274
    if (on) {
      float* out = a.phi_vec + i * K;
      const float half = st.eps_t / 2;
#pragma unroll
      for (int p = 0; p < HP; ++p) {
        const f32x2 phi_k = pi_a[p] * phi_sum;
        const f32x2 ng = grads[p] * a.Nn;
        f32x2 in = a.alpha - phi_k;
        in = in + ng;
        const f32x2 drift = in * half;
        const f32x2 aa = phi_k + drift;
        const f32x2 s2 = aa + f32x2{s_nz[64 * p], s_nz[64 * p + 32]};  // sqrt(eps_t phi_k) * noise (noise off: * 1, phi.cc:673-677)
        const float v0 = fabsf(s2.x), v1 = fabsf(s2.y);
        __builtin_nontemporal_store(v0 > 1e-24f ? v0 : 1e-24f, out + l + 64 * p);
        __builtin_nontemporal_store(v1 > 1e-24f ? v1 : 1e-24f, out + l + 64 * p + 32);
      }
    }
  }
  if (live && a.noise_on) {
#pragma unroll
    for (int v = 0; v < NS; ++v) a.seeds[(uint64_t)g * VL + l + 32 * v] = rs[v];
  }
}

template <int KPT, int D, int U, int VL>
int launch_phi_pair(ammsb_ctx* ctx, const PhiArgs& a, uint32_t n_groups, hipStream_t s) {
  const size_t lds = (size_t)(D + 1) * 2 * 32 * KPT * sizeof(float) + 2 * sizeof(uint32_t) * a.n;
  static const std::string name = ammsb_kname("update_phi_pair_kernel<%d, %d, %d, %d>", KPT, D, U, VL);
  ctx->kernel_name[AMMSB_KN_PHI] = name.c_str();
  update_phi_pair_kernel<KPT, D, U, VL><<<(n_groups + 1) / 2, 64, lds, s>>>(a);
  AMMSB_LAUNCH_CHECK(ctx);
  return AMMSB_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Small launches (link mini-batches: a few dozen nodes on an otherwise empty chip): ONE NODE PER BLOCK OF RW + 1 WAVES.
//
// A lone wave working through a node is a latency chain: n neighbour rows one after the other (load, probs, WG_SUM
// tree, reciprocal, two divisions each) and then K / wg ziggurat draws per lane -- 19 us at K = 256, 50 us at K = 1024
// whatever the ring depth, and half of all iterations are such launches.  What is sequential in the
// reference is only the ACCUMULATION grads_k += term_q,k over the neighbours q = 0 .. n-1 (phi.cc:259-263) and the
// draw order inside one RNG stream; the terms themselves are independent.  So here the first RW waves of the block take
// the rows round-robin (wave w: q = w, w + RW, ...), each one running the one-wave-per-node arithmetic on its row --
// same lane ownership, same WG_SUM chain and tree (VLane<VL>), same exact divisions -- and parking the row's term
// vector in LDS; the last wave meanwhile draws the node's normals (stream order = ascending column, as everywhere);
// then every thread takes some columns, adds their n terms IN ROW ORDER starting from 0 (the reference's accumulation,
// bit for bit) and makes the SGLD step for them.  LDS: n K floats of terms (128 KiB at K = 1024, n = 32: one block per
// CU, which is all a link batch needs).
template <int KPT, int RW, int VL>
__global__ __launch_bounds__(64 * (RW + 1)) void update_phi_wide_kernel(const PhiArgs a) {
  constexpr int K = 64 * KPT, T = 64 * (RW + 1);
  using VLn = VLane<VL>;
  constexpr int KV = KPT * VLn::PER;  // normals per virtual lane
  extern __shared__ __align__(16) char smem[];  // [n][K] terms, [K] normals, [K] pi_a, [n] u32 (id | link bit)
  __shared__ ZigTables zig;
  const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
  const uint32_t n = a.n;
  float* term = reinterpret_cast<float*>(smem);
  float* s_noise = term + (size_t)n * K;
  float* s_pia = s_noise + K;
  uint32_t* s_nb = reinterpret_cast<uint32_t*>(s_pia + K);

  const PhiStep st = phi_step(a);
  note_stamp(a.stamps, a.desc, AMMSB_STAMP_PHI);
  const uint32_t g = a.group_begin + blockIdx.x;
  if (g >= st.group_end) return;  // block-uniform
  const float EPS = a.epsilon;
  if (a.noise_on) zig_load(&zig);

  float bf[KPT];
  bool beta_safe = true;
#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    const float b = a.beta[2 * (ln + 64 * j) + 1];
    bf[j] = b - EPS;
    beta_safe = beta_safe && in_range(b, EPS, kBetaHi);
  }
  ammsb_seed rs = {0, 0};
  const bool noise_wave = wv == RW;
  if (a.noise_on && noise_wave) rs = a.seeds[(uint64_t)g * VL + VLn::vlane(ln)];

  for (uint64_t i = g; i < st.n_nodes; i += st.G) {
    const uint32_t node = a.nodes[i];
    __syncthreads();  // orders the LDS traffic of consecutive nodes
    for (uint32_t q = tid; q < n; q += T) {
      const uint32_t nbq = a.neighbors[i * n + q];
      const bool y = set_has(a.set, make_edge(node, nbq));
      s_nb[q] = nbq | (y ? 0x80000000u : 0u);
    }
    const float phi_sum = a.phi_sum[node];
    const float inv_phi_sum = 1.0f / phi_sum;
    const float* row_a = rpm_row(a.pi, node);
    float pi_a[KPT], rden[KPT];
    bool node_safe = beta_safe && in_range(phi_sum, kPhiSumLo, kPhiSumHi);
#pragma unroll
    for (int j = 0; j < KPT; ++j) pi_a[j] = row_a[ln + 64 * j];
    if (noise_wave) {
#pragma unroll
      for (int j = 0; j < KPT; ++j) s_pia[ln + 64 * j] = pi_a[j];
    } else {
#pragma unroll
      for (int j = 0; j < KPT; ++j) {
        const float den = pi_a[j] * phi_sum;
        rden[j] = exact_rcp(den);
        node_safe = node_safe && in_range(den, kDenLo, kDenHi);
      }
    }
    __syncthreads();  // s_nb complete

    if (!noise_wave) {
      // this wave's rows: q = wv, wv + RW, ...; the next row's loads are issued before the current row is reduced
      float cur[KPT], nxt[KPT];
      auto load_row = [&](float (&dst)[KPT], uint32_t q) {
        const uint32_t w = __builtin_amdgcn_readfirstlane(s_nb[q < n ? q : n - 1] & 0x7fffffffu);
        const float* row = rpm_row(a.pi, w);
#pragma unroll
        for (int j = 0; j < KPT; ++j) dst[j] = __builtin_nontemporal_load(row + ln + 64 * j);
      };
      if ((uint32_t)wv < n) load_row(cur, wv);
      for (uint32_t q = wv; q < n; q += RW) {
        load_row(nxt, q + RW);  // unconditional (clamped): the last trip re-requests the last row
        const bool y = __builtin_amdgcn_readfirstlane(s_nb[q] >> 31) != 0;
        const float e = y ? EPS : 1.0f - EPS;
        float pr[KPT];
        float partial = 0.0f, lo = 1.0f;
#pragma unroll
        for (int j = 0; j < KPT; ++j) {  // phi.cc:241-253 (columns ascending: j = 0, 1, ... is this lane's chain order)
          const float tt0 = cur[j] * bf[j];
          const float tt = y ? tt0 + e : e - tt0;
          pr[j] = pi_a[j] * tt;
          VLn::chain(partial, pr[j]);
          lo = fminf(lo, fabsf(pr[j]));
        }
        const float probs_sum = VLn::tree(partial);  // phi.cc:254-257
        float* tq = term + (size_t)q * K;
        // phi.cc:259-263: the row's contribution (probs / probs_sum) / (pi * phi_sum) - 1 / phi_sum
        if (node_safe && lo >= kProbsLo && in_range(probs_sum, kPsumLo, kPsumHi)) {
          const float rps = exact_rcp(probs_sum);
#pragma unroll
          for (int j = 0; j < KPT; ++j) {
            float qv = div_exact3(pr[j], probs_sum, rps);
            qv = div_exact3(qv, pi_a[j] * phi_sum, rden[j]);
            tq[ln + 64 * j] = qv - inv_phi_sum;
          }
        } else {
#pragma unroll
          for (int j = 0; j < KPT; ++j) {
            float qv = pr[j] / probs_sum;
            qv = qv / (pi_a[j] * phi_sum);
            tq[ln + 64 * j] = qv - inv_phi_sum;
          }
        }
#pragma unroll
        for (int j = 0; j < KPT; ++j) cur[j] = nxt[j];
      }
    } else if (a.noise_on) {
      // the node's normals: virtual lane's draw number j belongs to physical column j / PER of the lane that keeps it
#pragma unroll 1
      for (uint32_t j = 0; j < (uint32_t)KV; ++j) {
        const float z = rng_normal(rs, &zig);
        if (VLn::keeps(ln, j)) s_noise[ln + 64 * (j / VLn::PER)] = z;
      }
    }
    __syncthreads();  // all terms, the normals and pi_a are in LDS

    // columns tid, tid + T, ...: accumulate the n terms in row order (grads = ((0 + t_0) + t_1) + ...), then the SGLD
    // step of phi.cc:265-274
    float* out = a.phi_vec + i * K;
    const float half = st.eps_t / 2;
    for (int c = tid; c < K; c += T) {
      float grads = 0.0f;
      for (uint32_t q = 0; q < n; ++q) grads += term[(size_t)q * K + c];
      const float noise = a.noise_on ? s_noise[c] : 1.0f;
      const float phi_k = s_pia[c] * phi_sum;
      const float ng = a.Nn * grads;
      float in = a.alpha - phi_k;
      in = in + ng;
      const float drift = half * in;
      const float aa = phi_k + drift;
      const float ep = st.eps_t * phi_k;
      const float sq = sqrtf(ep);
      const float bb = sq * noise;
      const float v = fabsf(aa + bb);
      out[c] = v > 1e-24f ? v : 1e-24f;
    }
  }
  if (a.noise_on && noise_wave && ln < VL) a.seeds[(uint64_t)g * VL + ln] = rs;
}

template <int KPT, int RW, int VL>
int launch_phi_wide(ammsb_ctx* ctx, const PhiArgs& a, uint32_t n_groups, hipStream_t s) {
  const size_t lds = sizeof(float) * 64 * KPT * ((size_t)a.n + 2) + sizeof(uint32_t) * a.n;
  static const bool big = [] {  // more than the default 64 KiB of dynamic LDS: once per instantiation
    return hipFuncSetAttribute(reinterpret_cast<const void*>(update_phi_wide_kernel<KPT, RW, VL>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048) == hipSuccess;
  }();
  (void)big;
  static const std::string name = ammsb_kname("update_phi_wide_kernel<%d, %d, %d>", KPT, RW, VL);
  ctx->kernel_name[AMMSB_KN_PHI_SMALL] = name.c_str();  // (its own slot: a loop enqueues both forms for every step)
  update_phi_wide_kernel<KPT, RW, VL><<<n_groups, 64 * (RW + 1), lds, s>>>(a);
  AMMSB_LAUNCH_CHECK(ctx);
  return AMMSB_OK;
}

// ---------------------------------------------------------------------------------------------------------
// K = 256, large launches, OPT-IN (AMMSB_PHI_STREAM=1): persistent one-wave blocks that stream the neighbour rows of
// consecutive nodes as one sequence and fetch the next node's prologue while the current node's rows are being reduced.
//
// Why it was built: in-kernel stamps of update_phi_lds2_kernel<4, 8, 4> under load
// (profiles/r03_phi_k256_trace_loaded.txt) show a node taking ~42 000 cycles of which 9 400 are prologue -- node id ->
// neighbour ids -> edge-set probes / own row / phi_sum, three dependent round trips, plus the ziggurat tables and the
// beta row loaded once per NODE because a block is a node -- with nothing in flight for the wave meanwhile.  Here a
// block walks several virtual groups (g = b, b + B, ...; B = what the chip holds), the tables and beta are loaded once
// per block, and during node k's row loop the wave issues node k+1's prologue in stages (ids; probe bins; probe test
// and link flags into the other half of a double-buffered id list; own row, phi_sum, stream state) and requests node
// k+1's first rows in node k's last step: the row stream never drains.  Virtual group g still owns stream g L + l and
// node(s) g (, g + G): arithmetic, WG_SUM order and draws are update_phi_lds2_kernel's, bit for bit (tests).
// The counted vmcnt waits stay valid with the extra requests around: every prologue request is issued AFTER a step's
// wait and before the next step's row requests, i.e. it is OLDER than everything a later wait leaves outstanding.
//
// What was measured (profiles/r03_c2_stream_ab.log, r03_phi_k256_stream_trace.txt): the stamps confirm the design --
// no step waits for rows any more and the staged prologue costs 300 - 700 cycles of issue per stage -- but the kernel
// is SLOWER than the one it was to replace, 82 against 69 us at C2 (wg 32: 91 against 70).  Its staging area and the
// registers that carry a second node's state leave 10 - 11 blocks per CU where update_phi_lds2_kernel has 16, and at
// K = 256 the launch is not waiting for memory in the first place: the SQ counters put the VALU at 74 % busy while all
// 16 waves of a CU are resident (2 611 VALU instructions per node, profiles/r03_c2_pmc_sq.txt), so what the prologue
// overlap buys is paid back by the lost waves.  Kept opt-in and tested; not dispatched by default.
template <int KPT, int D, int U, int VL>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3))) void update_phi_stream_kernel(const PhiArgs a) {
  constexpr int L = 64, KW = 64 * KPT, K = L * KPT, PIECES = KPT / 4, HP = KPT / 2;
  using VLn = VLane<VL>;
  constexpr int KV = KPT * VLn::PER;  // normals per virtual lane
  static_assert(KPT == U && D == 2 * U && PIECES == 1, "K = 256: four rows per step, two steps in the ring");
  // LDS: [D][KW] ring | [2][64] u32 id lists (id | link bit) | staging of the NEXT node's prologue, all of it filled by
  // LDS-DMA: [64] neighbour ids, [2][64] node id (64 copies), then 4 KiB that hold the four 16-byte halves of the probes'
  // two bins ([4][64][16 B]) and later, once those have been tested, the node's own row (1 KiB), its group's stream
  // states (1 KiB) and phi_sum (64 copies).
  // Why LDS-DMA for a few dozen bytes: a plain load inside this loop makes hipcc's wait-count pass emit vmcnt(0) at
  // the load's first use (it cannot count across the loop's branches), which drains the row requests in flight -- the
  // first form of this kernel had seven such drains per node and was slower than the kernel it replaces.  DMA writes
  // have no register result; their arrival is covered by the loop's own counted waits (they are always older than the
  // requests a wait leaves outstanding) and they are read back with ordinary LDS loads.
  extern __shared__ __align__(16) char smem[];
  __shared__ ZigTables zig;
  const int tid = threadIdx.x;
  float* ring = reinterpret_cast<float*>(smem);
  uint32_t* s_nb_all = reinterpret_cast<uint32_t*>(smem + (size_t)D * KW * sizeof(float));
  uint32_t* st_ids = s_nb_all + 128;
  uint32_t* st_node = st_ids + 64;
  char* st_big = reinterpret_cast<char*>(st_node + 128);  // 4 KiB, 16-byte aligned (st_node is [2][64]: current, next)
  float* st_row = reinterpret_cast<float*>(st_big);                          // [KW] the next node's own pi row
  ammsb_seed* st_seeds = reinterpret_cast<ammsb_seed*>(st_big + 1024);       // [64] its group's stream states
  float* st_phisum = reinterpret_cast<float*>(st_big + 2048);                // [64] copies of its phi_sum
  auto dma4 = [&](const void* src_lane, void* dst_uniform) {
    __builtin_amdgcn_global_load_lds((glb_void_t*)src_lane, (lds_void_t*)dst_uniform, 4, 0, 0);
  };
  auto dma16 = [&](const void* src_lane, void* dst_uniform) {
    __builtin_amdgcn_global_load_lds((glb_void_t*)src_lane, (lds_void_t*)dst_uniform, 16, 0, 0);
  };

  const PhiStep st = phi_step(a);
  if (st.n_nodes == 0) return;  // (uniform) a skipped step
  note_stamp(a.stamps, a.desc, AMMSB_STAMP_PHI);
  const uint32_t B = gridDim.x, n = a.n, NIT = n / U;
  PHI_TRACE(0);
  const uint32_t g0 = a.group_begin + blockIdx.x;
  if (g0 >= st.group_end) return;  // block-uniform
  const float EPS = a.epsilon;
  if (a.noise_on) zig_load(&zig);

  f32x2 bf[HP];
  bool beta_safe = true;
#pragma unroll
  for (int p = 0; p < HP; ++p) {
    const float b0 = a.beta[2 * (tid + 2 * L * p) + 1];
    const float b1 = a.beta[2 * (tid + 2 * L * p + L) + 1];
    bf[p] = f32x2{b0 - EPS, b1 - EPS};
    beta_safe = beta_safe && in_range(b0, EPS, kBetaHi) && in_range(b1, EPS, kBetaHi);
  }

  // rows q0 .. q0+U-1 of the id list `nb` into the ring half `half`
  auto request = [&](const uint32_t* nb, uint32_t q0, uint32_t half) {
#pragma unroll
    for (int r = 0; r < U; ++r) {
      const uint32_t nbr = __builtin_amdgcn_readfirstlane(nb[q0 + r] & 0x7fffffffu);
      const float* src = rpm_row(a.pi, nbr) + 4 * tid;
      __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(smem + (size_t)(half * U + r) * (KW * sizeof(float))), 16, 0, 2);
    }
  };
  auto wave_sync = [&]() {  // one wave: its LDS operations complete in order
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  };

  // ---- the first node of this block: plain prologue, parked in the staging area like every later node's
  uint32_t cur_g = g0;
  uint64_t cur_i = g0;
  uint32_t sb = 0, gg = 0;  // id-list buffer of the current node; running count of row groups (ring half = gg & 1)
  {
    const uint32_t node0 = a.nodes[cur_i];
    __syncthreads();  // the ziggurat tables
    uint32_t nb = 0;
    if ((uint32_t)tid < n) {
      nb = a.neighbors[cur_i * n + tid];
      if (set_has(a.set, make_edge(node0, nb))) nb |= 0x80000000u;
    }
    s_nb_all[tid] = nb;
    st_node[tid] = node0;
    st_phisum[tid] = a.phi_sum[node0];
    const float* row_a = rpm_row(a.pi, node0);
#pragma unroll
    for (int j = 0; j < KPT; ++j) st_row[tid + 64 * j] = __builtin_nontemporal_load(row_a + tid + 64 * j);
    ammsb_seed rs0 = {0, 0};
    if (a.noise_on) rs0 = a.seeds[(uint64_t)cur_g * VL + VLn::vlane(tid)];
    st_seeds[tid] = rs0;
  }
  wave_sync();
  request(s_nb_all, 0, 0);
  PHI_TRACE(1);
  uint32_t trace_node = 0;

  // A node's results are stored one step late -- behind the first wait of the NEXT node's row loop -- so that no
  // counted wait ever has stores of unknown age in front of it (vmcnt counts stores and retires in order: stores
  // issued between two row requests would have to be waited for before the later rows count as landed).
  bool pend = false, pend_seeds = false;
  float pend_v[KPT];
  uint64_t pend_i = 0;
  uint32_t pend_g = 0;
  ammsb_seed pend_rs = {0, 0};
#pragma unroll
  for (int j = 0; j < KPT; ++j) pend_v[j] = 0.0f;
  auto flush = [&]() {
    if (!pend) return;
    float* out = a.phi_vec + pend_i * K;
#pragma unroll
    for (int p = 0; p < HP; ++p) {
      __builtin_nontemporal_store(pend_v[2 * p], out + tid + 2 * L * p);
      __builtin_nontemporal_store(pend_v[2 * p + 1], out + tid + 2 * L * p + L);
    }
    if (pend_seeds && tid < VL) a.seeds[(uint64_t)pend_g * VL + tid] = pend_rs;
    pend = false;
  };

  ammsb_seed rs = {0, 0};
  bool take_seeds = true;  // the staged stream states belong to a group this block is entering
  for (;;) {
    // the item after this one: the group's second node (more nodes than groups), else the block's next group
    uint32_t nx_g = cur_g;
    uint64_t nx_i = cur_i + st.G;
    if (nx_i >= st.n_nodes) {
      nx_g = cur_g + B;
      nx_i = nx_g;
    }
    const bool nx_valid = nx_g < st.group_end && nx_i < st.n_nodes;
    const uint32_t* s_nb = s_nb_all + sb * 64;
    float phi_sum = 1.0f, inv_phi_sum = 1.0f;
    f32x2 pi_a[HP], grads[HP], rden[HP];
    float nz[KPT];
    bool node_safe = false;
#pragma unroll
    for (int p = 0; p < HP; ++p) {
      pi_a[p] = grads[p] = rden[p] = f32x2{0.0f, 0.0f};
      nz[2 * p] = nz[2 * p + 1] = 0.0f;
    }
    const uint32_t sB = NIT / 2 - 1, sC = NIT - 2;  // (NIT >= 4: dispatch)
    uint64_t nx_key = 0;
    uint32_t nx_nb = 0;

    for (uint32_t it = 0; it < NIT; ++it, ++gg) {
      const uint32_t q = it * U;
      float* row[U];
#pragma unroll
      for (int r = 0; r < U; ++r) row[r] = ring + ((gg & 1u) * U + r) * KW;
      // every LDS read of the previous row group has been consumed; its half of the ring takes the next group:
      // the current node's next rows, or -- in its last step -- the first rows of the next node
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (trace_node < 3) PHI_TRACE(8 + 32 * trace_node + 3 * it);
      bool more = true;
      if (it + 1 < NIT) request(s_nb, q + U, (gg + 1) & 1u);
      else if (nx_valid) request(s_nb_all + (sb ^ 1u) * 64, 0, (gg + 1) & 1u);
      else more = false;
      // one of the node's normals per step, drawn while the rows are on their way (stream order = ascending column);
      // the first step's draw waits until the stream state has been read from the staging area (below)
      float z_now = 0.0f;
      if (a.noise_on && it >= 1 && it < (uint32_t)KV) z_now = rng_normal(rs, &zig);
      if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(U * PIECES) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (trace_node < 3) PHI_TRACE(8 + 32 * trace_node + 3 * it + 1);

      if (it == 0) {
        flush();  // the previous node's row (and stream state)
        // this node's own data, staged by the previous node's last step (or by the block's prologue)
        phi_sum = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(st_phisum[0])));
        inv_phi_sum = 1.0f / phi_sum;
        node_safe = beta_safe && in_range(phi_sum, kPhiSumLo, kPhiSumHi);
#pragma unroll
        for (int p = 0; p < HP; ++p) {
          pi_a[p] = f32x2{st_row[tid + 2 * L * p], st_row[tid + 2 * L * p + L]};
          const f32x2 den = pi_a[p] * phi_sum;
          rden[p] = f32x2{exact_rcp(den.x), exact_rcp(den.y)};
          node_safe = node_safe && in_range(den.x, kDenLo, kDenHi) && in_range(den.y, kDenLo, kDenHi);
          const f32x2 ep = den * st.eps_t;
          nz[2 * p] = sqrtf(ep.x);
          nz[2 * p + 1] = sqrtf(ep.y);
        }
        if (take_seeds) rs = st_seeds[tid];
        if (a.noise_on && 0 < KV) z_now = rng_normal(rs, &zig);
      }
      if (a.noise_on && it < (uint32_t)KV && VLn::keeps(tid, it)) {
#pragma unroll
        for (int j = 0; j < KPT; ++j)
          if ((uint32_t)j == it / VLn::PER) nz[j] = nz[j] * z_now;
      }

      // ---- the next node's prologue, a stage at a time, all through LDS-DMA (see the top of the kernel)
      if (nx_valid) {
        if (it == 0) {  // its neighbour ids and its node id
          dma4(a.neighbors + nx_i * n + ((uint32_t)tid < n ? (uint32_t)tid : n - 1), st_ids);
          dma4(a.nodes + nx_i, st_node + 64);  // (the other half of [2][64]: the current node's id stays readable)
        }
        if (it == sB) {  // the four 16-byte halves of the two bins each lane's probe looks at
          const uint32_t nx_node = __builtin_amdgcn_readfirstlane(st_node[64]);
          nx_nb = st_ids[tid];
          nx_key = make_edge(nx_node, nx_nb);
          const uint64_t h1 = fast_mod(kSetPrimes[2 * a.set.prime_idx] * nx_key, a.set.mod);
          const uint64_t h2 = fast_mod(nx_key ^ kSetPrimes[2 * a.set.prime_idx + 1], a.set.mod);
          const uint64_t* b1 = a.set.slots + h1 * 4;
          const uint64_t* b2 = a.set.slots + (a.set.num_bins + h2) * 4;
          dma16(b1, st_big);
          dma16(b1 + 2, st_big + 1024);
          dma16(b2, st_big + 2048);
          dma16(b2 + 2, st_big + 3072);
        }
        if (it == sC) {  // test them; the id list of the next node is complete
          // (read with explicit instructions: hipcc puts a vmcnt(0) in front of an LDS load it can see aliasing an
          // LDS-DMA target, which here would drain the row requests; the counted wait above already covers the probes)
          ulonglong2 pv[4];
          const uint32_t lds_at = (uint32_t)(uintptr_t)(lds_void_t*)(st_big + 16 * tid);
          asm volatile(
              "ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\t"
              "ds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
              : "=&v"(pv[0]), "=&v"(pv[1]), "=&v"(pv[2]), "=&v"(pv[3])
              : "v"(lds_at)
              : "memory");
          bool hit = false;
#pragma unroll
          for (int w = 0; w < 4; ++w) hit = hit || pv[w].x == nx_key || pv[w].y == nx_key;
          uint32_t v = 0;
          if ((uint32_t)tid < n) v = nx_nb | (hit ? 0x80000000u : 0u);
          s_nb_all[(sb ^ 1u) * 64 + tid] = v;  // read by the request of the last step (LDS operations of a wave are ordered)
        }
        if (it == NIT - 1) {  // its own row, phi_sum and stream states take the probe area (tested two steps ago)
          const uint32_t nx_node = __builtin_amdgcn_readfirstlane(st_node[64]);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every read of the area has returned
          dma16(rpm_row(a.pi, nx_node) + 4 * tid, st_row);
          dma4(a.phi_sum + nx_node, st_phisum);
          if (a.noise_on && nx_g != cur_g) dma16(a.seeds + (uint64_t)nx_g * VL + VLn::vlane(tid), st_seeds);
        }
      }

      float ee[U], sg[U], part[U], lo[U], psum[U];
      f32x2 prr[U][HP];
      bool fast[U], all_fast = true;
#pragma unroll
      for (int r = 0; r < U; ++r) {
        const bool y = (__builtin_amdgcn_readfirstlane(s_nb[q + r]) >> 31) != 0;
        ee[r] = y ? EPS : 1.0f - EPS;
        sg[r] = y ? 1.0f : -1.0f;  // e - pin (beta - EPS) == e + (-(pin (beta - EPS))): the negation is exact
        part[r] = 0.0f;
        lo[r] = 1.0f;
      }
      // pass 1 of all U rows (phi.cc:241-253): probs[], lane partials in ascending column order
#pragma unroll
      for (int p = 0; p < HP; ++p) {
        float vx[U], vy[U];
#pragma unroll
        for (int r = 0; r < U; ++r) {
          const f32x2 pin = f32x2{row[r][tid + 128 * p], row[r][tid + 128 * p + 64]};
          const f32x2 tt = (pin * bf[p]) * sg[r] + ee[r];
          const f32x2 pr = pi_a[p] * tt;
          prr[r][p] = pr;
          vx[r] = pr.x;
          vy[r] = pr.y;
          lo[r] = fminf(fminf(lo[r], fabsf(pr.x)), fabsf(pr.y));
        }
        VLn::template chain_rows<U>(part, vx);  // all U rows' chains advance by the column pair (ammsb_dev.h)
        VLn::template chain_rows<U>(part, vy);
      }
      VLn::template tree_rows<U>(part, psum);  // phi.cc:254-257
#pragma unroll
      for (int r = 0; r < U; ++r) {
        fast[r] = node_safe && lo[r] >= kProbsLo && in_range(psum[r], kPsumLo, kPsumHi);
        all_fast = all_fast && fast[r];
      }
      // pass 2 (phi.cc:259-263): grads += (probs / probs_sum) / (pi * phi_sum) - 1 / phi_sum, rows in order
      if (all_fast) {
        float ps = phi_sum;
        asm volatile("" : "+v"(ps));  // keeps pi_a * phi_sum from being hoisted into KPT more registers
        f32x2 s2[U], r2[U];
#pragma unroll
        for (int r = 0; r < U; ++r) {
          const float rc = exact_rcp(psum[r]);
          s2[r] = f32x2{psum[r], psum[r]};
          r2[r] = f32x2{rc, rc};
        }
#pragma unroll
        for (int p = 0; p < HP; ++p) {
          const f32x2 den = pi_a[p] * ps;
          f32x2 v[U];
#pragma unroll
          for (int r = 0; r < U; ++r) v[r] = div_exact3(div_exact3(prr[r][p], s2[r], r2[r]), den, rden[p]);
#pragma unroll
          for (int r = 0; r < U; ++r) grads[p] += v[r] - inv_phi_sum;
        }
      } else {
#pragma unroll
        for (int r = 0; r < U; ++r) {
          const float probs_sum = psum[r];
          if (fast[r]) {
            const float rps = exact_rcp(probs_sum);
            const f32x2 psum2 = f32x2{probs_sum, probs_sum}, rps2 = f32x2{rps, rps};
#pragma unroll
            for (int p = 0; p < HP; ++p) {
              f32x2 qv = div_exact3(prr[r][p], psum2, rps2);
              qv = div_exact3(qv, pi_a[p] * phi_sum, rden[p]);
              grads[p] += qv - inv_phi_sum;
            }
          } else {
#pragma unroll
            for (int p = 0; p < HP; ++p) {
              const f32x2 den = pi_a[p] * phi_sum;
              float v0 = prr[r][p].x / probs_sum, v1 = prr[r][p].y / probs_sum;
              v0 = v0 / den.x;
              v1 = v1 / den.y;
              grads[p] += f32x2{v0 - inv_phi_sum, v1 - inv_phi_sum};
            }
          }
        }
      }
    }
    if (trace_node < 3) PHI_TRACE(8 + 32 * trace_node + 3 * NIT);
    ++trace_node;
    // normals the loop did not get to (fewer steps than normals per virtual lane)
    if (a.noise_on) {
#pragma unroll 1
      for (uint32_t j = NIT; j < (uint32_t)KV; ++j) {
        const float z = rng_normal(rs, &zig);
        if (VLn::keeps(tid, j)) {
#pragma unroll
          for (int c = 0; c < KPT; ++c)
            if ((uint32_t)c == j / VLn::PER) nz[c] = nz[c] * z;
        }
      }
    }

    // SGLD step, phi.cc:265-274 (parked: stored behind the next node's first wait, or below if this was the last)
    const float half = st.eps_t / 2;
#pragma unroll
    for (int p = 0; p < HP; ++p) {
      const f32x2 bb = f32x2{nz[2 * p], nz[2 * p + 1]};
      const f32x2 phi_k = pi_a[p] * phi_sum;
      const f32x2 ng = grads[p] * a.Nn;
      f32x2 in = a.alpha - phi_k;
      in = in + ng;
      const f32x2 drift = in * half;
      const f32x2 aa = phi_k + drift;
      const f32x2 s2 = aa + bb;
      const float v0 = fabsf(s2.x), v1 = fabsf(s2.y);
      pend_v[2 * p] = v0 > 1e-24f ? v0 : 1e-24f;
      pend_v[2 * p + 1] = v1 > 1e-24f ? v1 : 1e-24f;
    }
    pend = true;
    pend_i = cur_i;
    // the group's stream goes back when the block leaves the group
    pend_seeds = a.noise_on && (!nx_valid || nx_g != cur_g);
    if (pend_seeds) {
      pend_g = cur_g;
      pend_rs = rs;
    }
    take_seeds = nx_g != cur_g;
    if (!nx_valid) {
      flush();
      break;
    }
    cur_g = nx_g;
    cur_i = nx_i;
    sb ^= 1u;
  }
}

template <int KPT, int D, int U, int VL>
int launch_phi_stream(ammsb_ctx* ctx, const PhiArgs& a, uint32_t n_groups, hipStream_t s) {
  const size_t lds = (size_t)D * sizeof(float) * 64 * KPT + 5 * 64 * sizeof(uint32_t) + 4096;  // ring, id lists, staging
  static const int per_cu = [] {  // resident one-wave blocks per CU (registers, LDS): the persistent grid is that times the CUs
    int v = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, update_phi_stream_kernel<KPT, D, U, VL>, 64,
                                                     (size_t)D * sizeof(float) * 64 * KPT + 5 * 64 * sizeof(uint32_t) + 4096) != hipSuccess || v < 1)
      v = 8;
    return v;
  }();
  // as many blocks as the chip holds -- fewer if that makes every block walk the same number of groups (8193 groups on
  // 3072 slots: 2731 blocks of three, not 3072 of which a third idles through the last round)
  const uint32_t max_blocks = (uint32_t)per_cu * (uint32_t)ctx->num_cus;
  const uint32_t rounds = (n_groups + max_blocks - 1) / max_blocks;
  uint32_t blocks = (n_groups + rounds - 1) / rounds;
  static const std::string name = ammsb_kname("update_phi_stream_kernel<%d, %d, %d, %d>", KPT, D, U, VL);
  ctx->kernel_name[AMMSB_KN_PHI] = name.c_str();
  update_phi_stream_kernel<KPT, D, U, VL><<<blocks, 64, lds, s>>>(a);
  AMMSB_LAUNCH_CHECK(ctx);
  return AMMSB_OK;
}

template <int KPT, int D, int U, int VL = 64>
int launch_phi_lds2(ammsb_ctx* ctx, const PhiArgs& a, uint32_t n_groups, hipStream_t s) {
  const size_t lds = (size_t)(D + (KPT == U ? 0 : 1)) * sizeof(float) * 64 * KPT + sizeof(uint32_t) * a.n;
  const bool one = a.pi.num_blocks == 1;
  static const std::string name1 = ammsb_kname("update_phi_lds2_kernel<%d, %d, %d, %d, true>", KPT, D, U, VL);
  static const std::string name0 = ammsb_kname("update_phi_lds2_kernel<%d, %d, %d, %d, false>", KPT, D, U, VL);
  ctx->kernel_name[AMMSB_KN_PHI] = (one ? name1 : name0).c_str();
  // AMMSB_PHI_PERSIST=1|2|3 (A/B runs): a persistent grid for launches of more groups than the chip holds at once
  // (1: resident slots when the excess over whole rounds is small, else the groups spread evenly over the fewest rounds;
  // 2: always the resident slots; 3: always the even spread).  Measured equal or slightly slower than one block per
  // group at C2 (69.6 - 71.3 against 68.5 - 70.2 us): not the default.
  static const int mode = [] {
    const char* f = getenv("AMMSB_PHI_PERSIST");
    return f ? atoi(f) : 0;
  }();
  static const int per_cu = [] {
    int v = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, update_phi_lds2_kernel<KPT, D, U, VL, true>, 64,
                                                     (size_t)(D + (KPT == U ? 0 : 1)) * sizeof(float) * 64 * KPT + 128) != hipSuccess || v < 1)
      v = 0;
    return v;
  }();
  uint32_t grid = n_groups;
  const uint64_t slots = (uint64_t)per_cu * (uint64_t)ctx->num_cus;
  if (mode > 0 && slots > 0 && n_groups > slots) {
    const uint64_t rounds = (n_groups + slots - 1) / slots;
    grid = (uint32_t)((n_groups + rounds - 1) / rounds);
    if (mode == 2 || (mode == 1 && n_groups - (rounds - 1) * slots <= slots / 16)) grid = (uint32_t)slots;
  }
  static const size_t pad = [] {  // AMMSB_PHI_LDS_PAD=<bytes>: extra LDS per block, i.e. fewer resident blocks per CU (experiments)
    const char* f = getenv("AMMSB_PHI_LDS_PAD");
    return f ? (size_t)atoi(f) : (size_t)0;
  }();
  if (one) update_phi_lds2_kernel<KPT, D, U, VL, true><<<grid, 64, lds + pad, s>>>(a);
  else update_phi_lds2_kernel<KPT, D, U, VL, false><<<grid, 64, lds + pad, s>>>(a);
  AMMSB_LAUNCH_CHECK(ctx);
  return AMMSB_OK;
}

template <int KPT, int W, int D = 2, int NB = 1, int VL = 64>
int launch_phi_lds(ammsb_ctx* ctx, const PhiArgs& a, uint32_t n_groups, hipStream_t s) {
  const size_t per_node = (size_t)W * (D + 1) * sizeof(float) * 64 * KPT + ((sizeof(uint32_t) * a.n + 15) & ~(size_t)15);
  const bool one = a.pi.num_blocks == 1;
  static const std::string name1 = ammsb_kname("update_phi_lds_kernel<%d, %d, %d, %d, %d, true>", KPT, W, D, NB, VL);
  static const std::string name0 = ammsb_kname("update_phi_lds_kernel<%d, %d, %d, %d, %d, false>", KPT, W, D, NB, VL);
  ctx->kernel_name[AMMSB_KN_PHI] = (one ? name1 : name0).c_str();
  if (one) update_phi_lds_kernel<KPT, W, D, NB, VL, true><<<(n_groups + NB - 1) / NB, 64 * W * NB, per_node * NB, s>>>(a);
  else update_phi_lds_kernel<KPT, W, D, NB, VL, false><<<(n_groups + NB - 1) / NB, 64 * W * NB, per_node * NB, s>>>(a);
  AMMSB_LAUNCH_CHECK(ctx);
  return AMMSB_OK;
}

template <int KPT, int VL = 64>
int launch_phi_lds3(ammsb_ctx* ctx, const PhiArgs& a, uint32_t n_groups, hipStream_t s) {
  const size_t lds = (size_t)3 * sizeof(float) * 64 * KPT + ((sizeof(uint32_t) * a.n + 15) & ~(size_t)15);
  const bool one = a.pi.num_blocks == 1;
  static const std::string name1 = ammsb_kname("update_phi_lds3_kernel<%d, %d, true>", KPT, VL);
  static const std::string name0 = ammsb_kname("update_phi_lds3_kernel<%d, %d, false>", KPT, VL);
  ctx->kernel_name[AMMSB_KN_PHI] = (one ? name1 : name0).c_str();
  if (one) update_phi_lds3_kernel<KPT, VL, true><<<n_groups, 64, lds, s>>>(a);
  else update_phi_lds3_kernel<KPT, VL, false><<<n_groups, 64, lds, s>>>(a);
  AMMSB_LAUNCH_CHECK(ctx);
  return AMMSB_OK;
}

// update_pi, phi.cc:177-197: copy phi_vec row into pi, WG_NORMALIZE it, phi_sum[node] = sum.
template <int L, int KPT>
__global__ __launch_bounds__(Group<L>::BLOCK) void update_pi_kernel(ammsb_rpm pi, float* phi_sum,
                                                                     const float* phi_vec, const uint32_t* nodes,
                                                                     uint32_t n_nodes, uint32_t K,
                                                                     const ammsb_step_desc* desc,
                                                                     unsigned long long* stamps) {
  using Grp = Group<L>;
  __shared__ float aux[Grp::AUX];
  if (desc) n_nodes = desc->n_nodes;  // captured graph: the grid covers the largest mini-batch
  note_stamp(stamps, desc, AMMSB_STAMP_PI);
  const int l = Grp::lane();
  const uint64_t i = (uint64_t)blockIdx.x * Grp::PER_BLOCK + Grp::slot();
  const bool on = i < n_nodes;
  const float* src = phi_vec + i * K;
  float v[KPT];
  float partial = 0.0f;
#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    const uint32_t k = l + j * L;
    v[j] = (on && k < K) ? src[k] : 0.0f;
    partial += v[j];
  }
  int phase = 0;
  const float sum = Grp::sum(partial, aux, phase);
  if (on) {
    const uint32_t node = nodes[i];
    float* row = rpm_row(pi, node);
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      const uint32_t k = l + j * L;
      if (k < K) row[k] = v[j] / sum;
    }
    if (l == 0) phi_sum[node] = sum;
  }
}

template <int L, int KPT>
int launch_phi(ammsb_ctx* ctx, const PhiArgs& a, uint32_t n_groups, hipStream_t s) {
  using Grp = Group<L>;
  constexpr int DEPTH = KPT >= 32 ? 2 : 4;
  const uint32_t blocks = (n_groups + Grp::PER_BLOCK - 1) / Grp::PER_BLOCK;
  const size_t lds = sizeof(uint32_t) * Grp::PER_BLOCK * a.n;
  if constexpr (KPT <= 2 && (L == 32 || L == 64)) {  // the latency-bound shapes (C1): the single-block form pays there
    if (a.pi.num_blocks == 1) {
      static const std::string name_full1 = ammsb_kname("update_phi_kernel<%d, %d, %d, true, true>", L, KPT, DEPTH);
      static const std::string name_part1 = ammsb_kname("update_phi_kernel<%d, %d, %d, false, true>", L, KPT, DEPTH);
      ctx->kernel_name[AMMSB_KN_PHI] = (a.K == (uint32_t)(L * KPT) ? name_full1 : name_part1).c_str();
      if (a.K == (uint32_t)(L * KPT))
        update_phi_kernel<L, KPT, DEPTH, true, true><<<blocks, Grp::BLOCK, lds, s>>>(a);
      else
        update_phi_kernel<L, KPT, DEPTH, false, true><<<blocks, Grp::BLOCK, lds, s>>>(a);
      AMMSB_LAUNCH_CHECK(ctx);
      return AMMSB_OK;
    }
  }
  static const std::string name_full = ammsb_kname("update_phi_kernel<%d, %d, %d, true, false>", L, KPT, DEPTH);
  static const std::string name_part = ammsb_kname("update_phi_kernel<%d, %d, %d, false, false>", L, KPT, DEPTH);
  ctx->kernel_name[AMMSB_KN_PHI] = (a.K == (uint32_t)(L * KPT) ? name_full : name_part).c_str();
  if (a.K == (uint32_t)(L * KPT))
    update_phi_kernel<L, KPT, DEPTH, true><<<blocks, Grp::BLOCK, lds, s>>>(a);
  else
    update_phi_kernel<L, KPT, DEPTH, false><<<blocks, Grp::BLOCK, lds, s>>>(a);
  AMMSB_LAUNCH_CHECK(ctx);
  return AMMSB_OK;
}

template <int L, int KPT>
int launch_pi(ammsb_ctx* ctx, const ammsb_rpm& pi, float* phi_sum, const float* phi_vec, const uint32_t* nodes,
              uint32_t n_nodes, uint32_t K, const ammsb_step_desc* desc, unsigned long long* stamps, hipStream_t s) {
  using Grp = Group<L>;
  const uint32_t blocks = (n_nodes + Grp::PER_BLOCK - 1) / Grp::PER_BLOCK;
  static const std::string name = ammsb_kname("update_pi_kernel<%d, %d>", L, KPT);
  ctx->kernel_name[AMMSB_KN_PI] = name.c_str();
  update_pi_kernel<L, KPT><<<blocks, Grp::BLOCK, 0, s>>>(pi, phi_sum, phi_vec, nodes, n_nodes, K, desc, stamps);
  AMMSB_LAUNCH_CHECK(ctx);
  return AMMSB_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Generic form: any K <= 1024 * (blockDim.x / 64), any power-of-two reference work-group size L <= blockDim.x.
//
// The reference loops K_PER_THREAD = ceil(K / L) columns per work-item generically (phi.cc:214-302), so e.g. its
// default phi_wg_size = 32 (main.cc:61) is valid at K = 4096: 128 columns per work-item, which no register- or
// LDS-resident per-lane layout above holds.  Here the elementwise work of a node is spread over all T = blockDim.x
// threads (thread t owns columns t + T i, at most CPT of them) independently of L; what L fixes -- the WG_SUM
// association order and which stream draws the noise of which column -- is emulated lane by lane: vgroup_sum()
// (ammsb_dev.h) for the sum, and for the noise thread l < L owns stream g L + l and draws for columns l, l + L, ...
// in ascending order (phi.cc:266-274, 291, 300).  Same operations in the same order on every value as
// update_phi_kernel<L, KPT>: bit-identical to it (and to the oracle) wherever both run.
// The noise.  Stream g L + l draws the normals of columns l, l + L, ... of group g's nodes one after the other
// (phi.cc:266-274): K / L sequential ziggurat draws per node and lane -- 128 at K = 4096, wg 32, about 1000 cycles each
// for a wave, on 32 of a block's 512 threads: more than half of a node's time when drawn inside update_phi_gen_kernel
// (C5-sized launch: 27 ms of which ~15 ms draws).  But the draws depend on nothing except the stream: phi_noise_kernel
// makes them first, ONE THREAD PER STREAM over the whole launch (2 M streams side by side), into the phi_vec rows the
// update is about to overwrite; update_phi_gen_kernel reads its noise from there.  Same streams, same order.
__global__ __launch_bounds__(256) void phi_noise_kernel(const PhiArgs a, uint32_t L, uint32_t lgL) {
  __shared__ ZigTables zig;
  const PhiStep st = phi_step(a);
  if (st.n_nodes == 0) return;  // (uniform) a skipped step
  note_stamp(a.stamps, a.desc, AMMSB_STAMP_PHI);
  zig_load(&zig);
  __syncthreads();
  const uint64_t sid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;  // stream within the launch's group range
  const uint32_t g = a.group_begin + (uint32_t)(sid >> lgL), l = (uint32_t)sid & (L - 1);
  if (g >= st.group_end) return;
  ammsb_seed rs = a.seeds[(uint64_t)g * L + l];
  for (uint64_t i = g; i < st.n_nodes; i += st.G) {
    float* out = a.phi_vec + i * a.K;
    for (uint32_t k = l; k < a.K; k += L) out[k] = rng_normal(rs, &zig);
  }
  a.seeds[(uint64_t)g * L + l] = rs;
}

// U neighbour rows go through one barrier phase together (n % U == 0, U * L <= blockDim.x): their U WG_SUM chains
// run side by side on U * L threads (vgroup_sum<U>) -- at L = 32 a lone chain leaves 15 of a block's 16 waves idle
// for 128 dependent adds per row.
template <int CPT, int U>
__global__ __launch_bounds__(512) void update_phi_gen_kernel(const PhiArgs a, uint32_t L, uint32_t lgL) {
  extern __shared__ __align__(16) char smem[];  // [U][K] probs, [U L] lane partials, [2 U] sums, [n] u32
  const uint32_t K = a.K, n = a.n, T = blockDim.x, t = threadIdx.x;
  float* s_vals = reinterpret_cast<float*>(smem);
  float* s_aux = s_vals + (size_t)U * K;
  float* s_res = s_aux + U * L;
  uint32_t* s_nb = reinterpret_cast<uint32_t*>(s_res + 2 * U);

  const PhiStep st = phi_step(a);
  if (!a.noise_on) note_stamp(a.stamps, a.desc, AMMSB_STAMP_PHI);  // (with noise the step starts in phi_noise_kernel)
  const uint32_t g = a.group_begin + blockIdx.x;
  if (g >= st.group_end) return;  // block-uniform
  const float EPS = a.epsilon;

  auto col = [&](int j) -> uint32_t { return t + (uint32_t)j * T; };
  auto ccol = [&](int j) -> uint32_t {
    const uint32_t k = t + (uint32_t)j * T;
    return k < K ? k : K - 1;
  };
  auto has = [&](int j) -> bool { return t + (uint32_t)j * T < K; };

  float bf[CPT];
  bool beta_safe = true;
#pragma unroll
  for (int j = 0; j < CPT; ++j) {
    const float b = a.beta[2 * ccol(j) + 1];
    bf[j] = b - EPS;
    beta_safe = beta_safe && in_range(b, EPS, kBetaHi);
  }
  int phase = 0;
  for (uint64_t i = g; i < st.n_nodes; i += st.G) {
    const uint32_t node = a.nodes[i];
    __syncthreads();  // orders the LDS traffic of consecutive nodes
    for (uint32_t q = t; q < n; q += T) {
      const uint32_t nbq = a.neighbors[i * n + q];
      const bool y = set_has(a.set, make_edge(node, nbq));
      s_nb[q] = nbq | (y ? 0x80000000u : 0u);
    }
    __syncthreads();

    const float phi_sum = a.phi_sum[node];
    const float inv_phi_sum = 1.0f / phi_sum;
    const float* row_a = rpm_row(a.pi, node);
    float* out = a.phi_vec + i * K;
    float pi_a[CPT], grads[CPT], rden[CPT], noise[CPT];
    bool node_safe = beta_safe && in_range(phi_sum, kPhiSumLo, kPhiSumHi);
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      const float x = row_a[ccol(j)];
      pi_a[j] = has(j) ? x : 0.0f;
      grads[j] = 0.0f;
      const float den = pi_a[j] * phi_sum;
      rden[j] = exact_rcp(den);
      node_safe = node_safe && (in_range(den, kDenLo, kDenHi) || !has(j));
      noise[j] = a.noise_on ? out[ccol(j)] : 1.0f;  // the column's normal, left there by phi_noise_kernel
    }

    float cur[U][CPT], nxt[U][CPT];
    auto load_rows = [&](float (&dst)[U][CPT], uint32_t q0) {
#pragma unroll
      for (int r = 0; r < U; ++r) {
        const uint32_t q = q0 + r < n ? q0 + r : n - 1;  // unconditional: past the end re-requests the last row
        const uint32_t w = __builtin_amdgcn_readfirstlane(s_nb[q] & 0x7fffffffu);
        const float* row = rpm_row(a.pi, w);
#pragma unroll
        for (int j = 0; j < CPT; ++j) dst[r][j] = row[ccol(j)];
      }
    };
    if (n > 0) load_rows(cur, 0);
    for (uint32_t q = 0; q < n; q += U) {  // n is a multiple of U (dispatch)
      load_rows(nxt, q + U);
      float pr[U][CPT], lo[U];
#pragma unroll
      for (int r = 0; r < U; ++r) {
        const bool y = __builtin_amdgcn_readfirstlane(s_nb[q + r] >> 31) != 0;
        const float e = y ? EPS : 1.0f - EPS;
        lo[r] = 1.0f;
#pragma unroll
        for (int j = 0; j < CPT; ++j) {  // phi.cc:241-253
          const float tt0 = cur[r][j] * bf[j];
          const float tt = y ? tt0 + e : e - tt0;
          pr[r][j] = pi_a[j] * tt;
          if (has(j)) s_vals[(size_t)r * K + col(j)] = pr[r][j];
          lo[r] = fminf(lo[r], has(j) ? fabsf(pr[r][j]) : 1.0f);
        }
      }
      __syncthreads();
      float sums[U];
      if constexpr (U == 1) {
        const float* const vv[1] = {s_vals};
        vgroup_sum<1>(vv, K, L, lgL, s_aux, s_res, phase, sums);  // phi.cc:254-257
      } else if constexpr (U == 2) {
        const float* const vv[2] = {s_vals, s_vals + K};
        vgroup_sum<2>(vv, K, L, lgL, s_aux, s_res, phase, sums);
      } else {
        const float* const vv[4] = {s_vals, s_vals + K, s_vals + 2 * (size_t)K, s_vals + 3 * (size_t)K};
        vgroup_sum<4>(vv, K, L, lgL, s_aux, s_res, phase, sums);
      }
      // phi.cc:259-263, rows in order: grads += (probs / probs_sum) / (pi * phi_sum) - 1 / phi_sum
#pragma unroll
      for (int r = 0; r < U; ++r) {
        const float probs_sum = sums[r];
        if (node_safe && lo[r] >= kProbsLo && in_range(probs_sum, kPsumLo, kPsumHi)) {
          const float rps = exact_rcp(probs_sum);
#pragma unroll
          for (int j = 0; j < CPT; ++j) {
            float qv = div_exact3(pr[r][j], probs_sum, rps);
            qv = div_exact3(qv, pi_a[j] * phi_sum, rden[j]);
            grads[j] += qv - inv_phi_sum;
          }
        } else {
#pragma unroll
          for (int j = 0; j < CPT; ++j) {
            float qv = pr[r][j] / probs_sum;
            qv = qv / (pi_a[j] * phi_sum);
            grads[j] += qv - inv_phi_sum;
          }
        }
      }
#pragma unroll
      for (int r = 0; r < U; ++r)
#pragma unroll
        for (int j = 0; j < CPT; ++j) cur[r][j] = nxt[r][j];
    }

    // SGLD step, phi.cc:265-274
    const float half = st.eps_t / 2;
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      if (has(j)) {
        const float phi_k = pi_a[j] * phi_sum;
        const float ng = a.Nn * grads[j];
        float in = a.alpha - phi_k;
        in = in + ng;
        const float drift = half * in;
        const float aa = phi_k + drift;
        const float ep = st.eps_t * phi_k;
        const float sq = sqrtf(ep);
        const float bb = sq * noise[j];
        const float v = fabsf(aa + bb);
        out[col(j)] = v > 1e-24f ? v : 1e-24f;
      }
    }
  }
}

// update_pi for the same shapes (phi.cc:177-197): one block per node, elementwise over all threads, WG_SUM emulated
template <int CPT>
__global__ __launch_bounds__(512) void update_pi_gen_kernel(ammsb_rpm pi, float* phi_sum, const float* phi_vec,
                                                             const uint32_t* nodes, uint32_t n_nodes, uint32_t K,
                                                             uint32_t L, uint32_t lgL, const ammsb_step_desc* desc,
                                                             unsigned long long* stamps) {
  extern __shared__ __align__(16) char smem[];  // [K] values, [L] lane partials, [2] sums
  float* s_vals = reinterpret_cast<float*>(smem);
  float* s_aux = s_vals + K;
  float* s_res = s_aux + L;
  if (desc) n_nodes = desc->n_nodes;
  note_stamp(stamps, desc, AMMSB_STAMP_PI);
  const uint64_t i = blockIdx.x;
  if (i >= n_nodes) return;  // block-uniform
  const uint32_t T = blockDim.x, t = threadIdx.x;
  const float* src = phi_vec + i * K;
  float v[CPT];
#pragma unroll
  for (int j = 0; j < CPT; ++j) {
    const uint32_t k = t + (uint32_t)j * T;
    v[j] = k < K ? src[k] : 0.0f;
    if (k < K) s_vals[k] = v[j];
  }
  __syncthreads();
  int phase = 0;
  const float* const vv[1] = {s_vals};
  float sums[1];
  vgroup_sum<1>(vv, K, L, lgL, s_aux, s_res, phase, sums);
  const float sum = sums[0];
  const uint32_t node = nodes[i];
  float* row = rpm_row(pi, node);
#pragma unroll
  for (int j = 0; j < CPT; ++j) {
    const uint32_t k = t + (uint32_t)j * T;
    if (k < K) row[k] = v[j] / sum;
  }
  if (t == 0) phi_sum[node] = sum;
}

constexpr uint32_t kGenMaxK = 8192;  // 512 threads x 16 columns

// columns per thread of the generic kernels: 8 up to K = 4096 (more threads per node, no register pressure), 16 above
inline int gen_cpt(uint64_t K) { return K <= 4096 ? 8 : 16; }

// threads per block of a generic kernel running `sums` concurrent WG_SUM chains: enough for the columns, and at least
// sums * L so that every virtual lane has a thread; 0 if the shape does not fit a 512-thread block
inline uint32_t gen_threads(uint64_t K, uint32_t L, uint32_t sums) {
  if (K > kGenMaxK) return 0;
  const uint32_t per_wave = 64u * (uint32_t)gen_cpt(K);
  uint32_t T = 64u * (uint32_t)((K + per_wave - 1) / per_wave);
  if (T < sums * L) T = sums * L;
  if (T < 64) T = 64;
  return T <= 512 ? T : 0;
}

template <int CPT, int U>
int launch_phi_gen_u(ammsb_ctx* ctx, const PhiArgs& a, uint32_t wg, uint32_t T, uint32_t n_groups, hipStream_t s) {
  const size_t lds = sizeof(float) * ((size_t)U * a.K + U * wg + 2 * U) + sizeof(uint32_t) * a.n;
  static const bool big = [] {  // (more than the default 64 KiB of dynamic LDS at K = 8192 with several rows per phase)
    return hipFuncSetAttribute(reinterpret_cast<const void*>(update_phi_gen_kernel<CPT, U>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048) == hipSuccess;
  }();
  (void)big;
  static const std::string name = ammsb_kname("update_phi_gen_kernel<%d, %d>", CPT, U);
  ctx->kernel_name[AMMSB_KN_PHI] = name.c_str();
  if (a.noise_on) {  // the launch's normals first, one thread per stream (see phi_noise_kernel)
    const uint64_t streams = (uint64_t)n_groups * wg;
    phi_noise_kernel<<<(unsigned)((streams + 255) / 256), 256, 0, s>>>(a, wg, ilog2_u32(wg));
    AMMSB_LAUNCH_CHECK(ctx);
  }
  update_phi_gen_kernel<CPT, U><<<n_groups, T, lds, s>>>(a, wg, ilog2_u32(wg));
  AMMSB_LAUNCH_CHECK(ctx);
  return AMMSB_OK;
}

int launch_phi_gen(ammsb_ctx* ctx, const PhiArgs& a, uint32_t wg, uint32_t n_groups, hipStream_t s) {
  const uint32_t T = gen_threads(a.K, wg, 1);
  if (!T) return AMMSB_ERANGE;
  // rows per barrier phase: as many WG_SUM chains side by side as the block has threads for (and n divides into)
  const uint32_t u = (a.n % 4 == 0 && 4 * wg <= T) ? 4 : (a.n % 2 == 0 && 2 * wg <= T) ? 2 : 1;
  if (gen_cpt(a.K) == 8) {
    if (u == 4) return launch_phi_gen_u<8, 4>(ctx, a, wg, T, n_groups, s);
    if (u == 2) return launch_phi_gen_u<8, 2>(ctx, a, wg, T, n_groups, s);
    return launch_phi_gen_u<8, 1>(ctx, a, wg, T, n_groups, s);
  }
  return launch_phi_gen_u<16, 1>(ctx, a, wg, T, n_groups, s);  // (16 columns per thread leave no registers for a second row)
}

// smallest instantiated KPT >= ceil(K / L), or 0 if K is too long for this work-group size
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

// In-process A/B of the K = 1024 kernel forms (tools/phi_forms_ab.py): -1 = follow the environment switches.  A form
// comparison across PROCESSES measures where pi landed (DESIGN_HISTORY.md R4.7), not the kernels.
static int g_phi_form_lds3 = -1, g_phi_form_nb = -1, g_phi_form_ring = -1;
extern "C" int ammsb_debug_phi_forms(int lds3, int nb, int ring) {
  g_phi_form_lds3 = lds3;
  g_phi_form_nb = nb;
  g_phi_form_ring = ring;
  return AMMSB_OK;
}

static int update_phi_common(ammsb_ctx* ctx, const float* beta, const ammsb_rpm* pi, const float* phi_sum,
                             const ammsb_set* training_set, const uint32_t* nodes, const uint32_t* neighbors,
                             uint32_t n_nodes, uint32_t step_count, ammsb_seed* seeds, uint32_t wg, uint32_t flags,
                             uint32_t group_begin, uint32_t group_end, float* phi_vec, const ammsb_step_desc* desc,
                             unsigned long long* stamps, void* stream) {
  AMMSB_CHECK_ARG(ctx, ctx && beta && pi && phi_sum && training_set && nodes && neighbors && seeds && phi_vec,
                  "null argument");
  AMMSB_CHECK_ARG(ctx, n_nodes > 0, "mini-batch nodes size = 0");  // phi.cc:732
  AMMSB_CHECK_ARG(ctx, pi->num_blocks >= 1 && pi->num_blocks <= AMMSB_RPM_MAX_BLOCKS && pi->rows_in_block > 0,
                  "bad pi descriptor");
  AMMSB_CHECK_ARG(ctx, pi->num_cols == ctx->params.K && pi->num_rows == ctx->params.N, "pi shape != (N, K)");
  AMMSB_CHECK_ARG(ctx, training_set->slots && training_set->num_bins > 0 && training_set->prime_idx < 4,
                  "bad set descriptor");
  AMMSB_CHECK_ARG(ctx, ctx->params.N < (1ull << 31), "N must be < 2^31");
  AMMSB_CHECK_ARG(ctx, is_pow2(wg) && wg >= 16 && wg <= 1024, "phi wg must be a power of two in [16, 1024]");
  const ammsb_params& p = ctx->params;
  const int kpt = pick_kpt(p.K, wg);
  // AMMSB_PHI_FORM=g: the generic kernel wherever it fits (tests compare it with the specialised ones)
  static const bool force_gen = [] {
    const char* f = getenv("AMMSB_PHI_FORM");
    return f && f[0] == 'g';
  }();
  const bool generic = kpt == 0 || (force_gen && gen_threads(p.K, wg, 1) != 0);
  if (generic && gen_threads(p.K, wg, 1) == 0) {
    snprintf(ctx->err, sizeof ctx->err, "ammsb_update_phi: K=%llu at wg=%u: more than 32 columns per work-item needs K <= %u",
             (unsigned long long)p.K, wg, kGenMaxK);
    return AMMSB_ERANGE;
  }
  PhiArgs a;
  a.beta = beta;
  a.pi = *pi;
  a.phi_sum = phi_sum;
  a.set = dev_set(*training_set);
  a.nodes = nodes;
  a.neighbors = neighbors;
  a.seeds = seeds;
  a.phi_vec = phi_vec;
  a.n_nodes = n_nodes;
  a.G = n_nodes < AMMSB_MAX_GROUPS ? n_nodes : AMMSB_MAX_GROUPS;  // phi.cc:745-747
  a.group_begin = group_begin;
  a.group_end = group_end < a.G ? group_end : a.G;
  a.K = (uint32_t)p.K;
  a.n = p.num_node_sample;
  a.eps_t = desc ? 0.0f : ammsb_eps_t(&p, step_count);
  a.desc = desc;
  a.stamps = stamps;
  a.alpha = p.alpha;
  a.epsilon = p.epsilon;
  a.Nn = (1.0f * (float)p.N) / (float)p.num_node_sample;  // phi.cc:265
  a.noise_on = (flags & AMMSB_NOISE_OFF) ? 0u : 1u;
  {
    static const int nt_mode = [] {  // AMMSB_PHI_NT=0|1 forces the hint off / on (A/B runs)
      const char* f = getenv("AMMSB_PHI_NT");
      return f ? atoi(f) : -1;
    }();
    // Non-temporal row requests pay when pi is far larger than the last-level cache (C3, 4 GB: -4 % per launch,
    // same-box A/B in round 1) and cost when it fits (C2, 100 MB in the 256 MB Infinity Cache, every row read 2.7
    // times per launch: 72.4 -> 66.3 us without the hint, profiles/r03_nt_ab.log).
    const uint64_t pi_bytes = pi->num_rows * pi->num_cols * sizeof(float);
    a.rows_nt = nt_mode >= 0 ? (uint32_t)nt_mode : (pi_bytes > (256ull << 20) ? 1u : 0u);
  }
  if (a.group_begin >= a.group_end) return AMMSB_OK;
  const uint32_t n_groups = a.group_end - a.group_begin;
  hipStream_t s = as_stream(stream);
  // AMMSB_PHI_LDS3=0|1: the three-slot K = 1024 kernel (update_phi_lds3_kernel) off / on (A/B runs; default below)
  static const bool lds3_env = [] {
    const char* f = getenv("AMMSB_PHI_LDS3");
    return f ? atoi(f) != 0 : AMMSB_PHI_LDS3_DEFAULT;
  }();
  const bool lds3 = g_phi_form_lds3 >= 0 ? g_phi_form_lds3 != 0 : lds3_env;  // (ammsb_debug_phi_forms: in-process A/B)
  static const bool force_reg = [] {
    const char* f = getenv("AMMSB_PHI_FORM");
    return f && f[0] == 'r';
  }();
  // Small launches (link mini-batches): one node per block of 8 row waves + a noise wave (update_phi_wide_kernel).
  // AMMSB_PHI_WIDE=0 turns it off; the value is the largest launch (in nodes) that takes it.
  static const uint32_t wide_max = [] {
    const char* f = getenv("AMMSB_PHI_WIDE");
    return f ? (uint32_t)atoi(f) : 512u;
  }();
  if (n_groups <= wide_max && !(flags & AMMSB_PHI_STREAMING) && !force_reg && !force_gen && (wg == 32 || wg == 64) &&
      (p.K == 256 || p.K == 512 || p.K == 1024) && a.n >= 1 && n_nodes <= AMMSB_MAX_GROUPS &&
      sizeof(float) * p.K * ((size_t)a.n + 2) + sizeof(uint32_t) * a.n <= 150 * 1024) {
    // row waves per node: 15 (+ the noise wave = 1024 threads) while 128 registers hold a lane's state, 11 at K = 1024
    static const int rw_alt = [] {  // AMMSB_PHI_WIDE_RW=8: the eight-row-wave blocks (A/B runs)
      const char* f = getenv("AMMSB_PHI_WIDE_RW");
      return f ? atoi(f) : 0;
    }();
    if (wg == 64) {
      if (p.K == 256) return rw_alt == 8 ? launch_phi_wide<4, 8, 64>(ctx, a, n_groups, s) : launch_phi_wide<4, 15, 64>(ctx, a, n_groups, s);
      if (p.K == 512) return rw_alt == 8 ? launch_phi_wide<8, 8, 64>(ctx, a, n_groups, s) : launch_phi_wide<8, 11, 64>(ctx, a, n_groups, s);
      return rw_alt == 8 ? launch_phi_wide<16, 8, 64>(ctx, a, n_groups, s) : launch_phi_wide<16, 11, 64>(ctx, a, n_groups, s);
    }
    if (p.K == 256) return rw_alt == 8 ? launch_phi_wide<4, 8, 32>(ctx, a, n_groups, s) : launch_phi_wide<4, 15, 32>(ctx, a, n_groups, s);
    if (p.K == 512) return rw_alt == 8 ? launch_phi_wide<8, 8, 32>(ctx, a, n_groups, s) : launch_phi_wide<8, 11, 32>(ctx, a, n_groups, s);
    return rw_alt == 8 ? launch_phi_wide<16, 8, 32>(ctx, a, n_groups, s) : launch_phi_wide<16, 11, 32>(ctx, a, n_groups, s);
  }
  // K = 256, large launches: persistent blocks streaming across nodes (update_phi_stream_kernel), opt-in with
  // AMMSB_PHI_STREAM=1: bit-identical but slower than the one-block-per-node kernels (see the kernel's comment).
  static const bool stream_form = [] {
    const char* f = getenv("AMMSB_PHI_STREAM");
    return f && atoi(f) != 0;
  }();
  if (stream_form && !force_reg && !force_gen && (wg == 32 || wg == 64) && p.K == 256 && pi->num_cols % 4 == 0 &&
      a.n >= 16 && a.n <= 64 && a.n % 4 == 0 && n_groups > 1024) {
    return wg == 64 ? launch_phi_stream<4, 8, 4, 64>(ctx, a, n_groups, s) : launch_phi_stream<4, 8, 4, 32>(ctx, a, n_groups, s);
  }
  // Short rows, two nodes per wave (update_phi_pair_kernel): K = 256 / 512 at wg 32 or 64, n a multiple of 2.
  // Opt-in (AMMSB_PHI_PAIR=1; 2 / 3 pick other ring depths / rows per step): bit-identical, but SLOWER at C2 in same-box
  // A/B runs -- 110 / 89 us (wg 64 / 32) against 70 / 79 us for the one-node-per-wave kernels, link steps 50 against
  // 35 us.  A lone pair wave needs ~1.8 x the time of a lone one-node wave: what a wave's time is made of at K = 256 is
  // the per-lane column work and its LDS round trips (twice as long with 8 columns per lane), not the tree / reciprocal
  // the pairing shares (profiles/r03_c2_pair_ab.log).
  static const int pair_form = [] {
    const char* f = getenv("AMMSB_PHI_PAIR");
    return f ? atoi(f) : 0;
  }();
  if (pair_form > 0 && !force_reg && !force_gen && (wg == 32 || wg == 64) && (p.K == 256 || p.K == 512) &&
      pi->num_cols % 4 == 0 && a.n >= 2 && a.n % 2 == 0 && a.n * sizeof(uint32_t) <= 4096) {
    if (p.K == 256) {
      if (pair_form == 3 && a.n % 4 == 0)
        return wg == 32 ? launch_phi_pair<8, 8, 4, 32>(ctx, a, n_groups, s) : launch_phi_pair<8, 8, 4, 64>(ctx, a, n_groups, s);
      if (pair_form == 2)
        return wg == 32 ? launch_phi_pair<8, 8, 2, 32>(ctx, a, n_groups, s) : launch_phi_pair<8, 8, 2, 64>(ctx, a, n_groups, s);
      return wg == 32 ? launch_phi_pair<8, 4, 2, 32>(ctx, a, n_groups, s) : launch_phi_pair<8, 4, 2, 64>(ctx, a, n_groups, s);
    }
    return wg == 32 ? launch_phi_pair<16, 4, 2, 32>(ctx, a, n_groups, s) : launch_phi_pair<16, 4, 2, 64>(ctx, a, n_groups, s);
  }
  // The reference's default work-group size (32, main.cc:61) on rows of 256 .. 2048 columns: 32 virtual lanes in the
  // one-wave-per-node LDS-streamed kernels (VLane<32>); K / 32 columns per work-item = 2 x the columns per physical lane
  if (wg == 32 && !force_reg && !force_gen && pi->num_cols % 4 == 0 && a.n * sizeof(uint32_t) <= 8192) {
    switch (p.K) {
      case 256:
        if (a.n & 1) return launch_phi_lds<4, 1, 8, 1, 32>(ctx, a, n_groups, s);
        if (a.n & 3) return launch_phi_lds2<4, 8, 2, 32>(ctx, a, n_groups, s);
        return launch_phi_lds2<4, 8, 4, 32>(ctx, a, n_groups, s);
      case 512:
        if (a.n & 1) return launch_phi_lds<8, 1, 4, 1, 32>(ctx, a, n_groups, s);
        return launch_phi_lds2<8, 4, 2, 32>(ctx, a, n_groups, s);
      // (tried: two rows per step at K = 1024 -- update_phi_lds2_kernel<16, 4, 2, VL>, which at wg 32 also halves the
      // swaps of the chain -- 2.19 against 1.88 ms at wg 32, 2.05 against 1.73 ms at wg 64 on C3: its 20 KiB of LDS
      // leave 8 waves per CU where the one-row kernel has 11.)
      case 1024:
        if (lds3 && a.n >= 32 * 2 / 1 / 2 + 2 + 16) return launch_phi_lds3<16, 32>(ctx, a, n_groups, s);  // n >= KV + 2, KV = 32
        return launch_phi_lds<16, 1, 2, 1, 32>(ctx, a, n_groups, s);
      case 2048: return launch_phi_lds<32, 1, 2, 1, 32>(ctx, a, n_groups, s);
    }
  }
  if (generic) return launch_phi_gen(ctx, a, wg, n_groups, s);
  // LDS-streamed kernels: K == wg * kpt exactly.  One wave per node up to K = 2048 (wg 64); for longer rows the
  // node is spread over wg / 64 waves with 16 columns per lane (K = 4096: wg 256, K = 8192: wg 512, ...).
  if (!force_reg && p.K == (uint64_t)wg * kpt && pi->num_cols % 4 == 0 && a.n * sizeof(uint32_t) <= 8192) {
    static const int ring_env = [] {  // AMMSB_PHI_RING=2|4|8: ring depth of the short-row kernels (A/B runs)
      const char* f = getenv("AMMSB_PHI_RING");
      return f ? atoi(f) : 0;
    }();
    const int ring = g_phi_form_ring >= 0 ? g_phi_form_ring : ring_env;
    // (tried: rings deep enough to hold all of a node's rows -- <4, 32, 4>, <16, 1, 8> -- for launches of a few dozen
    // nodes, i.e. link mini-batches, where the chip is empty: the kernels got slower, 18.8 -> 22.0 us at K = 256 and
    // 50.4 -> 54.4 us at K = 1024 (kernel trace).  A lone wave is not waiting for rows: it is the dependent-instruction
    // latency of one wave working through a node -- the reference's one work-group per node, whose lane partials and
    // RNG streams fix the column-to-lane map, cannot be spread over more waves bit-identically.  Not kept.)
    if (wg == 64) {
      switch (kpt) {
        case 4:
          if (ring == 2) return launch_phi_lds<4, 1, 2>(ctx, a, n_groups, s);
          if (ring == 4) return launch_phi_lds<4, 1, 4>(ctx, a, n_groups, s);
          if (ring == 8 || (a.n & 1)) return launch_phi_lds<4, 1, 8>(ctx, a, n_groups, s);
          if (ring == 42) return launch_phi_lds2<4, 4, 2>(ctx, a, n_groups, s);  // A/B: two rows, four slots
          // (tried: <4, 16, 4>, twelve rows in flight per wave and nine waves per CU instead of sixteen: 88 us against 69
          // at C2.  Fewer waves cost more than deeper prefetch gains, as at K = 1024 below.)
          if (ring == 22 || (a.n & 3)) return launch_phi_lds2<4, 8, 2>(ctx, a, n_groups, s);  // two rows per iteration
          return launch_phi_lds2<4, 8, 4>(ctx, a, n_groups, s);                               // four (n % 4 == 0)
        case 8:
          if (ring == 2) return launch_phi_lds<8, 1, 2>(ctx, a, n_groups, s);
          if (ring == 4 || (a.n & 1)) return launch_phi_lds<8, 1, 4>(ctx, a, n_groups, s);
          return launch_phi_lds2<8, 4, 2>(ctx, a, n_groups, s);
        case 16: {
          static const int nb_env = [] {  // AMMSB_PHI_NB=1|2: nodes per block of the K = 1024 kernel (A/B runs)
            const char* f = getenv("AMMSB_PHI_NB");
            return f ? atoi(f) : 1;
          }();
          const int nb = g_phi_form_nb >= 0 ? g_phi_form_nb : nb_env;
          if (nb == 2) return launch_phi_lds<16, 1, 2, 2>(ctx, a, n_groups, s);
          // A/B: three rows in flight, 8 waves per CU -- C3 update_phi 1.76 -> 2.04 ms (profiles/r03_c3_ring_ab.log)
          if (ring == 4) return launch_phi_lds<16, 1, 4>(ctx, a, n_groups, s);
          if (lds3 && ring == 0 && a.n >= 16 + 2) return launch_phi_lds3<16, 64>(ctx, a, n_groups, s);  // n >= KV + 2
          return launch_phi_lds<16, 1>(ctx, a, n_groups, s);
        }
        case 32: return launch_phi_lds<32, 1>(ctx, a, n_groups, s);
      }
    } else if (kpt == 8 && wg == 128) {
      return launch_phi_lds<8, 2>(ctx, a, n_groups, s);
    } else if (kpt == 16) {
      switch (wg) {
        case 128: return launch_phi_lds<16, 2>(ctx, a, n_groups, s);
        case 256: return launch_phi_lds<16, 4>(ctx, a, n_groups, s);
        case 512: return launch_phi_lds<16, 8>(ctx, a, n_groups, s);
      }
    }
  }
  AMMSB_DISPATCH_HOT_L(wg, AMMSB_DISPATCH_KPT(kpt, return (launch_phi<L_, KPT_>(ctx, a, n_groups, s))));
  return AMMSB_OK;
}

extern "C" int ammsb_update_phi(ammsb_ctx* ctx, const float* beta, const ammsb_rpm* pi, const float* phi_sum,
                                const ammsb_set* training_set, const uint32_t* nodes, const uint32_t* neighbors,
                                uint32_t n_nodes, uint32_t step_count, ammsb_seed* seeds, uint32_t wg, uint32_t flags,
                                uint32_t group_begin, uint32_t group_end, float* phi_vec, void* stream) {
  return update_phi_common(ctx, beta, pi, phi_sum, training_set, nodes, neighbors, n_nodes, step_count, seeds, wg, flags,
                           group_begin, group_end, phi_vec, nullptr, nullptr, stream);
}

int ammsb_update_phi_d(ammsb_ctx* ctx, const float* beta, const ammsb_rpm* pi, const float* phi_sum,
                       const ammsb_set* training_set, const uint32_t* nodes, const uint32_t* neighbors,
                       uint32_t n_nodes_cap, ammsb_seed* seeds, uint32_t wg, uint32_t flags, float* phi_vec,
                       const ammsb_step_desc* desc, unsigned long long* stamps, void* stream) {
  AMMSB_CHECK_ARG(ctx, ctx && desc, "null descriptor");
  return update_phi_common(ctx, beta, pi, phi_sum, training_set, nodes, neighbors, n_nodes_cap, 0, seeds, wg, flags, 0,
                           0xFFFFFFFFu, phi_vec, desc, stamps, stream);
}

// Diagnostic: resident blocks per CU of the LDS-streamed update_phi kernel picked for (K, wg), as the runtime's
// occupancy calculator sees it (registers, static + dynamic LDS).  0 if (K, wg) takes the register kernel.
template <int KPT, int W, int D>
static int occ_of(uint32_t n, int* out) {
  const size_t lds = (size_t)W * (D + 1) * sizeof(float) * 64 * KPT + sizeof(uint32_t) * n;
  return hipOccupancyMaxActiveBlocksPerMultiprocessor(out, update_phi_lds_kernel<KPT, W, D>, 64 * W, lds) == hipSuccess
             ? AMMSB_OK
             : AMMSB_EHIP;
}

extern "C" int ammsb_update_phi_occupancy(ammsb_ctx* ctx, uint32_t wg, int* blocks_per_cu, int* waves_per_block) {
  AMMSB_CHECK_ARG(ctx, ctx && blocks_per_cu && waves_per_block, "null argument");
  const ammsb_params& p = ctx->params;
  const int kpt = pick_kpt(p.K, wg);
  const uint32_t n = p.num_node_sample;
  *blocks_per_cu = 0;
  *waves_per_block = (int)(wg / 64);
  if (kpt == 0 || p.K != (uint64_t)wg * kpt) return AMMSB_OK;
  if (wg == 64 && kpt == 4) return occ_of<4, 1, 8>(n, blocks_per_cu);
  if (wg == 64 && kpt == 8) return occ_of<8, 1, 4>(n, blocks_per_cu);
  if (wg == 64 && kpt == 16) return occ_of<16, 1, 2>(n, blocks_per_cu);
  if (wg == 64 && kpt == 32) return occ_of<32, 1, 2>(n, blocks_per_cu);
  if (wg == 128 && kpt == 8) return occ_of<8, 2, 2>(n, blocks_per_cu);
  if (wg == 128 && kpt == 16) return occ_of<16, 2, 2>(n, blocks_per_cu);
  if (wg == 256 && kpt == 16) return occ_of<16, 4, 2>(n, blocks_per_cu);
  if (wg == 512 && kpt == 16) return occ_of<16, 8, 2>(n, blocks_per_cu);
  return AMMSB_OK;
}

static int update_pi_common(ammsb_ctx* ctx, const ammsb_rpm* pi, float* phi_sum, const float* phi_vec,
                            const uint32_t* nodes, uint32_t n_nodes, uint32_t wg, const ammsb_step_desc* desc,
                            unsigned long long* stamps, void* stream) {
  AMMSB_CHECK_ARG(ctx, ctx && pi && phi_sum && phi_vec && nodes, "null argument");
  AMMSB_CHECK_ARG(ctx, pi->num_blocks >= 1 && pi->num_blocks <= AMMSB_RPM_MAX_BLOCKS && pi->rows_in_block > 0,
                  "bad pi descriptor");
  AMMSB_CHECK_ARG(ctx, pi->num_cols == ctx->params.K, "pi cols != K");
  AMMSB_CHECK_ARG(ctx, is_pow2(wg) && wg >= 16 && wg <= 1024, "phi wg must be a power of two in [16, 1024]");
  if (n_nodes == 0) return AMMSB_OK;
  const int kpt = pick_kpt(ctx->params.K, wg);
  const uint32_t K = (uint32_t)ctx->params.K;
  hipStream_t s = as_stream(stream);
  static const bool force_gen = [] {
    const char* f = getenv("AMMSB_PHI_FORM");
    return f && f[0] == 'g';
  }();
  if (kpt == 0 || (force_gen && gen_threads(K, wg, 1) != 0)) {
    const uint32_t T = gen_threads(K, wg, 1);
    if (!T) {
      snprintf(ctx->err, sizeof ctx->err, "ammsb_update_pi: K=%u at wg=%u: more than 32 columns per work-item needs K <= %u",
               K, wg, kGenMaxK);
      return AMMSB_ERANGE;
    }
    const size_t lds = sizeof(float) * ((size_t)K + wg + 2);
    ctx->kernel_name[AMMSB_KN_PI] = gen_cpt(K) == 8 ? "update_pi_gen_kernel<8>" : "update_pi_gen_kernel<16>";
    if (gen_cpt(K) == 8)
      update_pi_gen_kernel<8><<<n_nodes, T, lds, s>>>(*pi, phi_sum, phi_vec, nodes, n_nodes, K, wg, ilog2_u32(wg), desc, stamps);
    else
      update_pi_gen_kernel<16><<<n_nodes, T, lds, s>>>(*pi, phi_sum, phi_vec, nodes, n_nodes, K, wg, ilog2_u32(wg), desc, stamps);
    AMMSB_LAUNCH_CHECK(ctx);
    return AMMSB_OK;
  }
  AMMSB_DISPATCH_HOT_L(wg, AMMSB_DISPATCH_KPT(kpt, return (launch_pi<L_, KPT_>(ctx, *pi, phi_sum, phi_vec, nodes,
                                                                                n_nodes, K, desc, stamps, s))));
  return AMMSB_OK;
}

extern "C" int ammsb_update_pi(ammsb_ctx* ctx, const ammsb_rpm* pi, float* phi_sum, const float* phi_vec,
                               const uint32_t* nodes, uint32_t n_nodes, uint32_t wg, void* stream) {
  return update_pi_common(ctx, pi, phi_sum, phi_vec, nodes, n_nodes, wg, nullptr, nullptr, stream);
}

int ammsb_update_pi_d(ammsb_ctx* ctx, const ammsb_rpm* pi, float* phi_sum, const float* phi_vec, const uint32_t* nodes,
                      uint32_t n_nodes_cap, uint32_t wg, const ammsb_step_desc* desc, unsigned long long* stamps,
                      void* stream) {
  AMMSB_CHECK_ARG(ctx, ctx && desc, "null descriptor");
  return update_pi_common(ctx, pi, phi_sum, phi_vec, nodes, n_nodes_cap, wg, desc, stamps, stream);
}
