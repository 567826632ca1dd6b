// Device-side building blocks shared by the gfx950 kernels: RNG streams, cuckoo lookup,
// row-partitioned-matrix addressing and the WG_SUM-ordered group reduction.
//
// Arithmetic contract (matches the CPU oracle): IEEE binary32, no FMA contraction (the TU is built
// with -ffp-contract=off), correctly rounded divide/sqrt (hipcc default), operations in the order
// the reference kernel text writes them; exp/log evaluated in binary64 and rounded once.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ammsb.h"

namespace ammsb {

// --------------------------------------------------------------------------------------- RNG
// xorshift128+ / uniform / randint / ziggurat normal / Marsaglia-Tsang gamma with the exact state
// update and draw order of mcmc/random.cl.inc:13-49, :229-273, :353-395.

#include "zig_tables.inc"  // zig_ytab / zig_wtab / zig_ktab (tools/gen_ziggurat_tables.py)

// LDS copy of the ziggurat tables, 1.5 KiB per block.  Only wtab and ktab are read on the fast path; ytab is read
// by about one draw in 80.  Leaving ytab in global memory (-DAMMSB_ZIG_YTAB_GLOBAL: 1 KiB of LDS) lets twelve
// instead of eleven one-wave blocks of the K = 1024 phi kernel share a CU (ammsb_update_phi_occupancy reports 12 vs
// 11), but every slow-path draw then stalls its whole wave on a global load: measured 3-9 % SLOWER per launch in
// three alternating same-box pairs (1.73 / 1.77 / 1.77 ms against 1.56 / 1.72 / 1.64 ms), so the tables stay in LDS.
struct ZigTables {
#ifndef AMMSB_ZIG_YTAB_GLOBAL
  float ytab[128];
#endif
  float wtab[128];
  uint32_t ktab[128];
};

__device__ __forceinline__ void zig_load(ZigTables* t) {
  for (int i = threadIdx.x; i < 128; i += blockDim.x) {
#ifndef AMMSB_ZIG_YTAB_GLOBAL
    t->ytab[i] = zig_ytab[i];
#endif
    t->wtab[i] = zig_wtab[i];
    t->ktab[i] = zig_ktab[i];
  }
}

__device__ __forceinline__ uint64_t rng_next(ammsb_seed& s) {  // random.cl.inc:13-25
  uint64_t s1 = s.x;
  const uint64_t s0 = s.y;
  s.x = s0;
  s1 ^= s1 << 23;
  s.y = s1 ^ s0 ^ (s1 >> 17) ^ (s0 >> 26);
  return s.y + s0;
}

__device__ __forceinline__ float rng_uniform(ammsb_seed& s) {  // random.cl.inc:34-35
  const float r = 1.0f * (float)rng_next(s);
  return r / 18446744073709551616.0f;  // (float)ULONG_MAX == 2^64
}

__device__ __forceinline__ float expf_cr(float x) { return (float)exp((double)x); }
__device__ __forceinline__ float logf_cr(float x) { return (float)log((double)x); }
__device__ __forceinline__ float powf_cr(float x, float y) { return (float)pow((double)x, (double)y); }

// gsl_ran_gaussian_ziggurat(sigma = 1), random.cl.inc:229-273
__device__ __forceinline__ float rng_normal(ammsb_seed& s, const ZigTables* t) {
  const float PARAM_R = 3.44428647676f;
  uint32_t i, j;
  int sign;
  float x, y;
  for (;;) {
    const uint64_t k = rng_next(s);
    i = (uint32_t)(k & 0xFF);
    j = (uint32_t)((k >> 8) & 0xFFFFFF);
    sign = (i & 0x80) ? +1 : -1;
    i &= 0x7f;
    x = (float)j * t->wtab[i];
    if (j < t->ktab[i]) break;
    if (i < 127) {
#ifndef AMMSB_ZIG_YTAB_GLOBAL
      const float y0 = t->ytab[i], y1 = t->ytab[i + 1];
#else
      const float y0 = zig_ytab[i], y1 = zig_ytab[i + 1];
#endif
      const float U1 = rng_uniform(s);
      const float d = y0 - y1;
      const float m = d * U1;
      y = y1 + m;
    } else {
      const float U1 = 1.0f - rng_uniform(s);
      const float U2 = rng_uniform(s);
      const float l = logf_cr(U1) / PARAM_R;
      x = PARAM_R - l;
      const float h = 0.5f * PARAM_R;
      const float tt = x - h;
      const float a = -PARAM_R * tt;
      y = expf_cr(a) * U2;
    }
    const float hx = -0.5f * x;
    const float xx = hx * x;
    // `y < expf_cr(xx)` is a yes / no question, and the correctly rounded exponential (a double-precision exp, ~70
    // instructions that 83 % of a wave's draws execute because one of 64 lanes is in a wedge) is only needed when y
    // is within rounding distance of the threshold.  v_exp_f32 is accurate to 1 ulp and the argument xx * log2(e)
    // (|.| <= 50: x <= 8.3 even in the tail) carries at most 1.5 * 2^-24 relative error, i.e. 4.5e-6 absolute, so the
    // hardware value is within 2^-18 of exp(xx), relatively; outside a band of 2^-14 around it the comparison with
    // the exact value has the same outcome.  Inside the band (about 1 wedge draw in 10^4) the exact value decides.
    const float e_fast = __builtin_amdgcn_exp2f(xx * 0x1.715476p+0f);
    const float tol = e_fast * 0x1p-14f;
    float thr = e_fast;
    if (fabsf(y - e_fast) <= tol) thr = expf_cr(xx);
    if (y < thr) break;
  }
  const float ss = (float)sign * 1.0f;
  return ss * x;
}

__device__ __forceinline__ float rng_uniform_pos(ammsb_seed& s) {  // random.cl.inc:310-317
  float x;
  do {
    x = rng_uniform(s);
  } while (x == 0);
  return x;
}

// gsl_ran_gamma, random.cl.inc:353-395 (non-recursive branch)
__device__ __forceinline__ float rng_gamma(ammsb_seed& s, const ZigTables* t, float a, float b) {
  float f = 1.0f;
  while (a < 1) {
    const float u = rng_uniform_pos(s);
    const float ia = 1.0f / a;
    f = f * powf_cr(u, ia);
    a = 1.0f + a;
  }
  float x, v, u;
  const float third = 1.0f / 3.0f;
  const float d = a - third;
  const float c = third / sqrtf(d);
  for (;;) {
    do {
      x = rng_normal(s, t);
      const float cx = c * x;
      v = 1.0f + cx;
    } while (v <= 0);
    const float v2 = v * v;
    v = v2 * v;
    u = rng_uniform_pos(s);
    float q = 0.0331f * x;
    q = q * x;
    q = q * x;
    q = q * x;
    if (u < 1.0f - q) break;
    const float hx = 0.5f * x;
    const float hxx = hx * x;
    const float omv = 1.0f - v;
    const float in = omv + logf_cr(v);
    const float din = d * in;
    if (logf_cr(u) < hxx + din) break;
  }
  float r = f * b;
  r = r * d;
  return r * v;
}

// x % d for a 64-bit draw and a 31-bit modulus without the ~200-instruction software divide: with
// m = floor((2^64 - 1) / d), q = mulhi64(x, m) is floor(x / d) or up to 2 less, so r = x - q d needs at most
// two corrections.  Exact for every x (the reference's randint is `rand % n`, random.cl.inc:37-39).
struct FastMod {
  uint64_t d, m;
};
__host__ __device__ inline FastMod fast_mod_init(uint64_t d) { return FastMod{d, ~0ull / d}; }
__device__ __forceinline__ uint64_t fast_mod(uint64_t x, const FastMod& f) {
  uint64_t r = x - __umul64hi(x, f.m) * f.d;
  while (r >= f.d) r -= f.d;
  return r;
}

// ---------------------------------------------------------------------------- exact division
// hipcc lowers an IEEE-correct binary32 `x / d` to
//     d' = div_scale(d), x' = div_scale(x); r0 = rcp(d'); e0 = fma(-d', r0, 1); r = fma(e0, r0, r0);
//     q0 = x' * r; e1 = fma(-d', q0, x'); q1 = fma(e1, r, q0); e2 = fma(-d', q1, x');
//     q  = div_fmas(e2, r, q1); result = div_fixup(q, d, x)
// (11 VALU instructions).  div_scale only rescales when an operand or the quotient is close to the
// denormal / overflow range, and div_fmas / div_fixup are the identity then, so for operands in the
// safe zone below the quotient equals the five-instruction tail q0..q with r computed once per
// divisor.  The phi kernel divides K/L numerators by the same probs_sum and the same K/L
// denominators across all n neighbours, so hoisting r removes more than half of its VALU work while
// staying bit-identical; operands outside the safe zone take the plain `/` path.
//
// When is the unscaled tail exact?  It needs the two remainders e1, e2 (about |x| * 2^-24) to be
// exactly representable, i.e. |x| >= 2^-102, a normal reciprocal, and a quotient away from the
// denormal / overflow range; then div_scale is a no-op too (its triggers: denormal d, 1/d denormal,
// exponent(x) <= 23, exponent(x) - exponent(d) >= 96, denormal quotient) and both sequences return the
// one correctly rounded quotient.  The phi kernel establishes this with range checks hoisted as far
// out as the data allows (ammsb_phi.hip); anything outside takes the plain `/` path.

constexpr float kProbsLo = 0x1p-100f;                       // smallest |numerator| of the fast path
constexpr float kPsumLo = 0x1p-30f, kPsumHi = 4.0f;         // probs_sum
constexpr float kDenLo = 0x1p-110f, kDenHi = 0x1p+40f;      // pi_a * phi_sum
constexpr float kPhiSumLo = 0x1p-20f, kPhiSumHi = 0x1p+40f; // phi_sum
constexpr float kBetaHi = 1.0f - 0x1p-20f;                  // beta_k in [EPSILON, kBetaHi] bounds tt to [2^-24, 2]

__device__ __forceinline__ bool in_range(float v, float lo, float hi) { return v >= lo && v <= hi; }

__device__ __forceinline__ float refined_rcp(float d) {
  const float r0 = __builtin_amdgcn_rcpf(d);
  const float e0 = __builtin_fmaf(-d, r0, 1.0f);
  return __builtin_fmaf(e0, r0, r0);
}

// x / d given r = refined_rcp(d); exact for operands in the safe zone
__device__ __forceinline__ float div_with_rcp(float x, float d, float r) {
  const float q0 = x * r;
  const float e1 = __builtin_fmaf(-d, q0, x);
  const float q1 = __builtin_fmaf(e1, r, q0);
  const float e2 = __builtin_fmaf(-d, q1, x);
  return __builtin_fmaf(e2, r, q1);
}

// Three-instruction form (Markstein's final step): with y = RN(1/d) -- the correctly rounded reciprocal,
// obtained here from one IEEE `1.0f / d` per divisor -- and q0 = RN(x * y),
//     rem = fma(-d, q0, x) is exact and fma(rem, y, q0) = RN(x / d)
// (P. Markstein, "Computation of elementary functions on the IBM RISC System/6000", 1990, Thm 4-5; the
// same no-underflow conditions as above).  A Newton-refined v_rcp_f32 is NOT always RN(1/d) (33 of the
// 3 * 2^23 seed cases miss, tests/cpp/div_check.c), hence the real division for y.  div_check.c also
// runs the three-instruction tail against `/` on 5e8 operand pairs without a mismatch.
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float exact_rcp(float d) { return 1.0f / d; }

__device__ __forceinline__ float div_exact3(float x, float d, float y) {
  const float q0 = x * y;
  const float rem = __builtin_fmaf(-d, q0, x);
  return __builtin_fmaf(rem, y, q0);
}

// two quotients per instruction (v_pk_mul_f32 / v_pk_fma_f32 round each half like the scalar forms)
__device__ __forceinline__ f32x2 div_exact3(f32x2 x, f32x2 d, f32x2 y) {
  const f32x2 q0 = x * y;
  const f32x2 rem = __builtin_elementwise_fma(-d, q0, x);
  return __builtin_elementwise_fma(rem, y, q0);
}

// ------------------------------------------------------------------------------------ cuckoo
// Set_HasEdge, mcmc/cuckoo.cc:27-69.  Both 32-byte bins are fetched at once (two independent
// 2 x 16 B loads each) instead of the reference's dependent second probe.

__device__ const uint64_t kSetPrimes[8] = {15485807ull, 920429591ull, 379906717ull, 740320571ull,
                                           256204747ull, 379927517ull, 13ull,        17ull};

__device__ __forceinline__ uint64_t make_edge(uint32_t a, uint32_t b) {  // learner.cc:22-27
  const uint32_t u = a < b ? a : b, v = a < b ? b : a;
  return ((uint64_t)u << 32) | v;
}

__device__ __forceinline__ bool set_has(const ammsb_set& set, uint64_t k) {
  const uint64_t h1 = (kSetPrimes[2 * set.prime_idx] * k) % set.num_bins;
  const uint64_t h2 = (k ^ kSetPrimes[2 * set.prime_idx + 1]) % set.num_bins;
  const ulonglong2* b1 = reinterpret_cast<const ulonglong2*>(set.slots + h1 * 4);
  const ulonglong2* b2 = reinterpret_cast<const ulonglong2*>(set.slots + (set.num_bins + h2) * 4);
  const ulonglong2 a0 = b1[0], a1 = b1[1], c0 = b2[0], c1 = b2[1];
  return (a0.x == k) | (a0.y == k) | (a1.x == k) | (a1.y == k) | (c0.x == k) | (c0.y == k) |
         (c1.x == k) | (c1.y == k);
}

// The kernels' own view of a set: the public descriptor plus the magic number of fast_mod() for num_bins, computed on
// the host when a launch is prepared.  hipcc lowers a 64-bit `%` by a run-time divisor to a software loop of several
// hundred cycles, and a probe needs two of them in front of its (dependent) loads; with the magic a modulo is a 64-bit
// high multiply and at most two corrections, exact for every operand (see fast_mod).
struct DevSet {
  const uint64_t* slots;
  uint64_t num_bins;
  FastMod mod;
  uint32_t prime_idx;
};
inline __host__ DevSet dev_set(const ammsb_set& s) {
  return DevSet{s.slots, s.num_bins, fast_mod_init(s.num_bins ? s.num_bins : 1), s.prime_idx};
}

// the two bins of key k, as loaded (set_probe) and their comparison with k (set_hit): split so that a caller can put
// independent work between issuing the loads and needing their result
struct SetProbe {
  ulonglong2 a0, a1, c0, c1;
};
__device__ __forceinline__ SetProbe set_probe(const DevSet& set, uint64_t k) {
  const uint64_t h1 = fast_mod(kSetPrimes[2 * set.prime_idx] * k, set.mod);
  const uint64_t h2 = fast_mod(k ^ kSetPrimes[2 * set.prime_idx + 1], set.mod);
  const ulonglong2* b1 = reinterpret_cast<const ulonglong2*>(set.slots + h1 * 4);
  const ulonglong2* b2 = reinterpret_cast<const ulonglong2*>(set.slots + (set.num_bins + h2) * 4);
  return SetProbe{b1[0], b1[1], b2[0], b2[1]};
}
__device__ __forceinline__ bool set_hit(const SetProbe& p, uint64_t k) {
  return (p.a0.x == k) | (p.a0.y == k) | (p.a1.x == k) | (p.a1.y == k) | (p.c0.x == k) | (p.c0.y == k) |
         (p.c1.x == k) | (p.c1.y == k);
}
__device__ __forceinline__ bool set_has(const DevSet& set, uint64_t k) { return set_hit(set_probe(set, k), k); }

// ------------------------------------------------------------------------- partitioned matrix
// TTRowPartitionedMatrix_Row, mcmc/partitioned-alloc.h:22-29, with 64-bit offsets.

__device__ __forceinline__ float* rpm_row(const ammsb_rpm& m, uint64_t row) {
#ifdef AMMSB_RPM_SINGLE  // experiment builds only: every kernel without the multi-block path (what it costs where ONE is not used)
  return reinterpret_cast<float*>(m.blocks[0]) + row * m.num_cols;
#endif
  if (m.num_blocks == 1) return reinterpret_cast<float*>(m.blocks[0]) + row * m.num_cols;
  // Row indices are vertex ids (32 bits) and a block holds at most that many rows: a 32-bit division.  Written with
  // 64-bit operands this line was ~120 instructions at every call site (hipcc's 64-bit divide with its own "do both fit
  // 32 bits" test in front) -- dead weight beside the single-block path that every configuration of interest takes on a
  // 288 GB device, but weight the instruction cache carried: update_phi at K = 32 ran 13 % faster without it (same-box
  // A/B of a build with the branch compiled out), at K = 256 3 %.
  const uint32_t r = (uint32_t)row, rib = (uint32_t)m.rows_in_block;
  const uint32_t blk = r / rib;
  const uint64_t off = (uint64_t)(r - blk * rib) * m.num_cols;
  return reinterpret_cast<float*>(m.blocks[blk]) + off;
}

// -------------------------------------------------------------------------------- group sum
// A "virtual group" is the reference's OpenCL work-group of L work-items (L a power of two).
//   L <= 64: 64/L groups share one 64-thread block (one wave); reductions stay in registers.
//   L  > 64: one group per block of L threads; partials cross waves through LDS.
// group_sum() returns, in every lane of the group, exactly the value WG_SUM_TT_LOCAL_
// (mcmc/algorithm/sum.cc:20-29) leaves in aux[0]: the halving tree aux[l] += aux[l+p2].  For the
// in-wave part an XOR butterfly is used: since a+b == b+a bitwise, lane l and lane l^p2 compute the
// same value at every step, so all lanes end with the tree's root.

template <int L>
struct Group {
  static_assert(L >= 1 && L <= 1024 && (L & (L - 1)) == 0, "L must be a power of two <= 1024");
  static constexpr int BLOCK = L < 64 ? 64 : L;
  static constexpr int PER_BLOCK = BLOCK / L;
  static constexpr int AUX = L > 64 ? 2 * L : 1;  // floats of LDS scratch (double buffered)

  __device__ __forceinline__ static int lane() { return threadIdx.x & (L - 1); }
  __device__ __forceinline__ static int slot() { return threadIdx.x / L; }

  // Full-wave float form without LDS traffic: v_permlane32_swap / v_permlane16_swap (gfx950) fold
  // lanes l+32 and l+16 onto l, four DPP row_shl adds fold l+8, l+4, l+2, l+1 -- at every step lane l
  // (l < p2) computes aux[l] + aux[l+p2], the reference tree -- and lane 0 is broadcast through an SGPR.
  __device__ __forceinline__ static float wave_tree64(float v) {
    typedef unsigned int u2 __attribute__((ext_vector_type(2)));
    u2 t = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(t[0]) + __uint_as_float(t[1]);
    t = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(t[0]) + __uint_as_float(t[1]);
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x108, 0xf, 0xf, true));  // row_shl:8
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x104, 0xf, 0xf, true));  // row_shl:4
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x102, 0xf, 0xf, true));  // row_shl:2
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x101, 0xf, 0xf, true));  // row_shl:1
    return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v)));
  }

  template <typename T>
  __device__ __forceinline__ static T wave_tree(T v) {
    constexpr int W = L < 64 ? L : 64;
    if constexpr (W == 64 && sizeof(T) == 4 && __is_same(T, float)) {
      return wave_tree64(v);
    } else if constexpr (W == 32 && sizeof(T) == 4 && __is_same(T, float)) {
      // two groups of 32 per wave (the reference's default work-group size): wave_tree64 without its first level on
      // both halves at once -- v_permlane16_swap folds lanes l + 16 onto l in each half, four DPP adds fold l + 8 ..
      // l + 1, lanes 0 and 32 are handed to their halves through SGPRs.  (The XOR butterfly below needs five
      // ds_bpermute round trips; the short-row kernels at K = 32 are nothing but this latency.)
      typedef unsigned int u2 __attribute__((ext_vector_type(2)));
      const u2 t = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
      float x = __uint_as_float(t[0]) + __uint_as_float(t[1]);
      x += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), 0x108, 0xf, 0xf, true));  // row_shl:8
      x += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), 0x104, 0xf, 0xf, true));  // row_shl:4
      x += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), 0x102, 0xf, 0xf, true));  // row_shl:2
      x += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), 0x101, 0xf, 0xf, true));  // row_shl:1
      const float lo = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(x), 0));
      const float hi = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(x), 32));
      return (threadIdx.x & 32) ? hi : lo;
    } else {
#pragma unroll
      for (int p2 = W >> 1; p2 > 0; p2 >>= 1) v += __shfl_xor(v, p2, 64);
      return v;
    }
  }

  // `phase` alternates 0/1 between consecutive calls so that one barrier per call suffices.
  template <typename T>
  __device__ __forceinline__ static T sum(T v, T* aux, int& phase) {
    if constexpr (L <= 64) {
      return wave_tree(v);
    } else {
      T* a = aux + phase * L;
      phase ^= 1;
      const int lid = threadIdx.x;
      a[lid] = v;
      __syncthreads();
#pragma unroll
      for (int p2 = L >> 1; p2 >= 64; p2 >>= 1) {
        if (lid < p2) a[lid] += a[lid + p2];
        __syncthreads();
      }
      return wave_tree(a[lid & 63]);
    }
  }
};

inline __host__ __device__ bool is_pow2(uint32_t v) { return v && !(v & (v - 1)); }

// ------------------------------------------------------------- 32 virtual lanes in one wave64
// The LDS-streamed kernels give a node (an edge slot) one whole wave: physical lane p owns columns p + 64 i.  The
// reference's DEFAULT work-group size is 32 (main.cc:61-64): virtual lane l owns columns l + 32 j, i.e. the columns of
// physical lanes l (even j = 2 i) and l + 32 (odd j = 2 i + 1).  The elementwise work does not care who owns a column;
// WG_SUM does: lane l's partial is the chain ((0 + x[l]) + x[l + 32]) + x[l + 64] ... in ascending j (sum.cc:20-22),
// which alternates between the two physical lanes.  v_permlane32_swap hands every lane both values of a column pair
// position (in all 64 lanes: t[0] = the lower half's, t[1] = the upper half's), so both halves run the identical
// chain -- twice the dependent adds of the 64-lane form, no LDS traffic -- and the tree over the 32 virtual lanes is
// wave_tree64 without its first level (the halves hold the same values).  VL = 64 is the plain one-column-per-add form.
template <int VL>
struct VLane {
  static_assert(VL == 32 || VL == 64, "virtual lanes per wave");
  typedef unsigned int u2 __attribute__((ext_vector_type(2)));
  // acc += the next column(s) of this lane's virtual lane, in the reference's order
  __device__ __forceinline__ static void chain(float& acc, float v) {
    if constexpr (VL == 64) {
      acc += v;
    } else {
      const u2 t = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
      acc += __uint_as_float(t[0]);
      acc += __uint_as_float(t[1]);
    }
  }
  // the halving tree over the VL lane partials (sum.cc:23-29), result in every lane
  __device__ __forceinline__ static float tree(float v) {
    if constexpr (VL == 64) {
      return Group<64>::wave_tree64(v);
    } else {
      u2 t = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
      v = __uint_as_float(t[0]) + __uint_as_float(t[1]);
      v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x108, 0xf, 0xf, true));  // row_shl:8
      v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x104, 0xf, 0xf, true));  // row_shl:4
      v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x102, 0xf, 0xf, true));  // row_shl:2
      v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x101, 0xf, 0xf, true));  // row_shl:1
      return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v)));
    }
  }
  // U = 2 or 4 rows at once, transposed: the same additions in the same association (a + b == b + a bit for bit, so
  // which lane forms a sum does not matter, only which two values it adds), but every instruction works for all U rows.
  //   VL = 32  the chain of a row alternates between the two physical lanes of a virtual lane, so the one-row form has
  //            both halves of the wave run the identical chain; here v_permlane32_swap(row r, row r + U/2) hands the
  //            lower half both values of row r and the upper half both values of row r + U/2: one swap and two adds
  //            per column pair for TWO rows (one-row form: a copy, a swap and two adds per row).
  //   tree     level 32 (VL = 64 only) folds rows r and r + U/2 with one swap + one add (result of r in the lower half,
  //            of r + U/2 in the upper); level 16 folds the two registers of U = 4 into one with v_permlane16_swap
  //            (16-lane row i then holds row i's partials); the four DPP levels run once for all rows; row i's sum is
  //            read from lane 16 i (U = 4) or 32 i (U = 2).
  // acc[] holds U accumulators (VL = 64) or U/2 (VL = 32), zero-initialised by the caller.
  template <int U>
  __device__ __forceinline__ static void chain_rows(float (&acc)[U], const float (&v)[U]) {
    static_assert(U == 2 || U == 4, "rows per step");
    if constexpr (VL == 64) {
#pragma unroll
      for (int r = 0; r < U; ++r) acc[r] += v[r];
    } else {
#pragma unroll
      for (int r = 0; r < U / 2; ++r) {
        const u2 t = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[r]), __float_as_uint(v[r + U / 2]), false, false);
        acc[r] += __uint_as_float(t[0]);
        acc[r] += __uint_as_float(t[1]);
      }
    }
  }
  template <int U>
  __device__ __forceinline__ static void tree_rows(const float (&acc)[U], float (&out)[U]) {
    static_assert(U == 2 || U == 4, "rows per step");
    float h[U / 2];  // rows r (lower half) and r + U/2 (upper half) after level 32
#pragma unroll
    for (int r = 0; r < U / 2; ++r) {
      if constexpr (VL == 64) {
        const u2 t = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[r]), __float_as_uint(acc[r + U / 2]), false, false);
        h[r] = __uint_as_float(t[0]) + __uint_as_float(t[1]);
      } else {
        h[r] = acc[r];
      }
    }
    float v;
    if constexpr (U == 4) {
      const u2 t = __builtin_amdgcn_permlane16_swap(__float_as_uint(h[0]), __float_as_uint(h[1]), false, false);
      v = __uint_as_float(t[0]) + __uint_as_float(t[1]);
    } else {
      const u2 t = __builtin_amdgcn_permlane16_swap(__float_as_uint(h[0]), __float_as_uint(h[0]), false, false);
      v = __uint_as_float(t[0]) + __uint_as_float(t[1]);
    }
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x108, 0xf, 0xf, true));  // row_shl:8
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x104, 0xf, 0xf, true));  // row_shl:4
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x102, 0xf, 0xf, true));  // row_shl:2
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x101, 0xf, 0xf, true));  // row_shl:1
#pragma unroll
    for (int r = 0; r < U; ++r)
      out[r] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), U == 4 ? 16 * r : 32 * r));
  }
  // tree_rows plus the U reciprocals RN(1 / sum_r) from ONE IEEE division: after the last DPP level the sums sit in
  // lanes 0, 16, 32, 48 of one register (U = 4; 0 and 32 for U = 2), so dividing that register once and reading the
  // same lanes back gives each row exactly the quotient 1.0f / out[r] would -- a division costs ~11 vector
  // instructions and the callers used to issue it once per row on a wave-uniform operand (44 of a four-row step's
  // ~200 instructions at K = 256).  The other lanes divide by partial sums nobody reads.
  template <int U>
  __device__ __forceinline__ static void tree_rows_rcp(const float (&acc)[U], float (&out)[U], float (&rcp)[U]) {
    static_assert(U == 2 || U == 4, "rows per step");
    float h[U / 2];
#pragma unroll
    for (int r = 0; r < U / 2; ++r) {
      if constexpr (VL == 64) {
        const u2 t = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[r]), __float_as_uint(acc[r + U / 2]), false, false);
        h[r] = __uint_as_float(t[0]) + __uint_as_float(t[1]);
      } else {
        h[r] = acc[r];
      }
    }
    float v;
    if constexpr (U == 4) {
      const u2 t = __builtin_amdgcn_permlane16_swap(__float_as_uint(h[0]), __float_as_uint(h[1]), false, false);
      v = __uint_as_float(t[0]) + __uint_as_float(t[1]);
    } else {
      const u2 t = __builtin_amdgcn_permlane16_swap(__float_as_uint(h[0]), __float_as_uint(h[0]), false, false);
      v = __uint_as_float(t[0]) + __uint_as_float(t[1]);
    }
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x108, 0xf, 0xf, true));  // row_shl:8
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x104, 0xf, 0xf, true));  // row_shl:4
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x102, 0xf, 0xf, true));  // row_shl:2
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x101, 0xf, 0xf, true));  // row_shl:1
    const float rv = 1.0f / v;  // (exact_rcp, every lane at once)
#pragma unroll
    for (int r = 0; r < U; ++r) {
      out[r] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), U == 4 ? 16 * r : 32 * r));
      rcp[r] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(rv), U == 4 ? 16 * r : 32 * r));
    }
  }
  // stream / virtual lane of physical lane `tid`, and whether this lane keeps the virtual lane's draw number j
  // (VL = 64: the thread's own index within its node's group, which may span several waves)
  __device__ __forceinline__ static int vlane(int tid) { return VL == 64 ? tid : (tid & (VL - 1)); }
  __device__ __forceinline__ static bool keeps(int tid, uint32_t j) { return VL == 64 || (j & 1u) == (uint32_t)(tid >> 5); }
  static constexpr int PER = 64 / VL;  // virtual columns per physical column: draw j belongs to physical column j / PER
};

// value of `v` in lane `src` of the wave, for a wave-UNIFORM src (0 .. 63): two v_readlane_b32 with a scalar lane select
// instead of the ds_bpermute round trips of __shfl (which cannot know that the index is uniform)
__device__ __forceinline__ unsigned long long wave_lane_u64(unsigned long long v, uint32_t src) {
  const int s = __builtin_amdgcn_readfirstlane((int)src);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, s);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), s);
  return ((unsigned long long)hi << 32) | lo;
}

// ----------------------------------------------------------------- virtual group sum (any K / L)
// The generic kernels (the shapes no specialised kernel takes: more columns per reference work-item than fit a lane's
// registers -- K = 1024 at the reference's default work-group size 32 for the gradient, K = 4096 at 32 for all three)
// spread the ELEMENTWISE work of a virtual group over all T = blockDim.x threads of a block (thread t owns columns
// t, t + T, ...) whatever the reference work-group size L is.  Only what depends on L is emulated lane by lane: the
// association order of WG_SUM -- virtual lane l < L chains vals[l], vals[l + L], ... in ascending order
// (sum.cc:20-22), then the halving tree aux[l] += aux[l + p2] over the L lane partials (sum.cc:23-29) -- and the
// stream-to-column map of the noise (in the kernels).  N sums at once: chain i runs on threads [i L, (i + 1) L).
//
// Block-uniform; needs N L <= T (and T >= 64 N when L > 64), vals[] visible to the block (barrier before the call);
// aux: [N L] floats of LDS, res: [2 N] floats of LDS (double-buffered through `phase`).  On return every thread holds
// the N sums and vals[] may be overwritten.
template <int N>
__device__ __forceinline__ void vgroup_sum(const float* const (&vals)[N], uint32_t K, uint32_t L, uint32_t lgL,
                                           float* aux, float* res, int& phase, float (&out)[N]) {
  const uint32_t t = threadIdx.x;
  const uint32_t which = t >> lgL, vl = t & (L - 1);
  float s = 0.0f;
  if (which < (uint32_t)N) {
    const float* v = vals[0];
#pragma unroll
    for (int i = 1; i < N; ++i) v = which == (uint32_t)i ? vals[i] : v;
    // the adds are the reference's chain (sequential by definition); the LDS reads feeding it are independent and are
    // issued eight at a time -- one read per add at LDS latency made this loop the whole run time of the generic
    // kernels (K = 4096 at wg 32: 128 dependent round trips per row)
    uint32_t k = vl;
    for (; k + 7 * L < K; k += 8 * L) {
      float x[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = v[k + i * L];
#pragma unroll
      for (int i = 0; i < 8; ++i) s += x[i];
    }
    for (; k < K; k += L) s += v[k];
  }
  float* r = res + phase * N;
  phase ^= 1;
  if (L > 64) {
    if (which < (uint32_t)N) aux[t] = s;  // which * L + vl == t
    __syncthreads();
    for (uint32_t lp = lgL - 1; lp >= 6; --lp) {  // p2 = L/2 .. 64
      const uint32_t p2 = 1u << lp, h = t >> lp, l = t & (p2 - 1);
      if (h < (uint32_t)N) aux[h * L + l] += aux[h * L + l + p2];
      __syncthreads();
    }
    const uint32_t w = t >> 6;  // wave w finishes sum w from its 64 remaining partials
    s = w < (uint32_t)N ? aux[w * L + (t & 63)] : 0.0f;
#pragma unroll
    for (int p2 = 32; p2 > 0; p2 >>= 1) s += __shfl_xor(s, p2, 64);
    if (w < (uint32_t)N && (t & 63) == 0) r[w] = s;
  } else {
    // L <= 64: chain i sits in the aligned L-lane block [i L, (i + 1) L) of one wave; a + b == b + a bitwise, so the
    // XOR butterfly leaves the tree's root in every lane of the block (Group<L>::wave_tree)
    for (uint32_t p2 = L >> 1; p2 > 0; p2 >>= 1) s += __shfl_xor(s, (int)p2, 64);
    if (which < (uint32_t)N && vl == 0) r[which] = s;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < N; ++i) out[i] = r[i];
}

__host__ __device__ inline uint32_t ilog2_u32(uint32_t v) {
  uint32_t r = 0;
  while (v > 1) {
    v >>= 1;
    ++r;
  }
  return r;
}

}  // namespace ammsb
