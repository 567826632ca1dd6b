/*
 * ammsb_oracle.c -- CPU restatement (plain C) of the reference hot path.
 * TEST INFRASTRUCTURE ONLY; see ammsb_oracle.h for the pin status
 * ("PARITY UNPINNED" for phi/beta/perplexity values) and the arithmetic
 * contract.  Every function cites the reference file:line it restates
 * (paths relative to the reference checkout).
 *
 * Virtual execution model: the reference kernels are grid-stride loops over
 * OpenCL work-items.  Here each virtual work-item / work-group is one
 * iteration of an ordinary C loop; "lane l of group g" is spelled out so the
 * RNG stream <-> (group, lane) mapping and the WG_SUM summation order are
 * explicit.  OpenMP (optional) parallelises across virtual groups only --
 * groups never communicate, so results do not depend on the thread count.
 */
#define _GNU_SOURCE
#include "ammsb_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "zig_tables.inc" /* orc_ytab / orc_wtab / orc_ktab, see tools/gen_ziggurat_tables.py */

#define ORC_MAX_GROUPS 65535u /* types.cc:537 GetMaxGroups() */

static int g_threads = 0;
int orc_num_threads(void) {
#ifdef _OPENMP
  return g_threads > 0 ? g_threads : omp_get_max_threads();
#else
  return 1;
#endif
}
void orc_set_num_threads(int n) {
  g_threads = n;
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#endif
}

/* transcendental contract: binary64 evaluation, one rounding to binary32 */
static inline float expf_cr(float x) { return (float)exp((double)x); }
static inline float logf_cr(float x) { return (float)log((double)x); }
static inline float powf_cr(float x, float y) { return (float)pow((double)x, (double)y); }

/* ------------------------------------------------------------------ params */

/* config.cc:57-64 float_to_string: std::scientific (6 digits) + "f" suffix,
 * parsed by the OpenCL compiler as a float literal. */
float orc_quantize_param(float v) {
  char buf[64];
  snprintf(buf, sizeof buf, "%e", (double)v);
  return strtof(buf, NULL);
}

/* learner.cc:41-43: EPS_A * pow(1 + step_count / EPS_B, -EPS_C) */
float orc_eps_t(const orc_params_t* p, uint32_t step_count) {
  float t = 1.0f + (float)step_count / p->b;
  return p->a * powf_cr(t, -p->c);
}

/* --------------------------------------------------------------------- RNG */

/* random.cc:31-43 RandomInit */
void orc_rng_init(orc_seed_t* seeds, uint64_t n, uint64_t sx, uint64_t sy) {
  for (uint64_t i = 0; i < n; ++i) {
    seeds[i].x = sx + i;
    seeds[i].y = sy + i;
  }
}

/* random.cl.inc:13-25 xorshift_128plus; :27-32 rand */
uint64_t orc_rand(orc_seed_t* s) {
  uint64_t s1 = s->x;
  uint64_t s0 = s->y;
  s->x = s0;
  s1 ^= s1 << 23;
  s->y = s1 ^ s0 ^ (s1 >> 17) ^ (s0 >> 26);
  return s->y + s0;
}

/* random.cl.inc:34-35: (1.0f * rand(s) / ULONG_MAX); ULONG_MAX -> float = 2^64 */
float orc_random(orc_seed_t* s) {
  float r = 1.0f * (float)orc_rand(s);
  return r / 18446744073709551616.0f;
}

/* random.cl.inc:37-39 (the "#if 1" branch; modulo bias kept) */
int32_t orc_randint(orc_seed_t* s, int32_t from, int32_t upto) {
  uint64_t range = (uint64_t)(int64_t)(upto + 1 - from);
  return (int32_t)((orc_rand(s) % range) + (uint64_t)(int64_t)from);
}

/* random.cl.inc:229-273 gsl_ran_gaussian_ziggurat(sigma = 1), range = 0xffffffff branch */
float orc_randn(orc_seed_t* s) {
  const float PARAM_R = 3.44428647676f; /* random.cl.inc:4 */
  uint64_t i, j;
  int sign;
  float x, y;
  for (;;) {
    uint64_t k = orc_rand(s);
    i = k & 0xFF;
    j = (k >> 8) & 0xFFFFFF;
    sign = (i & 0x80) ? +1 : -1;
    i &= 0x7f;
    x = (float)j * orc_wtab[i];
    if (j < orc_ktab[i]) break;
    if (i < 127) {
      float y0 = orc_ytab[i], y1 = orc_ytab[i + 1];
      float U1 = orc_random(s);
      float d = y0 - y1;
      float m = d * U1;
      y = y1 + m;
    } else {
      float U1 = 1.0f - orc_random(s);
      float U2 = orc_random(s);
      float l = logf_cr(U1) / PARAM_R;
      x = PARAM_R - l;
      float h = 0.5f * PARAM_R;
      float t = x - h;
      float a = -PARAM_R * t;
      y = expf_cr(a) * U2;
    }
    float hx = -0.5f * x;
    float xx = hx * x;
    if (y < expf_cr(xx)) break;
  }
  float ss = (float)sign * 1.0f;
  return ss * x;
}

/* random.cl.inc:310-317 */
static float orc_uniform_pos(orc_seed_t* s) {
  float x;
  do {
    x = orc_random(s);
  } while (x == 0);
  return x;
}

/* random.cl.inc:353-395 gsl_ran_gamma, non-recursive branch */
float orc_rand_gamma(orc_seed_t* s, float a, float b) {
  float f = 1.0f;
  while (a < 1) {
    float u = orc_uniform_pos(s);
    float ia = 1.0f / a;
    f = f * powf_cr(u, ia);
    a = 1.0f + a;
  }
  float x, v, u;
  const float third = 1.0f / 3.0f;
  float d = a - third;
  float c = third / sqrtf(d);
  for (;;) {
    do {
      x = orc_randn(s);
      float cx = c * x;
      v = 1.0f + cx;
    } while (v <= 0);
    float v2 = v * v;
    v = v2 * v;
    u = orc_uniform_pos(s);
    float q = 0.0331f * x;
    q = q * x;
    q = q * x;
    q = q * x;
    if (u < 1.0f - q) break;
    float hx = 0.5f * x;
    float hxx = hx * x;
    float omv = 1.0f - v;
    float in = omv + logf_cr(v);
    float din = d * in;
    if (logf_cr(u) < hxx + din) break;
  }
  float r = f * b;
  r = r * d;
  return r * v;
}

void orc_fill_rand(orc_seed_t* s, uint64_t* out, uint64_t n) {
  for (uint64_t i = 0; i < n; ++i) out[i] = orc_rand(s);
}
void orc_fill_random(orc_seed_t* s, float* out, uint64_t n) {
  for (uint64_t i = 0; i < n; ++i) out[i] = orc_random(s);
}
void orc_fill_randn(orc_seed_t* s, float* out, uint64_t n) {
  for (uint64_t i = 0; i < n; ++i) out[i] = orc_randn(s);
}
void orc_fill_gamma(orc_seed_t* s, float a, float b, float* out, uint64_t n) {
  for (uint64_t i = 0; i < n; ++i) out[i] = orc_rand_gamma(s, a, b);
}

/* ------------------------------------------------------------------ cuckoo */

#define ORC_NUM_BUCKETS 2u
#define ORC_NUM_SLOTS 4u
#define ORC_KEY_INVALID UINT64_MAX /* cuckoo.cc:91 */
static const uint64_t ORC_PRIMES[4][2] = {/* cuckoo.cc:92-96 */
                                          {15485807ull, 920429591ull},
                                          {379906717ull, 740320571ull},
                                          {256204747ull, 379927517ull},
                                          {13ull, 17ull}};

/* cuckoo.cc:98-104: N_ = 1 + ceil(1.15 n / (2*4)) */
uint64_t orc_set_num_bins(uint64_t n) {
  return (uint64_t)(1 + ceil((1.15 * (double)n) / (double)(ORC_NUM_BUCKETS * ORC_NUM_SLOTS)));
}

/* cuckoo.cc:197-206 Hash (64-bit wrap-around product) */
static inline uint64_t orc_hash(uint64_t k, unsigned bidx, uint32_t pidx, uint64_t bins) {
  return bidx == 0 ? (ORC_PRIMES[pidx][0] * k) % bins : (k ^ ORC_PRIMES[pidx][1]) % bins;
}

static inline uint64_t* orc_bin(uint64_t* slots, uint64_t bins, unsigned bidx, uint64_t h) {
  return slots + ((uint64_t)bidx * bins + h) * ORC_NUM_SLOTS;
}

/* cuckoo.cc:131-138 */
static int orc_slot_notfull_and_absent(uint64_t k, const uint64_t* slot) {
  int full = 1;
  for (unsigned i = 0; i < ORC_NUM_SLOTS; ++i) {
    if (slot[i] == ORC_KEY_INVALID) full = 0;
    if (slot[i] == k) return 0;
  }
  return !full;
}

/* cuckoo.cc:184-195 */
static uint64_t orc_insert_in_slot(uint64_t k, uint64_t* slot, unsigned* seed) {
  for (unsigned i = 0; i < ORC_NUM_SLOTS; ++i) {
    if (slot[i] == ORC_KEY_INVALID) {
      slot[i] = k;
      return ORC_KEY_INVALID;
    }
  }
  unsigned alt = (unsigned)rand_r(seed) % ORC_NUM_SLOTS;
  uint64_t old = slot[alt];
  slot[alt] = k;
  return old;
}

/* cuckoo.cc:140-161 Insert */
static int orc_set_insert(orc_set_t* set, uint64_t k, unsigned* seed, uint64_t disp_max) {
  uint64_t displacements = 0;
  do {
    for (unsigned b = 0; b < ORC_NUM_BUCKETS; ++b) {
      uint64_t h = orc_hash(k, b, set->prime_idx, set->num_bins);
      uint64_t* slot = orc_bin(set->slots, set->num_bins, b, h);
      if (orc_slot_notfull_and_absent(k, slot)) {
        orc_insert_in_slot(k, slot, seed);
        ++set->count;
        return 1;
      }
    }
    unsigned b = (unsigned)rand_r(seed) % ORC_NUM_BUCKETS;
    uint64_t h = orc_hash(k, b, set->prime_idx, set->num_bins);
    k = orc_insert_in_slot(k, orc_bin(set->slots, set->num_bins, b, h), seed);
  } while (++displacements < disp_max);
  return 0;
}

/* cuckoo.cc:98-129: Set(n) + SetContents(); seed_ = 42 persists across the
 * prime-pair retries and count_ is never reset (both as in the reference). */
int orc_set_build(orc_set_t* set, const uint64_t* keys, uint64_t n) {
  set->num_bins = orc_set_num_bins(n);
  set->count = 0;
  uint64_t cap = ORC_NUM_BUCKETS * set->num_bins * ORC_NUM_SLOTS;
  set->slots = (uint64_t*)malloc(cap * sizeof(uint64_t));
  if (!set->slots) return -2;
  unsigned seed = 42;
  uint64_t disp_max = n / 2 + 1;
  for (set->prime_idx = 0; set->prime_idx < 4; ++set->prime_idx) {
    for (uint64_t i = 0; i < cap; ++i) set->slots[i] = ORC_KEY_INVALID;
    int ok = 1;
    for (uint64_t i = 0; ok && i < n; ++i) ok = orc_set_insert(set, keys[i], &seed, disp_max);
    if (ok) return 0;
  }
  return -1;
}

void orc_set_free(orc_set_t* set) {
  free(set->slots);
  set->slots = NULL;
}

/* cuckoo.cc:39-65 Set_HasEdge (device) == cuckoo.cc:163-170 Has (host) */
int orc_set_has(const uint64_t* slots, uint64_t bins, uint32_t pidx, uint64_t k) {
  for (unsigned b = 0; b < ORC_NUM_BUCKETS; ++b) {
    uint64_t h = orc_hash(k, b, pidx, bins);
    const uint64_t* slot = slots + ((uint64_t)b * bins + h) * ORC_NUM_SLOTS;
    for (unsigned i = 0; i < ORC_NUM_SLOTS; ++i)
      if (slot[i] == k) return 1;
  }
  return 0;
}

void orc_set_has_many(const uint64_t* slots, uint64_t bins, uint32_t pidx, const uint64_t* keys,
                      uint64_t n, uint8_t* out) {
  for (uint64_t i = 0; i < n; ++i) out[i] = (uint8_t)orc_set_has(slots, bins, pidx, keys[i]);
}

/* --------------------------------------------------- row-partitioned matrix */

/* partitioned-alloc.h:22-29 (offset widened to 64 bit; the reference's uint
 * product overflows once rows_in_block * cols >= 2^32) */
void orc_rpm_locate(uint64_t rows_in_block, uint64_t num_cols, uint64_t row, uint32_t* block,
                    uint64_t* offset) {
  *block = (uint32_t)(row / rows_in_block);
  *offset = (row % rows_in_block) * num_cols;
}

/* ----------------------------------------------------------- wg primitives */

/* sum.cc:11-18 */
static inline uint32_t power_of_2(uint32_t v) {
  v |= v >> 1;
  v |= v >> 2;
  v |= v >> 4;
  v |= v >> 8;
  v |= v >> 16;
  return v + 1;
}

/* sum.cc:20-29 WG_SUM_TT_LOCAL_: all lanes act synchronously per step; for a
 * given p2 the reads (lid + p2 >= p2) and writes (lid < p2) do not overlap. */
static void wg_tree_f32(float* aux, uint32_t L) {
  for (uint32_t p2 = power_of_2(L) >> 1; p2 > 0; p2 >>= 1)
    for (uint32_t lid = 0; lid < p2 && lid + p2 < L; ++lid) aux[lid] += aux[lid + p2];
}
static void wg_tree_u32(uint32_t* aux, uint32_t L) {
  for (uint32_t p2 = power_of_2(L) >> 1; p2 > 0; p2 >>= 1)
    for (uint32_t lid = 0; lid < p2 && lid + p2 < L; ++lid) aux[lid] += aux[lid + p2];
}

/* sum.cc:31-42 WG_SUM_TT: per-lane strided partial, then the tree */
static float wg_sum_f32(const float* in, uint32_t len, uint32_t L, float* aux) {
  for (uint32_t lid = 0; lid < L; ++lid) {
    float lsum = 0;
    for (uint32_t i = lid; i < len; i += L) lsum += in[i];
    aux[lid] = lsum;
  }
  wg_tree_f32(aux, L);
  return aux[0];
}

float orc_wg_sum_f32(const float* in, uint32_t len, uint32_t L) {
  float* aux = (float*)malloc(sizeof(float) * L);
  float r = wg_sum_f32(in, len, L, aux);
  free(aux);
  return r;
}

uint32_t orc_wg_sum_u32(const uint32_t* in, uint32_t len, uint32_t L) {
  uint32_t* aux = (uint32_t*)malloc(sizeof(uint32_t) * L);
  for (uint32_t lid = 0; lid < L; ++lid) {
    uint32_t lsum = 0;
    for (uint32_t i = lid; i < len; i += L) lsum += in[i];
    aux[lid] = lsum;
  }
  wg_tree_u32(aux, L);
  uint32_t r = aux[0];
  free(aux);
  return r;
}

/* normalize.cc:13-23 WG_NORMALIZE_TT */
static float wg_normalize_f32(float* inout, uint32_t len, uint32_t L, float* aux) {
  float sum = wg_sum_f32(inout, len, L, aux);
  for (uint32_t i = 0; i < len; ++i) inout[i] = inout[i] / sum;
  return sum;
}

float orc_wg_normalize_f32(float* inout, uint32_t len, uint32_t L) {
  float* aux = (float*)malloc(sizeof(float) * L);
  float s = wg_normalize_f32(inout, len, L, aux);
  free(aux);
  return s;
}

/* sort.cc:11-32 WG_SORT_TT: bitonic network over L = len lanes, ties broken by index */
#define ORC_DEFINE_SORT(NAME, T)                                            \
  void NAME(const T* in, T* out, uint32_t len) {                            \
    T* aux = (T*)malloc(sizeof(T) * len);                                   \
    T* nxt = (T*)malloc(sizeof(T) * len);                                   \
    memcpy(aux, in, sizeof(T) * len);                                       \
    for (size_t length = 1; length < len; length <<= 1) {                   \
      for (size_t inc = length; inc > 0; inc >>= 1) {                       \
        for (size_t i = 0; i < len; ++i) {                                  \
          int direction = ((i & (length << 1)) != 0);                       \
          size_t j = i ^ inc;                                               \
          T idata = aux[i], jdata = aux[j];                                 \
          int smaller = (jdata < idata) || (jdata == idata && j < i);       \
          int swap = smaller ^ (j < i) ^ direction;                         \
          nxt[i] = swap ? jdata : idata;                                    \
        }                                                                   \
        T* t = aux;                                                         \
        aux = nxt;                                                          \
        nxt = t;                                                            \
      }                                                                     \
    }                                                                       \
    memcpy(out, aux, sizeof(T) * len);                                      \
    free(aux);                                                              \
    free(nxt);                                                              \
  }
ORC_DEFINE_SORT(orc_wg_sort_u32, uint32_t)
ORC_DEFINE_SORT(orc_wg_sort_f32, float)

/* ----------------------------------------------------------------- pi init */

/* random.cc:108-167: generate_gamma launched with G = min(N, 65535) groups of
 * 32 lanes; group i fills rows i, i+G, ...; lane lid fills columns lid,
 * lid+32, ... from stream seeds[i*32 + lid] (seed base {11,113} at the call
 * site, N*32 streams).  Then PartitionedNormalizer with wg = 32
 * (normalize.cc:34-52): WG_SUM over the row, row /= sum, g_sum[row] = sum. */
void orc_pi_init_gamma(float* pi, float* phi_sum, uint64_t N, uint64_t K, float eta0, float eta1,
                       uint64_t sx, uint64_t sy) {
  const uint32_t L = 32;
  uint64_t G = N < ORC_MAX_GROUPS ? N : ORC_MAX_GROUPS;
#pragma omp parallel for schedule(dynamic, 16)
  for (uint64_t g = 0; g < G; ++g) {
    orc_seed_t seed[32];
    float aux[32];
    for (uint32_t l = 0; l < L; ++l) {
      seed[l].x = sx + g * L + l;
      seed[l].y = sy + g * L + l;
    }
    for (uint64_t row = g; row < N; row += G) {
      float* r = pi + row * K;
      for (uint32_t l = 0; l < L; ++l)
        for (uint64_t j = l; j < K; j += L) r[j] = orc_rand_gamma(&seed[l], eta0, eta1);
      phi_sum[row] = wg_normalize_f32(r, (uint32_t)K, L, aux);
    }
  }
}

/* ------------------------------------------------------- neighbour sampler */

/* sample.cc:15-21 */
static inline uint32_t ns_h1(uint32_t k, uint32_t capacity) { return (k ^ 553105253u) % capacity; }
static inline uint32_t ns_h2(uint32_t capacity) { return 1u + (capacity << 1); }

/* sample.cc:23-46 generate_random_int */
static void ns_generate(orc_seed_t* seed, uint32_t* out, uint32_t capacity, uint32_t max_id,
                        uint32_t node) {
  uint32_t r, val;
  do {
    do {
      r = (uint32_t)orc_randint(seed, 0, (int32_t)max_id);
    } while (r == node);
    uint32_t l1 = ns_h1(r, capacity);
    uint32_t l2 = ns_h2(capacity);
    for (uint32_t i = 0;; ++i) {
      uint32_t offset = (l1 + i * l2) % capacity;
      val = out[offset];
      if (val == r) break;
      if (val == max_id + 1) {
        out[offset] = r;
        break;
      }
    }
  } while (val == r);
}

/* sample.cc:48-77 kernel, :111-121 launch: global = min(ceil(ns/wg), 65535/wg) * wg threads */
void orc_sample_neighbors(orc_seed_t* seeds, const uint32_t* nodes, uint32_t n_nodes, uint32_t N,
                          uint32_t n, uint32_t wg, uint32_t* table, uint32_t* packed) {
  uint32_t groups = n_nodes / wg + (n_nodes % wg ? 1 : 0);
  uint32_t maxg = ORC_MAX_GROUPS / wg;
  if (groups > maxg) groups = maxg;
  uint32_t gsize = groups * wg;
  uint32_t capacity = 2 * n;
#pragma omp parallel for schedule(static)
  for (uint32_t gid = 0; gid < gsize; ++gid) {
    if (gid >= n_nodes) continue;
    orc_seed_t seed = seeds[gid];
    for (uint32_t i = gid; i < n_nodes; i += gsize) {
      uint32_t* out = table + (uint64_t)i * capacity;
      uint32_t* pk = packed + (uint64_t)i * n;
      uint32_t node = nodes[i];
      for (uint32_t j = 0; j < capacity; ++j) out[j] = N;
      for (uint32_t j = 0; j < n; ++j) ns_generate(&seed, out, capacity, N - 1, node);
      uint32_t count = 0;
      for (uint32_t j = 0; j < capacity && count < n; ++j)
        if (out[j] != N) pk[count++] = out[j];
    }
    seeds[gid] = seed;
  }
}

/* ---------------------------------------------------------------- phi / pi */

/* learner.cc:22-27 MakeEdge(min, max) as used at phi.cc:96,237 */
static inline uint64_t make_edge(uint32_t a, uint32_t b) {
  uint32_t u = a < b ? a : b, v = a < b ? b : a;
  return ((uint64_t)u << 32) | v;
}

/* the final SGLD step shared by all phi variants: phi.cc:116-120 / :269-273 */
static inline float phi_step(float pi_k, float phi_sum, float grads_k, float eps_t, float alpha,
                             float Nn, float noise) {
  float phi_k = pi_k * phi_sum;
  float half = eps_t / 2;
  float ng = Nn * grads_k;
  float in = alpha - phi_k;
  in = in + ng;
  float drift = half * in;
  float a = phi_k + drift;
  float ep = eps_t * phi_k;
  float sq = sqrtf(ep);
  float b = sq * noise;
  float v = fabsf(a + b);
  return v > 1e-24f ? v : 1e-24f; /* MAX(v, 1e-24f) */
}

/* phi.cc:78-122 update_phi_for_node (thread mode) */
static void phi_node_thread(const orc_params_t* p, const float* beta, const float* pi,
                            const float* g_phi, const uint64_t* ss, uint64_t sb, uint32_t sp,
                            uint32_t node, const uint32_t* neighbors, float eps_t, float* grads,
                            float* probs, orc_seed_t* rseed, int noise_on, float* phi_vec) {
  const uint64_t K = p->K;
  const float EPS = p->epsilon;
  const float* pi_a = pi + (uint64_t)node * K;
  float phi_sum = g_phi[node];
  for (uint64_t k = 0; k < K; ++k) grads[k] = 0;
  for (uint32_t i = 0; i < p->n_neighbors; ++i) {
    uint32_t nb = neighbors[i];
    const float* pi_n = pi + (uint64_t)nb * K;
    int y = orc_set_has(ss, sb, sp, make_edge(node, nb));
    float e = y ? EPS : 1.0f - EPS;
    float probs_sum = 0;
    for (uint64_t k = 0; k < K; ++k) {
      float beta_k = beta[2 * k + 1];
      float f = y ? (beta_k - EPS) : (EPS - beta_k);
      float t = pi_n[k] * f;
      t = t + e;
      float pk = pi_a[k] * t;
      probs_sum += pk;
      probs[k] = pk;
    }
    for (uint64_t k = 0; k < K; ++k) {
      float q = probs[k] / probs_sum;
      float den = pi_a[k] * phi_sum;
      q = q / den;
      float inv = 1.0f / phi_sum;
      grads[k] += q - inv;
    }
  }
  float Nn = (1.0f * (float)p->N) / (float)p->n_neighbors;
  for (uint64_t k = 0; k < K; ++k) {
    float noise = noise_on ? orc_randn(rseed) : 1.0f; /* phi.cc:673-677 */
    phi_vec[k] = phi_step(pi_a[k], phi_sum, grads[k], eps_t, p->alpha, Nn, noise);
  }
}

/* phi.cc:214-275 update_phi_for_nodeWG: L lanes, K strided over lanes */
static void phi_node_wg(const orc_params_t* p, const float* beta, const float* pi,
                        const float* g_phi, const uint64_t* ss, uint64_t sb, uint32_t sp,
                        uint32_t node, const uint32_t* neighbors, float eps_t, uint32_t L,
                        float* grads, float* probs, float* aux, orc_seed_t* lane_seeds,
                        int noise_on, float* phi_vec) {
  const uint64_t K = p->K;
  const float EPS = p->epsilon;
  const float* pi_a = pi + (uint64_t)node * K;
  float phi_sum = g_phi[node];
  for (uint64_t k = 0; k < K; ++k) grads[k] = 0;
  for (uint32_t i = 0; i < p->n_neighbors; ++i) {
    uint32_t nb = neighbors[i];
    const float* pi_n = pi + (uint64_t)nb * K;
    int y = orc_set_has(ss, sb, sp, make_edge(node, nb));
    float e = y ? EPS : 1.0f - EPS;
    for (uint64_t k = 0; k < K; ++k) {
      float beta_k = beta[2 * k + 1];
      float f = y ? (beta_k - EPS) : (EPS - beta_k);
      float t = pi_n[k] * f;
      t = t + e;
      probs[k] = pi_a[k] * t;
    }
    float probs_sum = wg_sum_f32(probs, (uint32_t)K, L, aux); /* phi.cc:250-257 */
    for (uint64_t k = 0; k < K; ++k) {
      float q = probs[k] / probs_sum;
      float den = pi_a[k] * phi_sum;
      q = q / den;
      float inv = 1.0f / phi_sum;
      grads[k] += q - inv;
    }
  }
  float Nn = (1.0f * (float)p->N) / (float)p->n_neighbors;
  /* phi.cc:266-274: lane l draws for k = l, l+L, ... in that order */
  for (uint32_t l = 0; l < L; ++l)
    for (uint64_t k = l; k < K; k += L) {
      float noise = noise_on ? orc_randn(&lane_seeds[l]) : 1.0f;
      phi_vec[k] = phi_step(pi_a[k], phi_sum, grads[k], eps_t, p->alpha, Nn, noise);
    }
}

/* phi.cc:124-152 (thread) / :277-302 (WG) update_phi; launch shape phi.cc:740-747 */
void orc_update_phi(const orc_params_t* p, const float* beta, const float* pi, const float* phi_sum,
                    const uint64_t* ss, uint64_t sb, uint32_t sp, const uint32_t* nodes,
                    const uint32_t* neighbors, uint32_t n_nodes, uint32_t step_count,
                    orc_seed_t* seeds, uint32_t L, int mode_wg, int noise_on, float* phi_vec) {
  const uint64_t K = p->K;
  const uint32_t n = p->n_neighbors;
  float eps_t = orc_eps_t(p, step_count);
  if (!mode_wg) {
    uint32_t groups = n_nodes / L + (n_nodes % L ? 1 : 0);
    if (groups > ORC_MAX_GROUPS) groups = ORC_MAX_GROUPS;
    uint64_t gsize = (uint64_t)groups * L;
#pragma omp parallel
    {
      float* grads = (float*)malloc(sizeof(float) * K);
      float* probs = (float*)malloc(sizeof(float) * K);
#pragma omp for schedule(dynamic, 8)
      for (uint64_t t = 0; t < gsize; ++t) {
        if (t >= n_nodes) continue;
        orc_seed_t rseed = seeds[t];
        for (uint64_t i = t; i < n_nodes; i += gsize)
          phi_node_thread(p, beta, pi, phi_sum, ss, sb, sp, nodes[i], neighbors + i * n, eps_t,
                          grads, probs, &rseed, noise_on, phi_vec + i * K);
        seeds[t] = rseed;
      }
      free(grads);
      free(probs);
    }
  } else {
    uint32_t G = n_nodes < ORC_MAX_GROUPS ? n_nodes : ORC_MAX_GROUPS;
#pragma omp parallel
    {
      float* grads = (float*)malloc(sizeof(float) * K);
      float* probs = (float*)malloc(sizeof(float) * K);
      float* aux = (float*)malloc(sizeof(float) * L);
#pragma omp for schedule(dynamic, 8)
      for (uint32_t g = 0; g < G; ++g) {
        orc_seed_t* lane_seeds = seeds + (uint64_t)g * L; /* base_[GET_GLOBAL_ID()] */
        for (uint64_t i = g; i < n_nodes; i += G)
          phi_node_wg(p, beta, pi, phi_sum, ss, sb, sp, nodes[i], neighbors + i * n, eps_t, L,
                      grads, probs, aux, lane_seeds, noise_on, phi_vec + i * K);
      }
      free(grads);
      free(probs);
      free(aux);
    }
  }
}

/* phi.cc:154-173 (thread) / :177-197 (WG) update_pi */
void orc_update_pi(const orc_params_t* p, float* pi, float* phi_sum, const float* phi_vec,
                   const uint32_t* nodes, uint32_t n_nodes, uint32_t L, int mode_wg) {
  const uint64_t K = p->K;
#pragma omp parallel
  {
    float* aux = (float*)malloc(sizeof(float) * (L ? L : 1));
#pragma omp for schedule(static)
    for (uint32_t i = 0; i < n_nodes; ++i) {
      uint32_t nd = nodes[i];
      float* row = pi + (uint64_t)nd * K;
      const float* phi = phi_vec + (uint64_t)i * K;
      if (!mode_wg) {
        float sum = 0;
        for (uint64_t k = 0; k < K; ++k) sum += phi[k];
        for (uint64_t k = 0; k < K; ++k) row[k] = phi[k] / sum;
        phi_sum[nd] = sum;
      } else {
        for (uint64_t k = 0; k < K; ++k) row[k] = phi[k];
        phi_sum[nd] = wg_normalize_f32(row, (uint32_t)K, L, aux);
      }
    }
    free(aux);
  }
}

/* ------------------------------------------------------------ beta / theta */

/* beta.cc:30-37 */
void orc_sum_theta(const float* theta, float* theta_sum, uint64_t K) {
  for (uint64_t k = 0; k < K; ++k) theta_sum[k] = theta[2 * k] + theta[2 * k + 1];
}

/* One edge's contribution added into acc[2K].
 * thread mode: beta.cc:101-135; WG mode: beta.cc:145-171,195-223.
 * The two differ only in how pi_sum / probs_sum are reduced. */
static void beta_edge(const orc_params_t* p, const float* theta, const float* theta_sum,
                      const float* beta, const float* pi, const uint64_t* ss, uint64_t sb,
                      uint32_t sp, uint64_t edge, uint32_t L, int mode_wg, float* probs, float* fbuf,
                      float* aux, float* acc, double* dacc) {
  const uint64_t K = p->K;
  const float EPS = p->epsilon;
  uint32_t u = (uint32_t)(edge >> 32), v = (uint32_t)(edge & 0xffffffffu);
  uint32_t y = (uint32_t)orc_set_has(ss, sb, sp, make_edge(u, v));
  const float* pi_a = pi + (uint64_t)u * K;
  const float* pi_b = pi + (uint64_t)v * K;
  float pi_sum = 0, probs_sum = 0;
  if (!mode_wg) {
    for (uint64_t k = 0; k < K; ++k) {
      float f = pi_a[k] * pi_b[k];
      pi_sum += f;
      float beta_k = beta[2 * k + 1];
      float pk = y ? beta_k * f : (1.0f - beta_k) * f;
      probs[k] = pk;
      probs_sum += pk;
    }
  } else {
    for (uint64_t k = 0; k < K; ++k) {
      float f = pi_a[k] * pi_b[k];
      fbuf[k] = f;
      float beta_k = beta[2 * k + 1];
      probs[k] = y ? beta_k * f : (1.0f - beta_k) * f;
    }
    pi_sum = wg_sum_f32(fbuf, (uint32_t)K, L, aux);    /* "scratch" */
    probs_sum = wg_sum_f32(probs, (uint32_t)K, L, aux);
  }
  float w = y ? EPS : (1.0f - EPS);
  float prob_0 = w * (1.0f - pi_sum);
  probs_sum += prob_0;
  for (uint64_t k = 0; k < K; ++k) {
    float f = probs[k] / probs_sum;
    float one_over = 1.0f / theta_sum[k];
    float t0 = (float)(1 - y) / theta[2 * k];
    float t1 = (float)y / theta[2 * k + 1];
    float g0 = f * (t0 - one_over);
    float g1 = f * (t1 - one_over);
    if (dacc) {
      dacc[2 * k] += (double)g0;
      dacc[2 * k + 1] += (double)g1;
    } else {
      acc[2 * k] += g0;
      acc[2 * k + 1] += g1;
    }
  }
}

/* calculate_grads_partial + sum_grads.  Launch shapes beta.cc:345-364:
 * thread mode T = min(ceil(E/L), 65535) * L virtual threads, partial rows
 * P = min(T, E); WG mode G = min(E, 65535) groups, P = G. */
void orc_beta_grads(const orc_params_t* p, const float* theta, const float* theta_sum,
                    const float* beta, const float* pi, const uint64_t* ss, uint64_t sb, uint32_t sp,
                    const uint64_t* edges, uint32_t n_edges, uint32_t L, int mode_wg, int order,
                    float* grads_out) {
  const uint64_t K = p->K;
  uint64_t stride;
  if (!mode_wg) {
    uint32_t groups = n_edges / L + (n_edges % L ? 1 : 0);
    if (groups > ORC_MAX_GROUPS) groups = ORC_MAX_GROUPS;
    stride = (uint64_t)groups * L;
  } else {
    stride = n_edges < ORC_MAX_GROUPS ? n_edges : ORC_MAX_GROUPS;
  }
  uint64_t P = stride < n_edges ? stride : n_edges;
  if (order == 1) {
    double* total = (double*)calloc(2 * K, sizeof(double));
#pragma omp parallel
    {
      double* dacc = (double*)calloc(2 * K, sizeof(double));
      float* probs = (float*)malloc(sizeof(float) * K);
      float* fbuf = (float*)malloc(sizeof(float) * K);
      float* aux = (float*)malloc(sizeof(float) * L);
#pragma omp for schedule(static)
      for (uint32_t i = 0; i < n_edges; ++i)
        beta_edge(p, theta, theta_sum, beta, pi, ss, sb, sp, edges[i], L, mode_wg, probs, fbuf, aux,
                  NULL, dacc);
#pragma omp critical
      for (uint64_t j = 0; j < 2 * K; ++j) total[j] += dacc[j];
      free(dacc);
      free(probs);
      free(fbuf);
      free(aux);
    }
    for (uint64_t j = 0; j < 2 * K; ++j) grads_out[j] = (float)total[j];
    free(total);
    return;
  }
  float* part = (float*)calloc((size_t)P * 2 * K, sizeof(float));
#pragma omp parallel
  {
    float* probs = (float*)malloc(sizeof(float) * K);
    float* fbuf = (float*)malloc(sizeof(float) * K);
    float* aux = (float*)malloc(sizeof(float) * L);
#pragma omp for schedule(dynamic, 8)
    for (uint64_t g = 0; g < P; ++g)
      for (uint64_t i = g; i < n_edges; i += stride)
        beta_edge(p, theta, theta_sum, beta, pi, ss, sb, sp, edges[i], L, mode_wg, probs, fbuf, aux,
                  part + g * 2 * K, NULL);
    free(probs);
    free(fbuf);
    free(aux);
  }
  /* beta.cc:39-49 sum_grads: serial over partial rows, ascending */
#pragma omp parallel for schedule(static)
  for (uint64_t j = 0; j < 2 * K; ++j) {
    float sum = part[j];
    for (uint64_t q = 1; q < P; ++q) sum += part[j + q * 2 * K];
    grads_out[j] = sum;
  }
  free(part);
}

/* beta.cc:51-82 update_theta: stream k handles component k; r0 then r1 */
void orc_update_theta(const orc_params_t* p, float* theta, const float* grads, uint32_t step_count,
                      float scale, orc_seed_t* seeds, int noise_on) {
  float eps_t = orc_eps_t(p, step_count);
  float half = eps_t / 2.0f;
  for (uint64_t k = 0; k < p->K; ++k) {
    orc_seed_t rseed = seeds[k];
    for (int c = 0; c < 2; ++c) {
      float r = noise_on ? orc_randn(&rseed) : 1.0f;
      float g = grads[2 * k + c];
      float th = theta[2 * k + c];
      float eta = c == 0 ? p->eta0 : p->eta1;
      float ep = eps_t * th;
      float f = sqrtf(ep);
      float sg = scale * g;
      float in = eta - th;
      in = in + sg;
      float drift = half * in;
      float a = th + drift;
      float b = f * r;
      float v = fabsf(a + b);
      theta[2 * k + c] = v > 1e-24f ? v : 1e-24f;
    }
    seeds[k] = rseed;
  }
}

/* beta.cc:376-383: copy theta -> beta, Normalizer(slice = 2, wg = 1) */
void orc_beta_from_theta(const float* theta, float* beta, uint64_t K) {
  for (uint64_t k = 0; k < K; ++k) {
    float lsum = 0;
    lsum += theta[2 * k];
    lsum += theta[2 * k + 1];
    beta[2 * k] = theta[2 * k] / lsum;
    beta[2 * k + 1] = theta[2 * k + 1] / lsum;
  }
}

/* -------------------------------------------------------------- perplexity */

/* perplexity.cc:16-39 (thread) / :93-128 (WG) edge likelihood */
static float edge_likelihood(const orc_params_t* p, const float* pi_a, const float* pi_b,
                             const float* beta, int is_edge, uint32_t L, int mode_wg, float* scratch,
                             float* aux) {
  const uint64_t K = p->K;
  float s = 0;
  if (!mode_wg) {
    if (is_edge) {
      for (uint64_t k = 0; k < K; ++k) {
        float f = pi_a[k] * pi_b[k];
        s += f * beta[2 * k + 1];
      }
    } else {
      float sum = 0;
      for (uint64_t k = 0; k < K; ++k) {
        float f = pi_a[k] * pi_b[k];
        float ob = 1.0f - beta[2 * k + 1];
        s += f * ob;
        sum += f;
      }
      float t = 1.0f - sum;
      float u = 1.0f - p->epsilon;
      s += t * u;
    }
  } else {
    if (is_edge) {
      for (uint64_t k = 0; k < K; ++k) {
        float f = pi_a[k] * pi_b[k];
        scratch[k] = f * beta[2 * k + 1];
      }
      s = wg_sum_f32(scratch, (uint32_t)K, L, aux);
    } else {
      for (uint64_t k = 0; k < K; ++k) scratch[k] = pi_a[k] * pi_b[k];
      float sum = wg_sum_f32(scratch, (uint32_t)K, L, aux);
      for (uint64_t k = 0; k < K; ++k) {
        float f = pi_a[k] * pi_b[k];
        float ob = 1.0f - beta[2 * k + 1];
        scratch[k] = f * ob;
      }
      s = wg_sum_f32(scratch, (uint32_t)K, L, aux);
      float t = 1.0f - sum;
      float u = 1.0f - p->epsilon;
      s += t * u;
    }
  }
  if (s < 1.0e-30f) s = 1.0e-30f;
  return s;
}

/* perplexity.cc:41-65 per-edge running mean + log; :251-274 host accumulation.
 * Edges are used as stored (no canonicalisation, perplexity.cc:45-47). */
void orc_perplexity(const orc_params_t* p, const float* beta, const float* pi, const uint64_t* ss,
                    uint64_t sb, uint32_t sp, const uint64_t* edges, uint32_t n_edges,
                    uint32_t call_count, uint32_t L, int mode_wg, float* ppx_per_edge, float* edge_ll,
                    orc_ppx_sums_t* out) {
  const uint64_t K = p->K;
  double link_ll = 0, non_ll = 0;
  uint64_t link_cnt = 0, non_cnt = 0;
#pragma omp parallel
  {
    float* scratch = (float*)malloc(sizeof(float) * K);
    float* aux = (float*)malloc(sizeof(float) * (L ? L : 1));
#pragma omp for schedule(static) reduction(+ : link_ll, non_ll, link_cnt, non_cnt)
    for (uint32_t i = 0; i < n_edges; ++i) {
      uint64_t e = edges[i];
      uint32_t u = (uint32_t)(e >> 32), v = (uint32_t)(e & 0xffffffffu);
      int is_edge = orc_set_has(ss, sb, sp, e);
      float lik = edge_likelihood(p, pi + (uint64_t)u * K, pi + (uint64_t)v * K, beta, is_edge, L,
                                  mode_wg, scratch, aux);
      float ppx = ppx_per_edge[i];
      float m = ppx * (float)(call_count - 1);
      m = m + lik;
      ppx = m / (float)call_count;
      float ll = logf_cr(ppx);
      if (is_edge) {
        link_cnt += 1;
        link_ll += (double)ll;
      } else {
        non_cnt += 1;
        non_ll += (double)ll;
      }
      if (edge_ll) edge_ll[i] = ll;
      ppx_per_edge[i] = ppx;
    }
    free(scratch);
    free(aux);
  }
  out->link_ll = link_ll;
  out->nonlink_ll = non_ll;
  out->link_cnt = link_cnt;
  out->nonlink_cnt = non_cnt;
}

/* perplexity.cc:264-273 returns -avg; learner.cc:196-203 returns exp(that) */
double orc_ppx_value(const orc_ppx_sums_t* s) {
  double avg = 0.0;
  if (s->link_cnt + s->nonlink_cnt != 0)
    avg = (s->link_ll + s->nonlink_ll) / (double)(s->link_cnt + s->nonlink_cnt);
  return exp(-avg);
}
