/*
 * ammsb_oracle.h -- CPU restatement of the SG-MCMC a-MMSB hot path of
 * ielhelw/mcmc-ammsb-gpu.  TEST INFRASTRUCTURE ONLY.
 *
 * This library is the checker for the HIP kernels: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * product (mcmc-ammsb-gpu_amd/) never links, imports or calls it.
 *
 * PARITY PIN STATUS
 *   The reference cannot be built in this image (every translation unit needs
 *   glog / Boost / Boost.Compute / protobuf / CLCudaAPI, none of which exist
 *   here, and its kernels are OpenCL-C strings JIT-compiled at run time), and
 *   it ships no golden vectors.  The restatement is therefore pinned by the
 *   known-answer checks the reference's own tests hold:
 *     - wg-sum-test.cc:22-48      exact integer sums, length grid x wg grid
 *     - wg-normalize-test.cc:24-48  (i+1)/sum within 4 ULP
 *     - wg-sort-test.cc:23-44     == std::sort
 *     - cuckoo-test.cc:29-43      inserted keys present, others absent
 *     - random-test.cc:60-63      seed[i] = {sx+i, sy+i}
 *     - wg-sample-test.cc:47-68   neighbour-sampler table/packed invariants
 *     - test-partitioned-alloc.cc row -> (block, offset) addressing
 *     - wg-phi/beta/perplexity-test.cc  cross-mode agreement (2 % / 5 %)
 *   For the *values* of the phi / beta / perplexity updates the reference has
 *   no fixture and no runnable build here: those functions are
 *   "PARITY UNPINNED" -- a line-by-line restatement of the cited kernel text,
 *   cross-checked only against an independent float64 numpy model
 *   (tests/test_oracle_model.py).
 *
 * Arithmetic contract: IEEE-754 binary32, round-to-nearest-even, no FMA
 * contraction, operations in the order the reference kernel text writes them
 * (the reference itself is built with -cl-fast-relaxed-math, types.cc:522-535,
 * so its device results are implementation-defined; the IEEE evaluation of
 * its text is the target).  exp/log/pow are evaluated in binary64 and rounded
 * once to binary32.
 */
#ifndef AMMSB_ORACLE_H
#define AMMSB_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t x, y; } orc_seed_t; /* random.cc:13 ulong2 */

/* compile-time constants the reference bakes into kernels, config.cc:66-83 */
typedef struct {
  uint64_t N, K;
  uint32_t n_neighbors; /* NUM_NEIGHBORS */
  float alpha, a, b, c, epsilon, eta0, eta1;
} orc_params_t;

/* config.cc:57-64: floats reach the kernels as "%e" text + 'f' suffix */
float orc_quantize_param(float v);
/* learner.cc:41-43 get_eps_t */
float orc_eps_t(const orc_params_t* p, uint32_t step_count);

/* ---- RNG: random.cl.inc:13-395, random.cc:31-43 ---- */
void orc_rng_init(orc_seed_t* seeds, uint64_t n, uint64_t sx, uint64_t sy);
uint64_t orc_rand(orc_seed_t* s);
float orc_random(orc_seed_t* s);
int32_t orc_randint(orc_seed_t* s, int32_t from, int32_t upto);
float orc_randn(orc_seed_t* s);
float orc_rand_gamma(orc_seed_t* s, float a, float b);
/* bulk helpers for tests: n draws from one stream */
void orc_fill_rand(orc_seed_t* s, uint64_t* out, uint64_t n);
void orc_fill_random(orc_seed_t* s, float* out, uint64_t n);
void orc_fill_randn(orc_seed_t* s, float* out, uint64_t n);
void orc_fill_gamma(orc_seed_t* s, float a, float b, float* out, uint64_t n);

/* ---- cuckoo edge set: cuckoo.cc:92-220 (host), :27-69 (lookup) ---- */
typedef struct {
  uint64_t* slots;   /* [2][num_bins][4], layout of Set::Serialize() */
  uint64_t num_bins; /* N_ */
  uint32_t prime_idx;
  uint64_t count;
} orc_set_t;
uint64_t orc_set_num_bins(uint64_t n); /* cuckoo.cc:98-104 */
/* returns 0 on success, -1 if all 4 prime pairs fail */
int orc_set_build(orc_set_t* set, const uint64_t* keys, uint64_t n);
void orc_set_free(orc_set_t* set);
int orc_set_has(const uint64_t* slots, uint64_t num_bins, uint32_t prime_idx, uint64_t key);
void orc_set_has_many(const uint64_t* slots, uint64_t num_bins, uint32_t prime_idx,
                      const uint64_t* keys, uint64_t n, uint8_t* out);

/* ---- row-partitioned matrix addressing: partitioned-alloc.h:22-29 ---- */
void orc_rpm_locate(uint64_t rows_in_block, uint64_t num_cols, uint64_t row,
                    uint32_t* block, uint64_t* offset);

/* ---- wg_* primitives, L = virtual work-group size (any L >= 1) ---- */
float orc_wg_sum_f32(const float* in, uint32_t len, uint32_t L);   /* sum.cc:11-42 */
uint32_t orc_wg_sum_u32(const uint32_t* in, uint32_t len, uint32_t L);
float orc_wg_normalize_f32(float* inout, uint32_t len, uint32_t L); /* normalize.cc:13-23; returns sum */
void orc_wg_sort_u32(const uint32_t* in, uint32_t* out, uint32_t len); /* sort.cc:11-32, len = L pow2 */
void orc_wg_sort_f32(const float* in, float* out, uint32_t len);

/* ---- pi init: random.cc:108-167 (gamma rows, L=32 lanes, then normalise) ---- */
void orc_pi_init_gamma(float* pi, float* phi_sum, uint64_t N, uint64_t K, float eta0, float eta1,
                       uint64_t sx, uint64_t sy);

/* ---- neighbour sampler: sample.cc:13-78, launch shape :111-121 ---- */
/* seeds: stream array (>= launched threads), advanced in place. table: [n_nodes, 2n]; packed: [n_nodes, n] */
void orc_sample_neighbors(orc_seed_t* seeds, const uint32_t* nodes, uint32_t n_nodes, uint32_t N,
                          uint32_t n, uint32_t wg, uint32_t* table, uint32_t* packed);

/* ---- phi / pi: phi.cc:78-197 (thread), :214-302 (work-group) ---- */
/* mode_wg = 0: PHI_NODE_PER_THREAD (L = phi_wg_size only shapes the grid / stream map);
 * mode_wg = 1: PHI_NODE_PER_WORKGROUP_* with L virtual lanes. seeds advanced in place. */
void orc_update_phi(const orc_params_t* p, const float* beta, const float* pi, const float* phi_sum,
                    const uint64_t* set_slots, uint64_t set_bins, uint32_t set_prime,
                    const uint32_t* nodes, const uint32_t* neighbors, uint32_t n_nodes,
                    uint32_t step_count, orc_seed_t* seeds, uint32_t L, int mode_wg, int noise_on,
                    float* phi_vec);
void orc_update_pi(const orc_params_t* p, float* pi, float* phi_sum, const float* phi_vec,
                   const uint32_t* nodes, uint32_t n_nodes, uint32_t L, int mode_wg);

/* ---- beta / theta: beta.cc:30-233, :334-384 ---- */
void orc_sum_theta(const float* theta, float* theta_sum, uint64_t K);
/* grads_out [2K].  order = 0: reference order (per-group partial rows summed serially,
 * beta.cc:39-49, with the number of partial rows = number of groups actually written);
 * order = 1: float64 accumulation of the same per-edge terms ("mathematical sum"). */
void orc_beta_grads(const orc_params_t* p, const float* theta, const float* theta_sum, const float* beta,
                    const float* pi, const uint64_t* set_slots, uint64_t set_bins, uint32_t set_prime,
                    const uint64_t* edges, uint32_t n_edges, uint32_t L, int mode_wg, int order,
                    float* grads_out);
void orc_update_theta(const orc_params_t* p, float* theta, const float* grads, uint32_t step_count,
                      float scale, orc_seed_t* seeds /* [K] */, int noise_on);
void orc_beta_from_theta(const float* theta, float* beta, uint64_t K); /* beta.cc:376-383 */

/* ---- perplexity: perplexity.cc:14-181, :251-274 ---- */
/* out4: {link_ll, nonlink_ll} as double sums (reduction order unspecified in the reference,
 * perplexity.cc:318-331) and counts; per-edge outputs optional (may be NULL). */
typedef struct { double link_ll, nonlink_ll; uint64_t link_cnt, nonlink_cnt; } orc_ppx_sums_t;
void orc_perplexity(const orc_params_t* p, const float* beta, const float* pi,
                    const uint64_t* set_slots, uint64_t set_bins, uint32_t set_prime,
                    const uint64_t* edges, uint32_t n_edges, uint32_t call_count, uint32_t L, int mode_wg,
                    float* ppx_per_edge, float* edge_ll /* [n_edges] log(ppx) or NULL */,
                    orc_ppx_sums_t* out);
double orc_ppx_value(const orc_ppx_sums_t* s); /* perplexity.cc:264-273 -> -avg ; learner.cc:196-203 exp */

/* ---- host mini-batch samplers: sample.cc:177-303 + learner.cc:162-173 (ammsb_oracle_samplers.c) ---- */
/* iteration order of a libstdc++ std::unordered_set<uint64_t> after inserting keys[0..n) one by one (restated
 * container; test entry).  Returns the number of distinct keys written to out. */
uint64_t orc_uset_order(const uint64_t* keys, uint64_t n, uint64_t* out);
/* strategy: 0 Node, 1 NodeLink, 2 NodeNonLink, 3 BFLink, 4 BFNonLink, 5 BF.  ho_slots may be NULL.  Returns 0, or
 * -1 (bad argument / an output array too small: n_edges, n_nodes then hold the needed sizes). */
int orc_host_sample(uint64_t N, uint64_t E, uint64_t mini_batch, int strategy, unsigned int* seed,
                    const uint64_t* training_edges, uint64_t n_training, const uint64_t* tr_slots, uint64_t tr_bins,
                    uint32_t tr_prime, const uint64_t* ho_slots, uint64_t ho_bins, uint32_t ho_prime,
                    uint64_t* edges_out, uint64_t edges_cap, uint64_t* n_edges, uint32_t* nodes_out,
                    uint64_t nodes_cap, uint64_t* n_nodes, float* weight);

/* ---- the DEVICE mini-batch sampler (csrc/ammsb_minibatch.hip) restated serially: streams, draw, validity,
 * first-occurrence de-duplication, candidate order, padding, weight (ammsb_oracle_samplers.c has the statement) ---- */
void orc_rng_init_mixed(orc_seed_t* seeds, uint64_t n, uint64_t sx, uint64_t sy);
int orc_device_minibatch_nonlink(orc_seed_t* seeds, uint32_t n_candidates, uint32_t u, uint32_t m, uint64_t N,
                                 const uint64_t* tr_slots, uint64_t tr_bins, uint32_t tr_prime,
                                 const uint64_t* ho_slots, uint64_t ho_bins, uint32_t ho_prime, uint64_t* edges_out,
                                 uint32_t* nodes_out, uint32_t* count_out);
int orc_device_minibatch_link(const uint64_t* csr_offsets, const uint32_t* csr_targets, uint32_t u, uint64_t* edges_out,
                              uint32_t* nodes_out, uint32_t* n_out);
float orc_device_minibatch_weight(int link, uint64_t N, uint64_t E, uint32_t m);

/* number of OpenMP threads the library will use (1 if built without OpenMP) */
int orc_num_threads(void);
void orc_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
