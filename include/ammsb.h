/*
 * ammsb.h -- C ABI of the MI355X-native SG-MCMC a-MMSB hot path (libammsb_hip.so).
 *
 * Drop-in boundary for the per-iteration device work of ielhelw/mcmc-ammsb-gpu: each entry
 * point replaces one reference operator (a C++ functor that JIT-builds an OpenCL/CUDA program and
 * launches it through CLCudaAPI).  The reference has no FFI layer of its own; what a binding would
 * attach to is the functor API in mcmc/{phi,beta,perplexity,sample,random,cuckoo}.h, cited per
 * function below (paths relative to the reference checkout).
 *
 * Conventions
 *   - every function returns 0 on success or a negative AMMSB_E* code; nothing aborts, nothing
 *     throws.  ammsb_strerror() names a code; ammsb_last_error(ctx) adds detail.
 *   - device pointers are raw HBM addresses (hipMalloc / torch tensor data_ptr()); `stream` is a
 *     hipStream_t passed as void* (NULL = the null stream).  Calls only enqueue work: no hidden
 *     synchronisation, no allocation after ammsb_ctx_create() (safe inside hipGraph capture).
 *   - plain C types only; structs are POD, passed by pointer, copied before return.
 *   - "wg" arguments are the reference's work-group sizes (Config::phi_wg_size, beta_wg_size,
 *     ppx_wg_size, neighbor_sampler_wg_size).  They fix the RNG-stream <-> (group, lane) mapping and
 *     the WG_SUM summation order exactly as in the reference kernels; the physical launch shape is
 *     the library's own business.  Hot kernels need a power of two in [1, 1024].
 */
#ifndef AMMSB_H
#define AMMSB_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AMMSB_VERSION 200

enum {
  AMMSB_OK = 0,
  AMMSB_EINVAL = -1,   /* bad argument (null pointer, wg not a power of two, K/wg too large, ...) */
  AMMSB_EHIP = -2,     /* a HIP runtime call failed; see ammsb_last_error() */
  AMMSB_ENOMEM = -3,   /* workspace allocation failed */
  AMMSB_ENODEV = -4,   /* no usable gfx950 device */
  AMMSB_ERANGE = -5    /* size exceeds what the launch shape supports */
};

#define AMMSB_MAX_GROUPS 65535u /* mcmc/types.cc:537 GetMaxGroups() */
#define AMMSB_RPM_MAX_BLOCKS 32 /* mcmc/partitioned-alloc.h:15 */

/* one xorshift128+ stream; mcmc/random.h:17-18 random_seed_t (ulong2) */
typedef struct { uint64_t x, y; } ammsb_seed;

/* kernel constants the reference bakes in with -D flags, mcmc/config.cc:66-83 */
typedef struct {
  uint64_t N, K, E;
  uint32_t num_node_sample; /* NUM_NEIGHBORS */
  float alpha, a, b, c, epsilon, eta0, eta1;
} ammsb_params;

/* device view of RowPartitionedMatrix<float>, mcmc/partitioned-alloc.h:14-29 (64-bit offsets) */
typedef struct {
  void* blocks[AMMSB_RPM_MAX_BLOCKS];
  uint64_t rows_in_block;
  uint64_t num_rows, num_cols;
  uint32_t num_blocks;
} ammsb_rpm;

/* device view of the cuckoo edge set, mcmc/cuckoo.cc:17-21; slots = image of Set::Serialize()
 * (cuckoo.cc:211-220): [2 buckets][num_bins][4 slots] u64, empty = UINT64_MAX */
typedef struct {
  const uint64_t* slots;
  uint64_t num_bins;
  uint32_t prime_idx;
} ammsb_set;

/* result of one perplexity pass, device resident; mcmc/perplexity.h:60-63 */
typedef struct {
  double link_ll, nonlink_ll;       /* sums of log(ppx_e) */
  uint64_t link_cnt, nonlink_cnt;
} ammsb_ppx_sums;

typedef struct ammsb_ctx ammsb_ctx;

/* flags for ammsb_update_phi / ammsb_update_theta */
#define AMMSB_NOISE_OFF 1u /* Config::phi_disable_noise: PHI_RANDN(X) := 1, phi.cc:673-677 */
#define AMMSB_PHI_STREAMING 2u /* ammsb_update_phi only: take the throughput (one wave per node) kernels even for a launch
                                * small enough for the latency form (one node per multi-wave block); same results */

int ammsb_version(void);
const char* ammsb_strerror(int code);
const char* ammsb_last_error(const ammsb_ctx* ctx);

/* Apply the "%e" round trip floats take through MakeCompileFlags (config.cc:57-83). */
int ammsb_params_quantize(ammsb_params* p);
/* get_eps_t, learner.cc:41-43 (evaluated on the host and passed to kernels by value) */
float ammsb_eps_t(const ammsb_params* p, uint32_t step_count);

/* One context per device.  Allocates the (small) reduction workspace.  Replaces the per-operator
 * Program/Kernel construction in PhiUpdater / BetaUpdater / PerplexityCalculator ctors. */
int ammsb_ctx_create(int device_id, const ammsb_params* params, ammsb_ctx** out);
int ammsb_ctx_destroy(ammsb_ctx* ctx);
int ammsb_ctx_params(const ammsb_ctx* ctx, ammsb_params* out);

/* RandomInit, random.cc:31-43: seeds[i] = {sx + i, sy + i}.  Replaces OpenClRandomFactory::CreateRandom. */
int ammsb_rng_init(ammsb_ctx* ctx, ammsb_seed* seeds, uint64_t n, uint64_t sx, uint64_t sy, void* stream);
/* Streams for consumers the reference does not have (the device mini-batch sampler): seeds[i] = {mix(sx + 2i),
 * mix(sy + 2i + 1)}, mix = the SplitMix64 finaliser.  The reference layout above gives neighbouring streams nearly
 * identical states whose first xorshift128+ outputs are strongly correlated across streams. */
int ammsb_rng_init_mixed(ammsb_ctx* ctx, ammsb_seed* seeds, uint64_t n, uint64_t sx, uint64_t sy, void* stream);

/* Set_HasEdge over a key list (the `find` kernel of cuckoo-test.cc:45-53). out[i] = 0/1. */
int ammsb_set_has(ammsb_ctx* ctx, const ammsb_set* set, const uint64_t* keys, uint64_t n, uint8_t* out,
                  void* stream);

/* ---- parallel set construction on the device (new, opt-in; the reference inserts on one host thread with a rand_r
 *      random walk, cuckoo.cc:117-161 -- that image is what parity runs use and host/cuckoo.cc reproduces it) ----
 * Builds a VALID table of the same layout and hash pairs from n device-resident keys (distinct, != UINT64_MAX):
 * membership is exactly the key set; WHICH slot holds a key depends on the interleaving, so the image differs from
 * the host build's.  num_bins >= ammsb_set_num_bins(n) = 1 + ceil(1.15 n / 8); slots: [2 * num_bins * 4] u64;
 * scratch: one u32.  Tries the four hash pairs in turn like Set::SetContents (cuckoo.cc:117-129) and reports the
 * one that worked; AMMSB_ERANGE if none did.  Synchronises the stream (it needs the outcome of each attempt). */
uint64_t ammsb_set_num_bins(uint64_t n);
int ammsb_set_build(ammsb_ctx* ctx, const uint64_t* keys, uint64_t n, uint64_t* slots, uint64_t num_bins,
                    uint32_t* prime_idx_out, uint32_t* scratch, void* stream);

/* random::RandomGammaAndNormalize, random.cc:159-167: pi rows ~ Gamma(eta0, eta1) from N*32 streams
 * (caller initialises them with {11,113}), then row-normalise; phi_sum[row] = row sum. */
int ammsb_pi_init_gamma(ammsb_ctx* ctx, const ammsb_rpm* pi, float* phi_sum, float eta0, float eta1,
                        ammsb_seed* seeds, void* stream);

/* NeighborSampler::operator(), sample.cc:111-121 + kernel :48-77.  table: [n_nodes, 2n] scratch
 * (GetHash()), packed: [n_nodes, n] (GetData()).  seeds: >= min(ceil(n_nodes/wg), 65535/wg)*wg streams. */
int ammsb_sample_neighbors(ammsb_ctx* ctx, ammsb_seed* seeds, const uint32_t* nodes, uint32_t n_nodes,
                           uint32_t wg, uint32_t* table, uint32_t* packed, void* stream);

/* PhiUpdater::operator() first half, phi.cc:728-757, kernel update_phi (work-group variants
 * phi.cc:214-302).  Virtual group g (of G = min(n_nodes, 65535)) handles nodes g, g+G, ...; lane l of
 * group g owns stream seeds[g*wg + l].  Only groups in [group_begin, group_end) are processed (the
 * multi-GPU shard; pass 0, UINT32_MAX for all).  phi_vec: [n_nodes, K], row i for nodes[i]. */
int ammsb_update_phi(ammsb_ctx* ctx, const float* beta, const ammsb_rpm* pi, const float* phi_sum,
                     const ammsb_set* training_set, const uint32_t* nodes, const uint32_t* neighbors,
                     uint32_t n_nodes, uint32_t step_count, ammsb_seed* seeds, uint32_t wg, uint32_t flags,
                     uint32_t group_begin, uint32_t group_end, float* phi_vec, void* stream);

/* Diagnostic: blocks of the LDS-streamed update_phi kernel that fit one CU for this context's (K, n) and `wg`
 * (registers + LDS, from the runtime's occupancy calculator), and the waves per block.  0 blocks = the register
 * kernel is used for that shape. */
int ammsb_update_phi_occupancy(ammsb_ctx* ctx, uint32_t wg, int* blocks_per_cu, int* waves_per_block);

/* PhiUpdater::operator() second half, phi.cc:758-762, kernel update_pi (phi.cc:177-197):
 * pi[nodes[i], :] = phi_vec[i, :] / sum, phi_sum[nodes[i]] = sum (WG_SUM order for `wg`). */
int ammsb_update_pi(ammsb_ctx* ctx, const ammsb_rpm* pi, float* phi_sum, const float* phi_vec,
                    const uint32_t* nodes, uint32_t n_nodes, uint32_t wg, void* stream);

/* BetaUpdater::GetThetaSum(), beta.h:27: theta_sum[k] = theta[k,0] + theta[k,1] as the last ammsb_beta_grads of
 * this context computed it (kernel sum_theta, beta.cc:30-37), copied into out[K] (device) on `stream`. */
int ammsb_theta_sum(ammsb_ctx* ctx, float* out, void* stream);

/* BetaUpdater::operator() gradient half, beta.cc:334-366: sum_theta + calculate_grads_partial +
 * sum_grads over edges [edge_begin, edge_end) of the mini-batch.  grads_out: [2K] = the sum over
 * those edges (the mathematical sum; the reference's serial partial-row order is not reproduced, and
 * its stale-row read for > 65535 edges, beta.cc:354-361, is not replicated). */
int ammsb_beta_grads(ammsb_ctx* ctx, const float* theta, const float* beta, const ammsb_rpm* pi,
                     const ammsb_set* training_set, const uint64_t* edges, uint32_t n_edges,
                     uint32_t edge_begin, uint32_t edge_end, uint32_t wg, float* grads_out, void* stream);

/* ammsb_update_pi over nodes[0 .. n_edges] and ammsb_beta_grads over edges [0, n_edges) as ONE launch, for
 * node-stratified mini-batches (edge t = (nodes[0], nodes[t + 1]) in either order, n_nodes = n_edges + 1: what the
 * device sampler's Node strategies produce): every pi row of the mini-batch is normalised from its phi_vec row as the
 * gradient consumes it and written once -- update_pi_kernel's arithmetic, phi.cc:177-197, and the values
 * calculate_grads_partial (beta.cc:145-233) would have read back: bit-identical to the two calls, one launch and one
 * pass over the rows less.  Only for shapes ammsb_can_fuse_pi_beta() accepts (phi wg == beta wg in {32, 64} and
 * K <= 2 wg or K in {256, 512, 1024}); AMMSB_EINVAL otherwise.  The descriptor loop (ammsb_loop) uses the same kernel;
 * the multi-GPU schedule uses this entry so that every rank holds the whole gradient and needs no collective for it. */
int ammsb_can_fuse_pi_beta(ammsb_ctx* ctx, uint32_t phi_wg, uint32_t beta_wg);
int ammsb_update_pi_beta_grads(ammsb_ctx* ctx, const float* theta, const float* beta, const ammsb_rpm* pi,
                               float* phi_sum, const float* phi_vec, const uint32_t* nodes,
                               const ammsb_set* training_set, const uint64_t* edges, uint32_t n_edges, uint32_t wg,
                               float* grads_out, void* stream);

/* Multi-GPU gradient reduction after the all-gather of the R per-rank [cols] vectors into in[R, cols]:
 * out[c] = in[0][c] + in[1][c] + ... in ascending rank order (a fixed association, identical on every rank). */
int ammsb_sum_rows_f32(ammsb_ctx* ctx, const float* in, uint32_t rows, uint32_t cols, float* out, void* stream);

/* BetaUpdater::operator() update half, beta.cc:368-383: update_theta (SGLD, stream k for component k,
 * r0 then r1) followed by beta = pair-normalised theta.  seeds: [K]. */
int ammsb_update_theta(ammsb_ctx* ctx, float* theta, float* beta, const float* grads, uint32_t step_count,
                       float scale, ammsb_seed* seeds, uint32_t flags, void* stream);

/* beta = pair-normalised theta only (random.h:70-79 RandomAndNormalize's device half). */
int ammsb_beta_from_theta(ammsb_ctx* ctx, const float* theta, float* beta, void* stream);

/* PerplexityCalculator::operator(), perplexity.cc:251-274, kernel :159-181 + the four reductions
 * (:318-331).  Edges [edge_begin, edge_end) of the held-out list; ppx_per_edge is the running-mean
 * state (indexed by global edge position); call_count is 1-based.  out: an ammsb_ppx_sums in device memory, or in
 * host-mapped pinned memory (hipHostMalloc) -- the 32 bytes are then on the host once the stream has been
 * synchronised, without a copy (what both hosts of this repository do for the unsharded call). */
int ammsb_perplexity(ammsb_ctx* ctx, const float* beta, const ammsb_rpm* pi, const ammsb_set* heldout_set,
                     const uint64_t* edges, uint32_t n_edges, uint32_t edge_begin, uint32_t edge_end,
                     uint32_t call_count, uint32_t wg, float* ppx_per_edge, ammsb_ppx_sums* out, void* stream);

/* ---- device-side mini-batch sampling (new; the reference samples on one host thread,
 *      sample.cc:177-303 + learner.cc:162-185, which caps throughput once the kernels are fast) ----
 * Strategy "Node" (stratified random node, sample.cc:295-303) split in its two halves; the caller
 * flips the coin and picks u on the host, so that the mini-batch sizes are known without a device
 * round trip.  Same distribution as the host samplers, not the same random stream.
 * SCOPE: the Node family only (Node, NodeLink, NodeNonLink).  The breadth-first strategies (BFLink, BFNonLink, BF,
 * sample.cc:177-247) are a serial frontier walk over a queue -- every pick depends on what the walk has visited -- and
 * stay on the host: mcmc::sampleBreadthFirst* (host/sample.cc, bit-identical to the reference's rand_r stream) feed
 * the same device buffers through Learner::DoSample; none of BASELINE.json's configurations uses them. */

/* sampleNodeLink's edge set for vertex u: all training edges (u, v), v in CSR order.  n = deg(u) =
 * offsets[u+1] - offsets[u] must be > 0.  edges_out[n], nodes_out[n+1] = {u, v_0, ...}. */
int ammsb_minibatch_link(ammsb_ctx* ctx, const uint64_t* csr_offsets, const uint32_t* csr_targets, uint32_t u,
                         uint32_t n, uint64_t* edges_out, uint32_t* nodes_out, void* stream);

/* number of candidate draws (= RNG streams, = workspace entries) ammsb_minibatch_nonlink needs for
 * m distinct non-links out of N vertices when up to `excluded` vertices are invalid partners of u (u itself and
 * its neighbours in the training and held-out graphs): enough draws for m + excluded distinct values plus a
 * margin of 8 % + min(1024, N / 8).  0 if N is too small (needs N >= 2m and the target below 0.95 N).  Size the streams
 * and the workspace for the largest `excluded` of the graph (capacity), and pass the per-vertex value
 * (<= capacity) to each call.  ammsb_minibatch_candidates(N, m) = ..._for(N, m, 0). */
uint32_t ammsb_minibatch_candidates(uint64_t N, uint32_t m);
uint32_t ammsb_minibatch_candidates_for(uint64_t N, uint32_t m, uint64_t excluded);
/* bytes of workspace for a capacity of that many candidates.  The caller fills the workspace with 0xFF bytes ONCE
 * after allocating it; every ammsb_minibatch_nonlink call leaves it in that state again (no memset per call). */
uint64_t ammsb_minibatch_workspace_bytes(uint32_t capacity);

/* sampleNodeNonLink: m distinct v != u with (u,v) in neither set, in candidate order (candidate j =
 * one draw from stream j, j < n_candidates, kept if valid and the first occurrence of its v).  `capacity` is
 * the candidate count seeds / workspace were sized for (n_candidates <= capacity, both multiples of 256).
 * edges_out[m], nodes_out[m+1] = {u, v_0, ...}.  count_out: TWO words -- [0] = number of distinct valid
 * candidates found by this call, [1] += 1 whenever that number is below m (a sticky shortfall counter the caller
 * clears and checks at its synchronisation points: a short mini-batch repeats earlier entries to stay
 * memory-safe and is NOT a valid sample).  With n_candidates from ammsb_minibatch_candidates_for(N, m,
 * 1 + deg_training(u) + deg_heldout(u)) a shortfall needs an astronomically unlucky draw.  heldout_set may be NULL. */
int ammsb_minibatch_nonlink(ammsb_ctx* ctx, ammsb_seed* seeds, uint32_t n_candidates, uint32_t capacity, uint32_t u,
                            uint32_t m, const ammsb_set* training_set, const ammsb_set* heldout_set, void* workspace,
                            uint64_t* edges_out, uint32_t* nodes_out, uint32_t* count_out, void* stream);

/* ---- whole iterations as captured hipGraphs (new; replaces the launch-then-Finish discipline of
 *      learner.cc:237-242, phi.cc:755-761, beta.cc:339-383 where the iteration is launch-latency-bound) ----
 * One ammsb_loop owns 24 small graphs: [update_phi, update_pi, beta gradient, theta/beta step] of step i (main
 * stream) and [mini-batch + neighbour sampling of the mini-batch of step i + 2] (sampler stream), each specialised by
 * (link batch?, buffer set, parity); a learner iteration with device-side mini-batch sampling is one launch of each.
 * Sampling runs two steps ahead through three buffer sets (the caller's two and one the loop allocates); a run starts
 * from the caller's one pending mini-batch and leaves exactly one pending mini-batch in the caller's buffers.  The per-step scalars (sizes, eps_t, weight, u) are read by the kernels from device-resident
 * descriptors the loop uploads a chunk at a time, so a step costs the host two hipGraphLaunch calls and two events.
 * Same kernels, same arguments, same per-stream order as the eager entry points above: the trajectory is
 * bit-identical to calling them one by one.  The caller keeps choosing (link?, u) per mini-batch, as with the
 * eager sampler entry points. */
typedef struct ammsb_loop ammsb_loop;

#define AMMSB_LOOP_TIMESTAMPS 1u /* update_phi / update_pi note the device time at which their first block starts */

typedef struct {
  /* model state (learner.cc:80-91) */
  float* theta;
  float* beta;
  const ammsb_rpm* pi;
  float* phi_sum;
  const ammsb_set* training_set;
  const ammsb_set* heldout_set; /* may be NULL */
  /* PhiUpdater state: streams [min(max_nodes, 65535) * phi_wg], phi_vec [max_nodes, K] (phi.cc:620-629) */
  ammsb_seed* phi_seeds;
  float* phi_vec;
  uint32_t phi_wg, phi_flags;
  /* BetaUpdater state: streams [K], grads [2K] */
  ammsb_seed* beta_seeds;
  float* grads;
  uint32_t beta_wg, beta_flags;
  /* the two Sample buffers (sample.h:51-92): edges [max_edges], nodes [max_nodes], neighbour sampler output
   * [max_nodes, n], table [max_nodes, 2n] and streams, per parity */
  uint64_t* edges[2];
  uint32_t* nodes[2];
  uint32_t* neighbors[2];
  uint32_t* nbr_table[2];
  ammsb_seed* nbr_seeds[2];
  uint32_t nbr_wg;
  /* device mini-batch sampler: training CSR, candidate streams / workspace of capacity mb_candidates, count[2] */
  const uint64_t* csr_offsets;
  const uint32_t* csr_targets;
  ammsb_seed* mb_seeds;
  uint32_t mb_candidates;
  void* mb_workspace;
  uint32_t* mb_count;
  uint32_t mini_batch;  /* m */
  uint32_t max_fan_out; /* largest training degree: link batches are launched for this size */
  uint32_t max_nodes;   /* capacity of nodes[] (rows of neighbors[] / nbr_table[]): max(2m, 1 + max_fan_out), phi.cc:620-622 */
  uint32_t max_edges;   /* capacity of edges[]: max(m, max_fan_out), sample.cc:129 */
  uint32_t flags;       /* AMMSB_LOOP_* */
} ammsb_loop_config;

/* one mini-batch choice of the "Node" strategy family (sample.cc:249-303): link = all n = deg(u) training edges
 * of u (n > 0); non-link = m non-links of u drawn from n_candidates candidate streams */
typedef struct {
  uint32_t link, u, n, n_candidates;
} ammsb_mb_choice;

/* Captures the graphs; every buffer in cfg must stay valid (and keep its address) until ammsb_loop_destroy.
 * edges[] / nodes[] / neighbors[] / nbr_table[] must be 16-byte aligned (AMMSB_EINVAL otherwise): a run that ends with
 * its pending mini-batch in the loop's own third buffer set moves it into the caller's with 16-byte words -- exactly
 * the mini-batch's bytes, nothing beyond them, so the four buffers may be carved back to back from one allocation. */
int ammsb_loop_create(ammsb_ctx* ctx, const ammsb_loop_config* cfg, ammsb_loop** out);
int ammsb_loop_destroy(ammsb_loop* loop);
/* Enqueue n_steps iterations behind the work already queued on `stream` (and make `stream` wait for them).
 * `pending` describes the mini-batch that already sits in buffer pair `parity` (sampled + neighbour-sampled
 * by the eager entry points or by an earlier run); next[i] is the mini-batch sampled DURING step i and consumed
 * by step i + 1, so after the call next[n_steps - 1] is pending in pair parity ^ (n_steps & 1).  Step i uses
 * step_count = first_step_count + i for eps_t.  Does not synchronise. */
int ammsb_loop_run(ammsb_loop* loop, const ammsb_mb_choice* pending, const ammsb_mb_choice* next, uint32_t n_steps,
                   uint32_t first_step_count, uint32_t parity, void* stream);
/* The two chains of an iteration (main: phi, pi, beta; sampler: the mini-batch two steps ahead) are ordered on the
 * device by polling kernels, not by stream events.  A wait that is not satisfied within 5 s GIVES UP instead of
 * hanging the device, and everything behind it is skipped rather than run on a mini-batch that is not there: the
 * model, the RNG streams and the sample buffers stay those of the last completed step.  This call synchronises the
 * loop's streams; if a wait gave up since the last call it switches the loop to the stream-event hand-over for good,
 * re-enqueues the steps that did not run (same kernels, same arguments, same order: the trajectory is the undisturbed
 * one), waits for them and counts a fallback (ammsb_loop_status).  *wait_timeouts stays 0 unless that recovery itself
 * failed; callers treat a non-zero value as an error.  At ammsb_loop_create the loop checks once whether its two
 * streams can overlap at all (a kernel on one spins <= 20 ms for a flag a kernel on the other sets) and takes the
 * event hand-over from the start if they cannot (one hardware queue, tools that run one kernel at a time);
 * AMMSB_LOOP_HANDSHAKE=event|flag in the environment forces either form.  Runs of different loops are submitted one
 * whole run at a time (a process-wide lock), so concurrent learners cannot interleave their polling kernels. */
int ammsb_loop_check(ammsb_loop* loop, uint32_t* wait_timeouts);
/* *event_handover: 1 if the loop orders its chains with stream events (from the start, or since a fallback);
 * *fallbacks: runs that were finished on the event hand-over after a device-side wait gave up.  Either may be NULL. */
int ammsb_loop_status(const ammsb_loop* loop, uint32_t* event_handover, uint32_t* fallbacks);
/* (AMMSB_LOOP_TIMESTAMPS) device time in ns at which update_phi of steps first_step .. first_step + n_steps - 1
 * began, and at which the kernel after it (update_pi) began -- update_phi's duration plus one kernel boundary: a
 * slight over-estimate, never an under-estimate.  The last 8192 steps are kept.  Synchronises. */
int ammsb_loop_timestamps(ammsb_loop* loop, uint32_t first_step, uint32_t n_steps, double* begin_ns, double* end_ns);
/* (AMMSB_LOOP_TIMESTAMPS) every stamp of those steps, out_ns[i * AMMSB_LOOP_STAMP_SLOTS + k] in ns: k = 0 update_phi
 * starts, 1 update_pi starts (or the gradient kernel that has update_pi folded in), 2 the beta partial-row kernel
 * starts, 3 the partial-row sum + theta / beta step starts, 4 the step is released (its buffers may be re-sampled),
 * 5 the next step's mini-batch is available (the chain's last kernel ends).  Consecutive differences are the device
 * times of the categories PrintStats reports (learner.cc:252-299: PHI, PI, GRADS PAR, GRADS SUM + UPDATE THETA), each
 * including one kernel boundary; 5 - 4 is the time the main chain waited for the sampling chain. */
#define AMMSB_LOOP_STAMP_SLOTS 8
int ammsb_loop_step_stamps(ammsb_loop* loop, uint32_t first_step, uint32_t n_steps, double* out_ns);
/* Name of the kernel the last ammsb_update_phi / ammsb_update_pi / ammsb_beta_grads / ammsb_perplexity call on this
 * context dispatched to (which = 0 / 1 / 2 / 3), spelled as the rocprofv3 kernel trace spells it (a substring of the
 * trace's name column); "" before the first call.  update_phi has two slots: launches of at most AMMSB_PHI_WIDE (512)
 * groups -- link mini-batches -- take a several-waves-per-node kernel, recorded under which = 4; which = 0 is the form
 * the large launches take. */
const char* ammsb_last_kernel_name(const ammsb_ctx* ctx, int which);

/* Measurement aid (no reference counterpart): n_blocks one-wave blocks each idle for spin_us microseconds of the
 * 100 MHz device wall clock and write {shader cycles, wall ticks, XCC id} of that interval to out[3 * block ..]
 * (device memory, 24 bytes per block).  cycles / ticks x 100 MHz = the shader clock that XCD held meanwhile -- launched
 * on a second stream beside a kernel under test it reads the clock the chip holds UNDER that load (the chip lowers
 * its clock under load and devices differ: a roofline fraction is read against the clock it was taken at). */
int ammsb_clock_probe(ammsb_ctx* ctx, uint64_t* out, uint32_t n_blocks, uint32_t spin_us, void* stream);

/* ---- wg_* primitives (test entry points; kernels of algorithm/{sum,normalize,sort}.cc) ---- */
/* WG_SUM_KERNEL_TT, sum.cc:44-52: out[r] = WG_SUM(in + r*len, len) with `wg` lanes (any wg in [1,1024]) */
int ammsb_wg_sum_f32(ammsb_ctx* ctx, const float* in, float* out, uint32_t rows, uint32_t len, uint32_t wg,
                     void* stream);
int ammsb_wg_sum_u32(ammsb_ctx* ctx, const uint32_t* in, uint32_t* out, uint32_t rows, uint32_t len,
                     uint32_t wg, void* stream);
/* WG_NORMALIZE_KERNEL_TT, normalize.cc:25-32 (Normalizer<T>, normalize.h:16-55); sums optional */
int ammsb_wg_normalize_f32(ammsb_ctx* ctx, float* inout, float* sums, uint32_t rows, uint32_t len, uint32_t wg,
                           void* stream);
/* WG_SUM_PARTITIONED_KERNEL / WG_NORMALIZE_PARTITIONED_KERNEL, sum.cc:54-65, normalize.cc:34-52 */
int ammsb_rpm_sum_f32(ammsb_ctx* ctx, const ammsb_rpm* m, float* out, uint32_t wg, void* stream);
int ammsb_rpm_normalize_f32(ammsb_ctx* ctx, const ammsb_rpm* m, float* sums, uint32_t wg, void* stream);
/* WG_SORT_TT, sort.cc:11-32: one group sorts len = wg elements (power of two <= 1024) */
int ammsb_wg_sort_u32(ammsb_ctx* ctx, const uint32_t* in, uint32_t* out, uint32_t len, void* stream);
int ammsb_wg_sort_f32(ammsb_ctx* ctx, const float* in, float* out, uint32_t len, void* stream);
/* the `generate` kernel of random-test.cc:33-45: thread t writes `per_stream` normals from stream t */
int ammsb_randn_fill(ammsb_ctx* ctx, ammsb_seed* seeds, uint32_t n_streams, uint32_t per_stream, float* out,
                     void* stream);
/* the `fetch` kernel of test-partitioned-alloc.cc:53-62: out[0..1] = row[col], row[col+1] (as u32 bits) */
int ammsb_rpm_fetch(ammsb_ctx* ctx, const ammsb_rpm* m, uint64_t row, uint64_t col, uint32_t* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AMMSB_H */
