// Device-resident step descriptor + the descriptor-taking forms of the per-iteration enqueue functions.
//
// A captured hipGraph cannot take the per-iteration scalars (mini-batch sizes, eps_t, the sampler's vertex u)
// by value: kernel parameters are frozen at capture.  Every hot-path kernel therefore accepts an optional
// pointer to one of these; when it is non-null the kernel reads the sizes / step size from it (one scalar load
// at block start) and the launch grid is sized for the largest mini-batch of that kind, surplus blocks leaving
// at once.  With a null pointer the by-value arguments are used: the eager C-ABI path of include/ammsb.h.
#pragma once

#include <stdint.h>

#include "../../include/ammsb.h"

#define AMMSB_STAMP_CAP 8192u  // steps whose time stamps are kept (ring indexed by step number)
// Stamps of one step (device wall clock, one 8-word record per step): the first block of each kernel of the main chain
// notes when it starts, the hand-over lane of the last kernel when it has released the step and when its poll for the
// next mini-batch is satisfied.  Kernel k's duration (plus one kernel boundary) = stamp k + 1 - stamp k.
#define AMMSB_STAMP_SLOTS 8u
#define AMMSB_STAMP_PHI 0u       // update_phi starts
#define AMMSB_STAMP_PI 1u        // update_pi starts (or the gradient kernel that has update_pi folded in)
#define AMMSB_STAMP_GRADS 2u     // the beta partial-row kernel starts (== AMMSB_STAMP_PI when update_pi is folded in)
#define AMMSB_STAMP_THETA 3u     // the partial-row sum + theta / beta step starts
#define AMMSB_STAMP_RELEASED 4u  // the step's main chain is counted as done (the sampler chain may reuse its buffers)
#define AMMSB_STAMP_NEXT 5u      // the next step's mini-batch is available: the chain's last kernel may end

struct ammsb_step_desc {
  uint32_t n_nodes;  // mini-batch nodes (phi / pi / neighbour sampler)
  uint32_t n_edges;  // mini-batch edges (beta gradient)
  float eps_t;       // get_eps_t(step_count), learner.cc:41-43, evaluated on the host by ammsb_eps_t()
  float scale;       // the sampler's weight (sample.cc:268,292)
  uint32_t u;        // shared end point of the mini-batch (device sampler)
  uint32_t link;     // 1 = link batch (all training edges of u), 0 = non-link batch
  uint32_t n_cand;   // candidate draws of a non-link batch (multiple of 256, <= the sampler's capacity)
  uint32_t step;     // step_count (1-based), index of the optional timestamps
};

// A descriptor with n_nodes == 0 is a SKIP: the step (or the mini-batch) is not to be run.  The loop's device-side
// hand-over hands skips on once one of its waits has given up (ammsb_loop.hip): every kernel of both chains then leaves
// the model state, the stream states and the sample buffers untouched, so that the host finds the state of the last
// COMPLETED step and resumes from it.
#ifdef __HIPCC__
static __device__ __forceinline__ bool ammsb_desc_skip(const ammsb_step_desc* d) { return d && d->n_nodes == 0; }
#endif

// what update_theta does last in a captured step: hand the next two descriptors of the ring to the graph
// that runs next (which has the other buffer parity), and advance the cursor
struct ammsb_step_advance {
  const ammsb_step_desc* ring;  // device ring of upcoming descriptors
  uint32_t* cursor;             // index of the CURRENT step's descriptor in ring
  ammsb_step_desc* cur_out;     // <- ring[cursor + 1]
  ammsb_step_desc* nxt_out;     // <- ring[cursor + nxt_offset]: the batch the next step's sampler chain produces
  uint32_t nxt_offset;
  uint32_t* main_seq;  // completed main chains (the loop's device-side handshake, ammsb_loop.hip); bumped last
  // ... and then the same thread polls until the NEXT step's mini-batch is available (avail - main_seq >= 1), so that
  // the next main chain needs no polling kernel of its own in front; a poll that is not satisfied within max_ticks
  // wall-clock ticks gives up and counts itself in *timeouts
  const uint32_t* avail;
  uint32_t* timeouts;
  unsigned long long max_ticks;
  unsigned long long* stamps;  // optional: AMMSB_STAMP_RELEASED / AMMSB_STAMP_NEXT of this step
};

#ifdef __HIPCC__
// Device wall-clock stamps without extra launches: block 0 of every kernel of the main chain notes when it starts
// (AMMSB_STAMP_*).  The difference of two consecutive stamps is a kernel's duration plus one kernel boundary: a slight
// over-estimate, never an under-estimate.
// (not inlined on purpose: inlined at the top of update_phi_lds_kernel<16, 1> it cost the K = 1024 kernel two spilled
// registers -- 8 bytes of scratch per lane -- for a store one thread of one block makes)
static __device__ __noinline__ void note_stamp_slow(unsigned long long* stamps, const ammsb_step_desc* desc, uint32_t which) {
  if (desc && blockIdx.x == 0 && threadIdx.x == 0)
    stamps[AMMSB_STAMP_SLOTS * (desc->step % AMMSB_STAMP_CAP) + which] = wall_clock64();
}
static __device__ __forceinline__ void note_stamp(unsigned long long* stamps, const ammsb_step_desc* desc, uint32_t which) {
  if (stamps) note_stamp_slow(stamps, desc, which);  // wave-uniform: a call only when time stamps are on
}
#endif

// update_pi folded into the gradient kernel of a captured step (node-stratified mini-batches: edge t = (nodes[0],
// nodes[t + 1])): the arguments update_pi would have taken; all null = not fused
struct ammsb_pi_fusion {
  const float* phi_vec;
  float* phi_sum;
  const uint32_t* nodes;
  unsigned long long* stamps;
};
// whether ammsb_beta_step_d can take the fusion for these work-group sizes (else the caller launches update_pi)
bool ammsb_beta_can_fuse_pi(ammsb_ctx* ctx, uint32_t phi_wg, uint32_t beta_wg);

// ---- descriptor forms (same checks and dispatch as the extern "C" functions; `cap` sizes the grid)
int ammsb_update_phi_d(ammsb_ctx* ctx, const float* beta, const ammsb_rpm* pi, const float* phi_sum,
                       const ammsb_set* training_set, const uint32_t* nodes, const uint32_t* neighbors,
                       uint32_t n_nodes_cap, ammsb_seed* seeds, uint32_t wg, uint32_t flags, float* phi_vec,
                       const ammsb_step_desc* desc, unsigned long long* stamps, void* stream);
int ammsb_update_pi_d(ammsb_ctx* ctx, const ammsb_rpm* pi, float* phi_sum, const float* phi_vec, const uint32_t* nodes,
                      uint32_t n_nodes_cap, uint32_t wg, const ammsb_step_desc* desc, unsigned long long* stamps,
                      void* stream);
int ammsb_beta_grads_d(ammsb_ctx* ctx, const float* theta, const float* beta, const ammsb_rpm* pi,
                       const ammsb_set* training_set, const uint64_t* edges, uint32_t n_edges_cap, uint32_t wg,
                       float* grads_out, const ammsb_step_desc* desc, void* stream);
// (ammsb_beta_step_d trusts the context's table of per-column gradient constants: its own theta step keeps it current,
// and whoever starts a sequence of such steps calls this first)
int ammsb_theta_coef_d(ammsb_ctx* ctx, const float* theta, const float* beta, void* stream);
int ammsb_update_theta_d(ammsb_ctx* ctx, float* theta, float* beta, const float* grads, ammsb_seed* seeds,
                         uint32_t flags, const ammsb_step_desc* desc, const ammsb_step_advance* adv, void* stream);
int ammsb_beta_step_d(ammsb_ctx* ctx, float* theta, float* beta, const ammsb_rpm* pi, const ammsb_set* training_set,
                      const uint64_t* edges, uint32_t n_edges_cap, uint32_t wg, float* grads_out, ammsb_seed* seeds,
                      uint32_t flags, const ammsb_step_desc* desc, const ammsb_step_advance* adv,
                      const ammsb_pi_fusion* fuse, unsigned long long* stamps, void* stream);
int ammsb_sample_neighbors_d(ammsb_ctx* ctx, ammsb_seed* seeds, const uint32_t* nodes, uint32_t n_nodes_cap, uint32_t wg,
                             uint32_t* table, uint32_t* packed, const ammsb_step_desc* desc, void* stream);
int ammsb_minibatch_link_d(ammsb_ctx* ctx, const uint64_t* csr_offsets, const uint32_t* csr_targets, uint32_t n_cap,
                           uint64_t* edges_out, uint32_t* nodes_out, const ammsb_step_desc* desc, void* stream);
int ammsb_minibatch_nonlink_d(ammsb_ctx* ctx, ammsb_seed* seeds, uint32_t n_candidates_cap, uint32_t m,
                              const ammsb_set* training_set, const ammsb_set* heldout_set, void* workspace,
                              uint64_t* edges_out, uint32_t* nodes_out, uint32_t* count_out,
                              const ammsb_step_desc* desc, void* stream);
