// ammsb_loop: whole learner iterations replayed as captured hipGraphs.
//
// The reference's loop (mcmc/learner.cc:222-247) launches one kernel at a time and waits for each
// (queue.Finish() in phi.cc:755-761, beta.cc:339-383): seven launch + wait pairs per iteration, which is what
// an iteration costs at the small configurations (N = 10k..100k: a few tens of microseconds of device work).
// Here one iteration is two hipGraphLaunch calls on two streams: the MAIN chain of step i (update_phi, update_pi,
// the beta partial rows, one kernel for their sum and the theta/beta step) and the SAMPLER chain that produces the
// mini-batch of step i + 2 (mini-batch kernels + neighbour sampler).  Sampling runs TWO steps ahead through three
// buffer sets (the caller's two Sample buffers and one of the loop's own), so a short step -- a link batch is a few
// dozen edges -- does not wait for the ~60 us sampling chain of a 65536-edge batch.  The ring is ordered on the
// device (sampler(i) waits for main(i-1), main(i) for sampler(i-2)): two counters in device memory -- main chains
// completed, mini-batches available -- a one-lane wait kernel at the head of the sampler chain, a poll in the last
// kernel of the main chain (for the NEXT step's mini-batch) and a bump at the end of each (a
// cross-stream event wait costs ~14-20 us of device time per step on this runtime even when already satisfied;
// the wait kernel is one more graph node, ~1.6 us).  AMMSB_LOOP_HANDSHAKE=event orders the chains with stream
// events instead (no polling kernels in the graphs): for profilers that run one kernel at a time, under which a
// polling kernel would starve the chain it waits for (chosen automatically under rocprofv3 --pmc;
// AMMSB_LOOP_HANDSHAKE=flag insists on the polls).  A run starts with the caller's one pending
// mini-batch and ends with exactly one pending mini-batch in the caller's buffer, as the eager loop does, so the
// two forms can alternate and a checkpoint needs nothing new.  (A fork/join inside ONE graph costs ~40 us per
// replay on this runtime, an in-line chain exposes the sampler's latency, one step of look-ahead leaves every
// link step waiting for the next non-link batch's sampling: all three were measured)
//
// What changes from step to step (mini-batch sizes, eps_t, the sampler's vertex u) cannot be a kernel
// parameter of a captured graph.  It lives in device memory instead (ammsb_step_desc, ammsb_step.h): the host
// uploads the descriptors of the next <= CHUNK steps in one copy, and the last kernel of every step hands the
// following descriptors to the graphs that run next.  Graphs are specialised by (link batch?, physical buffer,
// parity): 24 small linear graphs, captured once.  A link batch has deg(u) edges, so its kernels are launched for
// the largest degree and read the real size from the descriptor; surplus blocks leave at once.
//
// Every kernel is the one the eager C ABI launches, with the same arguments in the same per-stream order, so
// the trajectory is bit-identical to the eager loop's (tests/test_gpu_graph_loop.py).
#include "ammsb_ctx.h"
#include "ammsb_step.h"

#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

namespace {

constexpr uint32_t CHUNK = 1024;                 // steps per descriptor upload
constexpr uint32_t STAMP_CAP = AMMSB_STAMP_CAP;  // steps whose update_phi timestamps are kept
constexpr int NBUF = 3;                          // sample buffer sets: the caller's two + one of the loop's own

// (timeouts: non-null with the device-side hand-over; once a wait has given up, the steps of later chunks and runs are
// skipped like the rest -- the prime kernel must not hand them a live descriptor)
__global__ void loop_prime_kernel(const ammsb_step_desc* ring, uint32_t* cursor, ammsb_step_desc* cur,
                                  ammsb_step_desc* nxt, const uint32_t* timeouts) {
  *cur = ring[0];  // the step that runs next
  *nxt = ring[2];  // the mini-batch its sampler chain produces (two steps ahead)
  *cursor = 0;
  if (timeouts && *timeouts != 0u) {
    cur->n_nodes = 0;
    cur->n_edges = 0;
  }
}

// ---- device-side hand-shake.  hs[0] = main chains completed, hs[1] = mini-batches available (both count over the
// loop's lifetime; at every ammsb_loop_run boundary hs[1] == hs[0] + 1: the one pending mini-batch), hs[2] = waits
// that gave up (sticky; ammsb_loop_check).  A wait holds while (int)(*a - *b) < min_diff; `b` is only written by the
// waiting stream itself.
constexpr int HS_MAIN = 0, HS_AVAIL = 1, HS_TIMEOUTS = 2;

// A wait that is not satisfied within max_ticks GIVES UP: it counts itself in *timeouts, and from then on the run is
// "poisoned" -- every later wait gives up at once, and what a given-up wait guards is skipped instead of run on data
// that is not there (skip descriptors, ammsb_step.h), so the state stays that of the last completed step and the host
// can resume from it (ammsb_loop_check).  `src` / `dst` (sampler chain): the chain's kernels read their descriptor from
// the private copy `dst`, which this kernel fills -- with *src when the wait is satisfied, with a skip when it gave up.
// fail_at (test hook, AMMSB_LOOP_TEST_FAIL_AT): the wait whose own counter *b equals fail_at - 1 gives up at once.
__global__ void loop_wait_kernel(const uint32_t* a, const uint32_t* b, int min_diff, uint32_t* timeouts,
                                 unsigned long long max_ticks, const ammsb_step_desc* src, ammsb_step_desc* dst,
                                 uint32_t fail_at) {
  const uint32_t have = *b;
  const unsigned long long t0 = wall_clock64();
  bool ok = false;
  for (;;) {
    if (__hip_atomic_load(timeouts, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;  // poisoned
    if (fail_at != 0u && have + 1u == fail_at) {
      atomicAdd(timeouts, 1u);
      break;
    }
    const uint32_t v = __hip_atomic_load(a, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    if ((int)(v - have) >= min_diff) {
      ok = true;
      break;
    }
    if (wall_clock64() - t0 > max_ticks) {  // a bug or a kernel-serialising tool: never hang the device
      atomicAdd(timeouts, 1u);
      break;
    }
    __builtin_amdgcn_s_sleep(8);
  }
  if (dst) {
    if (ok) {
      *dst = *src;
    } else {
      ammsb_step_desc skip = *src;
      skip.n_nodes = 0;
      skip.n_edges = 0;
      skip.n_cand = 0;
      *dst = skip;
    }
  }
}

// end of a sampler chain: one more mini-batch available (not when the chain was skipped)
__global__ void loop_bump_kernel(uint32_t* counter, const ammsb_step_desc* desc) {
  if (ammsb_desc_skip(desc)) return;
  __hip_atomic_store(counter, *counter + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// the run's epilogue copy of the pending mini-batch into the caller's buffer set -- unless the run is poisoned (the
// source may then not have been sampled, and the destination may hold a mini-batch the resumed run still needs)
// Exactly `bytes` bytes (a multiple of 4: the buffers hold u32 / u64 items): whole 16-byte chunks, then the tail word by
// word -- nothing past the declared size is written, so a caller may carve its buffers back to back from one allocation.
// dst / src are 16-byte aligned (checked by ammsb_loop_create; the loop's own set is 256-byte aligned).
__global__ void loop_copy_kernel(uint4* dst, const uint4* src, size_t bytes, const uint32_t* timeouts) {
  if (timeouts && *timeouts != 0u) return;
  const size_t n16 = bytes / 16;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = src[i];
  if (blockIdx.x == 0 && threadIdx.x < (bytes % 16) / 4)
    reinterpret_cast<uint32_t*>(dst + n16)[threadIdx.x] = reinterpret_cast<const uint32_t*>(src + n16)[threadIdx.x];
}

// create-time probe: can a kernel on one stream run while a kernel on the other is spinning?
__global__ void loop_probe_wait_kernel(const uint32_t* flag, uint32_t* saw, unsigned long long max_ticks) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 <= max_ticks) {
    if (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
      *saw = 1u;
      return;
    }
    __builtin_amdgcn_s_sleep(8);
  }
  *saw = 0u;
}
__global__ void loop_probe_set_kernel(uint32_t* flag) { __hip_atomic_store(flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }

struct Stage {
  ammsb_step_desc* ring;  // pinned host staging, [CHUNK + 4]
  hipEvent_t done;        // the upload that read it has executed
  bool used;
};

struct SampleBuf {  // one mini-batch: edges, nodes, the neighbour sampler's packed result and table image
  uint64_t* edges;
  uint32_t* nodes;
  uint32_t* neighbors;
  uint32_t* nbr_table;
};

}  // namespace

struct ammsb_loop {
  ammsb_ctx* ctx;
  ammsb_loop_config c;
  ammsb_rpm pi;
  ammsb_set training, heldout;
  bool has_heldout;
  hipStream_t main, side;
  hipEvent_t ev_in, ev_out, ev_prime, ev_first;
  hipEvent_t ev_main[NBUF], ev_samp[NBUF];  // step i's main chain / sampler chain finished (i % 3)
  SampleBuf buf[NBUF];                      // [0], [1]: the caller's Sample buffers; [2]: owned
  void* own_mem;                            // the allocation behind buf[2]
  ammsb_step_desc* d_ring;                  // [CHUNK + 4]
  uint32_t* d_cursor;
  ammsb_step_desc* d_cur[2];     // descriptor of the step that runs next, by step parity
  ammsb_step_desc* d_nxt[NBUF];  // descriptor of the mini-batch being sampled into buffer set b (a sampler chain may
                                 // still be reading it two main chains later: one per buffer set, not per parity)
  uint32_t* d_hs;                // hand-shake counters (HS_*)
  ammsb_step_desc* d_samp[NBUF]; // the sampler chain's private copy of its descriptor (filled by its wait kernel)
  bool use_events;               // the chains are ordered with stream events (no polling kernels)
  bool direct_only;              // the captured graphs are stale (they poll; the loop fell back to events): launch directly
  uint32_t fallbacks;            // runs that were finished on the event hand-over after a device-side wait gave up
  uint32_t test_fail_at;         // AMMSB_LOOP_TEST_FAIL_AT: the sampler chain of this lifetime mini-batch fails its wait
  struct Run {                   // what was enqueued since the last ammsb_loop_check (to resume from, if need be)
    ammsb_mb_choice pending;
    std::vector<ammsb_mb_choice> next;
    uint32_t n_steps, first_step_count, parity;
    hipStream_t stream;
    uint64_t main_start;         // lifetime main chains before this run
  };
  std::vector<Run>* runs;
  uint64_t main_total;           // lifetime main chains enqueued
  unsigned long long wait_ticks; // a wait gives up after this many wall-clock ticks
  unsigned long long* d_stamps;  // [STAMP_CAP][AMMSB_STAMP_SLOTS] or null
  Stage stage[2];
  int next_stage;
  hipGraphExec_t exec_main[2][NBUF][2];  // [this step link][physical buffer read][descriptor parity]
  hipGraphExec_t exec_samp[2][NBUF][2];  // [sampled batch link][physical buffer written][descriptor parity]
  uint32_t link_nodes_cap, link_edges_cap, n_nbr;
  float w_link, w_nonlink;
  int wall_khz;
  uint64_t graphs_launched;
  bool serial_launch;  // AMMSB_LOOP_LAUNCH=serial: both chains' graphs from the calling thread
  bool eager_launch;   // default: the chains' kernels launched one by one from two threads; AMMSB_LOOP_LAUNCH=graph
                       // replays the captured graphs from two threads instead
  bool host_prof;  // AMMSB_LOOP_HOSTPROF=1: time the two graph launches of a step on the host, print at destroy
  double t_side_us, t_main_us, t_run_us, t_pro_us, t_epi_us;
  uint64_t runs_done;
};

namespace {

#define LOOP_HIP(lp, call)                                                                              \
  do {                                                                                                  \
    hipError_t e_ = (call);                                                                             \
    if (e_ != hipSuccess) {                                                                             \
      snprintf((lp)->ctx->err, sizeof((lp)->ctx->err), "%s: %s -> %s", __func__, #call, hipGetErrorString(e_)); \
      return AMMSB_EHIP;                                                                                \
    }                                                                                                   \
  } while (0)

#define LOOP_RC(call)         \
  do {                        \
    const int rc_ = (call);   \
    if (rc_ != AMMSB_OK) return rc_; \
  } while (0)

// the sampling chain of one mini-batch (sample.cc:249-303 + learner.cc:162-194) into buffer set `b`, with the
// neighbour-sampler streams of Sample[sp]; the sizes / vertex come from `desc`
int record_sampler(ammsb_loop* lp, int nl, int b, int sp, const ammsb_step_desc* desc, hipStream_t st) {
  const ammsb_loop_config& c = lp->c;
  ammsb_ctx* ctx = lp->ctx;
  const SampleBuf& o = lp->buf[b];
  const uint32_t m = c.mini_batch;
  // mini-batch J = hs[AVAIL] goes into the buffer set step J - 3 read: wait for main chains 0 .. J - 3.  The chain's
  // kernels then read the wait kernel's private copy of the descriptor (a skip if the wait gave up).
  if (!lp->use_events) {
    loop_wait_kernel<<<1, 1, 0, st>>>(lp->d_hs + HS_MAIN, lp->d_hs + HS_AVAIL, -2, lp->d_hs + HS_TIMEOUTS, lp->wait_ticks,
                                      desc, lp->d_samp[b], lp->test_fail_at);
    LOOP_HIP(lp, hipGetLastError());
    desc = lp->d_samp[b];
  }
  if (nl)
    LOOP_RC(ammsb_minibatch_link_d(ctx, c.csr_offsets, c.csr_targets, lp->link_edges_cap, o.edges, o.nodes, desc, st));
  else
    LOOP_RC(ammsb_minibatch_nonlink_d(ctx, c.mb_seeds, c.mb_candidates, m, &lp->training,
                                      lp->has_heldout ? &lp->heldout : nullptr, c.mb_workspace, o.edges, o.nodes,
                                      c.mb_count, desc, st));
  LOOP_RC(ammsb_sample_neighbors_d(ctx, c.nbr_seeds[sp], o.nodes, nl ? lp->link_nodes_cap : m + 1, c.nbr_wg, o.nbr_table,
                                   o.neighbors, desc, st));
  if (!lp->use_events) {
    loop_bump_kernel<<<1, 1, 0, st>>>(lp->d_hs + HS_AVAIL, desc);
    LOOP_HIP(lp, hipGetLastError());
  }
  return AMMSB_OK;
}

// one step's chain (learner.cc:237-242) over buffer set `b`, descriptor parity dp
int record_main(ammsb_loop* lp, int cl, int b, int dp, hipStream_t st) {
  const ammsb_loop_config& c = lp->c;
  ammsb_ctx* ctx = lp->ctx;
  const SampleBuf& in = lp->buf[b];
  const ammsb_step_desc* cur = lp->d_cur[dp];
  const uint32_t m = c.mini_batch;
  const uint32_t cap_nodes = cl ? lp->link_nodes_cap : m + 1;
  const uint32_t cap_edges = cl ? lp->link_edges_cap : m;
  // step S = hs[MAIN] consumes mini-batch S, which must be available: the LAST kernel of step S - 1 (below) does not
  // finish before it is -- no polling kernel in front of this chain (saves a launch and a kernel boundary per step).
  // The first step of a run needs no poll: at every run boundary the pending mini-batch is available
  // (hs[AVAIL] == hs[MAIN] + 1, the invariant the epilogue of ammsb_loop_run re-establishes).
  // (AMMSB_LOOP_TIMESTAMPS) block 0 of update_phi and block 0 of update_pi note the device time they start at
  LOOP_RC(ammsb_update_phi_d(ctx, c.beta, &lp->pi, c.phi_sum, &lp->training, in.nodes, in.neighbors, cap_nodes,
                             c.phi_seeds, c.phi_wg, c.phi_flags, c.phi_vec, cur, lp->d_stamps, st));
  // update_pi: its own launch, or folded into the gradient kernel (every mini-batch of this loop is node-stratified:
  // edge t joins nodes[0] and nodes[t + 1])
  const bool fuse_pi = ammsb_beta_can_fuse_pi(ctx, c.phi_wg, c.beta_wg) && lp->pi.num_cols % 4 == 0;
  const ammsb_pi_fusion fuse = {c.phi_vec, c.phi_sum, in.nodes, lp->d_stamps};
  if (!fuse_pi)
    LOOP_RC(ammsb_update_pi_d(ctx, &lp->pi, c.phi_sum, c.phi_vec, in.nodes, cap_nodes, c.phi_wg, cur, lp->d_stamps, st));
  // the last kernel hands ring[c + 1] (next step) and ring[c + 3] over: the batch the next step's sampler chain
  // produces -- into the buffer set this step has just finished reading
  const bool hs = !lp->use_events;
  const ammsb_step_advance adv = {lp->d_ring, lp->d_cursor, lp->d_cur[1 - dp], lp->d_nxt[b], 3u,
                                  hs ? lp->d_hs + HS_MAIN : nullptr, hs ? lp->d_hs + HS_AVAIL : nullptr,
                                  hs ? lp->d_hs + HS_TIMEOUTS : nullptr, lp->wait_ticks, lp->d_stamps};
  LOOP_RC(ammsb_beta_step_d(ctx, c.theta, c.beta, &lp->pi, &lp->training, in.edges, cap_edges, c.beta_wg, c.grads,
                            c.beta_seeds, c.beta_flags, cur, &adv, fuse_pi ? &fuse : nullptr, lp->d_stamps, st));
  return AMMSB_OK;
}

// kind 1: main chain, 2: sampler chain (reading d_nxt[b]; its neighbour-sampler streams are Sample[dp]'s: the batch
// sampled during step i is consumed by step i + 2, which has the same parity)
int capture(ammsb_loop* lp, int kind, int link, int b, int dp, hipGraphExec_t* out) {
  hipStream_t st = kind == 2 ? lp->side : lp->main;
  LOOP_HIP(lp, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  const int rc = kind == 2 ? record_sampler(lp, link, b, dp, lp->d_nxt[b], st) : record_main(lp, link, b, dp, st);
  hipGraph_t g = nullptr;
  const hipError_t e = hipStreamEndCapture(st, &g);
  if (rc != AMMSB_OK) {
    if (g) (void)hipGraphDestroy(g);
    return rc;
  }
  if (e != hipSuccess || !g) {
    snprintf(lp->ctx->err, sizeof lp->ctx->err, "ammsb_loop: hipStreamEndCapture -> %s", hipGetErrorString(e));
    return AMMSB_EHIP;
  }
  const hipError_t e2 = hipGraphInstantiate(out, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e2 != hipSuccess) {
    snprintf(lp->ctx->err, sizeof lp->ctx->err, "ammsb_loop: hipGraphInstantiate -> %s", hipGetErrorString(e2));
    return AMMSB_EHIP;
  }
  return AMMSB_OK;
}

void destroy(ammsb_loop* lp) {
  if (!lp) return;
  if (lp->host_prof && lp->graphs_launched)
    fprintf(stderr,
            "ammsb_loop host profile: %llu steps, hipGraphLaunch sampler %.2f us, main %.2f us per step; %llu runs: %.1f us "
            "each (prologue %.1f, epilogue %.1f)\n",
            (unsigned long long)lp->graphs_launched, lp->t_side_us / lp->graphs_launched, lp->t_main_us / lp->graphs_launched,
            (unsigned long long)lp->runs_done, lp->t_run_us / lp->runs_done, lp->t_pro_us / lp->runs_done, lp->t_epi_us / lp->runs_done);
  if (lp->main) (void)hipStreamSynchronize(lp->main);
  if (lp->side) (void)hipStreamSynchronize(lp->side);
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < NBUF; ++b)
      for (int p = 0; p < 2; ++p) {
        if (lp->exec_main[a][b][p]) (void)hipGraphExecDestroy(lp->exec_main[a][b][p]);
        if (lp->exec_samp[a][b][p]) (void)hipGraphExecDestroy(lp->exec_samp[a][b][p]);
      }
  for (Stage& st : lp->stage) {
    if (st.ring) (void)hipHostFree(st.ring);
    if (st.done) (void)hipEventDestroy(st.done);
  }
  for (hipEvent_t ev : {lp->ev_in, lp->ev_out, lp->ev_prime, lp->ev_first, lp->ev_main[0], lp->ev_main[1], lp->ev_main[2],
                        lp->ev_samp[0], lp->ev_samp[1], lp->ev_samp[2]})
    if (ev) (void)hipEventDestroy(ev);
  if (lp->d_ring) (void)hipFree(lp->d_ring);  // one allocation: ring, cursor, cur/nxt
  if (lp->d_stamps) (void)hipFree(lp->d_stamps);
  if (lp->d_hs) (void)hipFree(lp->d_hs);
  if (lp->own_mem) (void)hipFree(lp->own_mem);
  if (lp->side) (void)hipStreamDestroy(lp->side);
  if (lp->main) (void)hipStreamDestroy(lp->main);
  delete lp->runs;
  delete lp;
}

// descriptor of the step that consumes mini-batch `ch` as step number `step`
ammsb_step_desc make_desc(const ammsb_loop* lp, const ammsb_mb_choice& ch, uint32_t step) {
  ammsb_step_desc d;
  memset(&d, 0, sizeof d);
  if (ch.link) {
    d.n_nodes = ch.n + 1;
    d.n_edges = ch.n;
    d.scale = lp->w_link;
  } else {
    d.n_nodes = lp->c.mini_batch + 1;
    d.n_edges = lp->c.mini_batch;
    d.scale = lp->w_nonlink;
  }
  d.eps_t = ammsb_eps_t(&lp->ctx->params, step);
  d.u = ch.u;
  d.link = ch.link ? 1u : 0u;
  d.n_cand = ch.n_candidates;
  d.step = step;
  return d;
}

int check_choice(ammsb_loop* lp, const ammsb_mb_choice& ch) {
  ammsb_ctx* ctx = lp->ctx;
  AMMSB_CHECK_ARG(ctx, ch.u < ctx->params.N, "mini-batch vertex out of range");
  if (ch.link) {
    AMMSB_CHECK_ARG(ctx, ch.n > 0 && ch.n <= lp->link_edges_cap, "link batch: degree 0 or above max_fan_out");
  } else {
    AMMSB_CHECK_ARG(ctx, ch.n_candidates >= lp->c.mini_batch && ch.n_candidates <= lp->c.mb_candidates &&
                             ch.n_candidates % 256 == 0,
                    "non-link batch: bad candidate count");
  }
  return AMMSB_OK;
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

extern "C" int ammsb_loop_create(ammsb_ctx* ctx, const ammsb_loop_config* cfg, ammsb_loop** out) {
  AMMSB_CHECK_ARG(ctx, ctx && cfg && out, "null argument");
  const ammsb_loop_config& c = *cfg;
  AMMSB_CHECK_ARG(ctx, c.theta && c.beta && c.pi && c.phi_sum && c.training_set && c.phi_seeds && c.phi_vec &&
                           c.beta_seeds && c.grads,
                  "null model / operator buffer");
  for (int i = 0; i < 2; ++i)
    AMMSB_CHECK_ARG(ctx, c.edges[i] && c.nodes[i] && c.neighbors[i] && c.nbr_table[i] && c.nbr_seeds[i],
                    "null sample buffer");
  for (int i = 0; i < 2; ++i)  // the hand-back of the pending mini-batch moves 16-byte words (loop_copy_kernel)
    AMMSB_CHECK_ARG(ctx, ((reinterpret_cast<uintptr_t>(c.edges[i]) | reinterpret_cast<uintptr_t>(c.nodes[i]) |
                           reinterpret_cast<uintptr_t>(c.neighbors[i]) | reinterpret_cast<uintptr_t>(c.nbr_table[i])) & 15u) == 0,
                    "edges / nodes / neighbors / nbr_table must be 16-byte aligned");
  AMMSB_CHECK_ARG(ctx, c.csr_offsets && c.csr_targets && c.mb_seeds && c.mb_workspace && c.mb_count,
                  "null mini-batch sampler buffer");
  AMMSB_CHECK_ARG(ctx, c.mini_batch > 0 && c.max_fan_out > 0, "mini_batch / max_fan_out must be positive");
  AMMSB_CHECK_ARG(ctx, c.mb_candidates >= c.mini_batch && c.mb_candidates % 256 == 0, "bad candidate capacity");
  AMMSB_CHECK_ARG(ctx, c.max_nodes >= c.mini_batch + 1 && c.max_nodes >= c.max_fan_out + 1 && c.max_edges >= c.mini_batch &&
                           c.max_edges >= c.max_fan_out,
                  "max_nodes / max_edges smaller than the largest mini-batch (sample.cc:129-131)");
  AMMSB_HIP(ctx, hipSetDevice(ctx->device));
  ammsb_loop* lp = new (std::nothrow) ammsb_loop();
  if (!lp) return AMMSB_ENOMEM;
  memset(static_cast<void*>(lp), 0, sizeof *lp);
  lp->ctx = ctx;
  lp->c = c;
  lp->pi = *c.pi;
  lp->training = *c.training_set;
  lp->has_heldout = c.heldout_set != nullptr;
  if (lp->has_heldout) lp->heldout = *c.heldout_set;
  lp->c.pi = &lp->pi;  // the copies outlive the caller's descriptors
  lp->c.training_set = &lp->training;
  lp->c.heldout_set = lp->has_heldout ? &lp->heldout : nullptr;
  lp->link_edges_cap = c.max_fan_out;
  lp->link_nodes_cap = c.max_fan_out + 1;
  lp->n_nbr = ctx->params.num_node_sample;
  lp->w_link = static_cast<float>(ctx->params.N);                                                    // sample.cc:268
  lp->w_nonlink = static_cast<float>(2 * ctx->params.E) / static_cast<float>(c.mini_batch);          // sample.cc:292
#define CREATE_HIP(call)                                                                           \
  do {                                                                                             \
    hipError_t e_ = (call);                                                                        \
    if (e_ != hipSuccess) {                                                                        \
      snprintf(ctx->err, sizeof ctx->err, "ammsb_loop_create: %s -> %s", #call, hipGetErrorString(e_)); \
      destroy(lp);                                                                                 \
      return AMMSB_EHIP;                                                                           \
    }                                                                                              \
  } while (0)
  {
    // AMMSB_LOOP_PRIO=1 (A/B runs): the main chain on the highest stream priority, the sampling chain on the lowest --
    // the sampling kernels of mini-batch i + 2 run beside update_phi of step i and cost it ~6 % (1.51 ms alone against
    // 1.60 in the loop at C3); they have two steps' time to finish in
    static const int prio = [] {
      const char* f = getenv("AMMSB_LOOP_PRIO");
      return f ? atoi(f) : 0;
    }();
    int least = 0, greatest = 0;
    if (prio) CREATE_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    // AMMSB_LOOP_SIDE_CUS=n (A/B runs): the sampling chain confined to n of the device's CUs (a CU mask on its stream)
    static const int side_cus = [] {
      const char* f = getenv("AMMSB_LOOP_SIDE_CUS");
      return f ? atoi(f) : 0;
    }();
    if (side_cus > 0 && side_cus < ctx->num_cus) {
      uint32_t mask[16] = {0};
      for (int i = 0; i < side_cus && i < 512; ++i) mask[i / 32] |= 1u << (i % 32);
      CREATE_HIP(hipStreamCreateWithFlags(&lp->main, hipStreamNonBlocking));
      CREATE_HIP(hipExtStreamCreateWithCUMask(&lp->side, (uint32_t)((ctx->num_cus + 31) / 32), mask));
    } else if (prio && least != greatest) {
      CREATE_HIP(hipStreamCreateWithPriority(&lp->main, hipStreamNonBlocking, greatest));
      CREATE_HIP(hipStreamCreateWithPriority(&lp->side, hipStreamNonBlocking, least));
    } else {
      CREATE_HIP(hipStreamCreateWithFlags(&lp->main, hipStreamNonBlocking));
      CREATE_HIP(hipStreamCreateWithFlags(&lp->side, hipStreamNonBlocking));
    }
  }
  for (hipEvent_t* ev : {&lp->ev_in, &lp->ev_out, &lp->ev_prime, &lp->ev_first, &lp->ev_main[0], &lp->ev_main[1],
                         &lp->ev_main[2], &lp->ev_samp[0], &lp->ev_samp[1], &lp->ev_samp[2]})
    CREATE_HIP(hipEventCreateWithFlags(ev, hipEventDisableTiming));
  // the caller's two Sample buffers and a third set of the same sizes (zeroed: whatever a kernel reads is a valid id)
  for (int i = 0; i < 2; ++i) lp->buf[i] = SampleBuf{c.edges[i], c.nodes[i], c.neighbors[i], c.nbr_table[i]};
  {
    const size_t e_b = align256(sizeof(uint64_t) * c.max_edges), n_b = align256(sizeof(uint32_t) * c.max_nodes);
    const size_t p_b = align256(sizeof(uint32_t) * (size_t)c.max_nodes * lp->n_nbr);
    const size_t t_b = align256(sizeof(uint32_t) * (size_t)c.max_nodes * 2 * lp->n_nbr);
    CREATE_HIP(hipMalloc(&lp->own_mem, e_b + n_b + p_b + t_b));
    CREATE_HIP(hipMemset(lp->own_mem, 0, e_b + n_b + p_b + t_b));
    char* p = static_cast<char*>(lp->own_mem);
    lp->buf[2].edges = reinterpret_cast<uint64_t*>(p);
    lp->buf[2].nodes = reinterpret_cast<uint32_t*>(p + e_b);
    lp->buf[2].neighbors = reinterpret_cast<uint32_t*>(p + e_b + n_b);
    lp->buf[2].nbr_table = reinterpret_cast<uint32_t*>(p + e_b + n_b + p_b);
  }
  lp->runs = new (std::nothrow) std::vector<ammsb_loop::Run>();
  if (!lp->runs) {
    destroy(lp);
    return AMMSB_ENOMEM;
  }
  // one device allocation: ring [CHUNK + 4], cur[2], nxt[3], samp[3], cursor
  const size_t n_desc = CHUNK + 4 + 8;
  CREATE_HIP(hipMalloc(&lp->d_ring, sizeof(ammsb_step_desc) * n_desc + 64));
  CREATE_HIP(hipMemset(lp->d_ring, 0, sizeof(ammsb_step_desc) * n_desc + 64));
  lp->d_cur[0] = lp->d_ring + CHUNK + 4;
  lp->d_cur[1] = lp->d_ring + CHUNK + 5;
  lp->d_nxt[0] = lp->d_ring + CHUNK + 6;
  lp->d_nxt[1] = lp->d_ring + CHUNK + 7;
  lp->d_nxt[2] = lp->d_ring + CHUNK + 8;
  for (int b = 0; b < NBUF; ++b) lp->d_samp[b] = lp->d_ring + CHUNK + 9 + b;
  lp->d_cursor = reinterpret_cast<uint32_t*>(lp->d_ring + n_desc);
  {
    CREATE_HIP(hipMalloc(&lp->d_hs, 64));
    const uint32_t hs0[4] = {0u, 1u, 0u, 0u};  // nothing run yet, the caller's pending mini-batch available
    CREATE_HIP(hipMemcpy(lp->d_hs, hs0, sizeof hs0, hipMemcpyHostToDevice));
    const char* mode = getenv("AMMSB_LOOP_HANDSHAKE");
    // rocprofv3 --pmc (its launcher exports ROCPROF_COUNTER_COLLECTION) runs one kernel at a time: a polling kernel
    // would starve the chain it waits for, so the event hand-over is the default there
    const char* pmc = getenv("ROCPROF_COUNTER_COLLECTION");
    auto truthy = [](const char* v) { return v && v[0] && strcmp(v, "0") != 0 && strcmp(v, "False") != 0 && strcmp(v, "false") != 0; };
    // (the runtime's own serialising switches make every launch wait for the kernel: a poll launched ahead of its
    // producer would then block the launching thread as well)
    const bool serialising_tool = truthy(pmc) || truthy(getenv("AMD_SERIALIZE_KERNEL")) || truthy(getenv("HIP_LAUNCH_BLOCKING")) ||
                                  truthy(getenv("CUDA_LAUNCH_BLOCKING"));
    lp->use_events = mode ? strcmp(mode, "event") == 0 : serialising_tool;
    int khz0 = 100000;
    CREATE_HIP(hipDeviceGetAttribute(&khz0, hipDeviceAttributeWallClockRate, ctx->device));
    if (!lp->use_events && !(mode && strcmp(mode, "flag") == 0)) {
      // Can the two streams overlap at all?  A kernel on `main` spins (<= 20 ms) for a flag that a kernel on `side` sets.
      // If it never sees it -- one hardware queue for both streams (GPU_MAX_HW_QUEUES=1), a tool that runs one kernel
      // at a time -- device-side polling would only ever time out: order the chains with stream events instead.
      uint32_t* probe = lp->d_hs + 8;  // [8] flag, [9] saw
      CREATE_HIP(hipMemset(probe, 0, 2 * sizeof(uint32_t)));
      loop_probe_wait_kernel<<<1, 1, 0, lp->main>>>(probe, probe + 1, 20ull * (unsigned long long)(khz0 > 0 ? khz0 : 100000));
      CREATE_HIP(hipGetLastError());
      loop_probe_set_kernel<<<1, 1, 0, lp->side>>>(probe);
      CREATE_HIP(hipGetLastError());
      CREATE_HIP(hipStreamSynchronize(lp->main));
      CREATE_HIP(hipStreamSynchronize(lp->side));
      uint32_t saw = 0;
      CREATE_HIP(hipMemcpy(&saw, probe + 1, sizeof saw, hipMemcpyDeviceToHost));
      if (!saw) lp->use_events = true;
    }
    if (const char* f = getenv("AMMSB_LOOP_TEST_FAIL_AT")) {  // (test hook: never set in a job's environment)
      const long v = strtol(f, nullptr, 10);
      if (v > 0) {
        lp->test_fail_at = (uint32_t)v;
        fprintf(stderr, "ammsb_loop: AMMSB_LOOP_TEST_FAIL_AT=%ld -- the sampler chain of mini-batch %ld will be made to give up (fault injection)\n", v, v);
      }
    }
    lp->host_prof = getenv("AMMSB_LOOP_HOSTPROF") != nullptr;
    const char* lm = getenv("AMMSB_LOOP_LAUNCH");
    lp->serial_launch = lm && strcmp(lm, "serial") == 0;
    lp->eager_launch = !(lm && strcmp(lm, "graph") == 0) && !lp->serial_launch && !lp->use_events;
    int khz = 100000;
    CREATE_HIP(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ctx->device));
    lp->wait_ticks = 5000ull * (unsigned long long)(khz > 0 ? khz : 100000);  // 5 s
    if (const char* w = getenv("AMMSB_LOOP_WAIT_MS")) {  // (tests: a short give-up time)
      char* end = nullptr;
      const long ms = strtol(w, &end, 10);
      if (end == w || *end != '\0' || ms < 1 || ms > 600000) {
        fprintf(stderr, "ammsb_loop: AMMSB_LOOP_WAIT_MS='%s' ignored (want 1 .. 600000); waits give up after 5 s\n", w);
      } else {
        lp->wait_ticks = (unsigned long long)ms * (unsigned long long)(khz > 0 ? khz : 100000);
        fprintf(stderr, "ammsb_loop: AMMSB_LOOP_WAIT_MS=%ld -- device-side waits give up after %ld ms\n", ms, ms);
      }
    }
  }
  for (Stage& st : lp->stage) {
    CREATE_HIP(hipHostMalloc(&st.ring, sizeof(ammsb_step_desc) * (CHUNK + 4), hipHostMallocDefault));
    CREATE_HIP(hipEventCreateWithFlags(&st.done, hipEventDisableTiming));
    st.used = false;
  }
  if (c.flags & AMMSB_LOOP_TIMESTAMPS) {
    CREATE_HIP(hipMalloc(&lp->d_stamps, sizeof(unsigned long long) * AMMSB_STAMP_SLOTS * STAMP_CAP));
    CREATE_HIP(hipMemset(lp->d_stamps, 0, sizeof(unsigned long long) * AMMSB_STAMP_SLOTS * STAMP_CAP));
    CREATE_HIP(hipDeviceGetAttribute(&lp->wall_khz, hipDeviceAttributeWallClockRate, ctx->device));
  }
#undef CREATE_HIP
  for (int link = 0; link < 2; ++link)
    for (int b = 0; b < NBUF; ++b)
      for (int dp = 0; dp < 2; ++dp) {
        int rc = capture(lp, 1, link, b, dp, &lp->exec_main[link][b][dp]);
        if (rc == AMMSB_OK) rc = capture(lp, 2, link, b, dp, &lp->exec_samp[link][b][dp]);
        if (rc != AMMSB_OK) {
          destroy(lp);
          return rc;
        }
      }
  *out = lp;
  return AMMSB_OK;
}

extern "C" int ammsb_loop_destroy(ammsb_loop* lp) {
  destroy(lp);
  return AMMSB_OK;
}

// Process-wide: a run's chains are submitted as one unit.  Two loops driven from two host threads could otherwise
// interleave their submissions on shared hardware queues so that each one's polling kernel sits ahead of the other's
// producer (a cycle across the queues: both would wait until they give up).  With whole runs submitted one after the
// other, every polling kernel's producer is ahead of it in submission order, whatever the stream-to-queue map is.
static std::mutex g_submit_mutex;

namespace {

// Enqueues steps [done0, n_steps) of a run whose mini-batches [0, avail0) are already available (avail0 = 1: only the
// pending one, the normal entry; more after a resumed run, whose earlier sampler chains had completed).
int submit(ammsb_loop* lp, const ammsb_mb_choice* pending, const ammsb_mb_choice* next, uint32_t n_steps,
           uint32_t first_step_count, uint32_t parity, hipStream_t s, uint32_t done0, uint32_t avail0) {
  ammsb_ctx* ctx = lp->ctx;
  const uint32_t p0 = parity;
  const bool ev = lp->use_events;
  using clk = std::chrono::steady_clock;
  auto us_since = [](clk::time_point t) { return std::chrono::duration<double, std::micro>(clk::now() - t).count(); };
  const clk::time_point t_run = clk::now();
  // mini-batch j of this call (j = 0: the pending one; j >= 1: next[j - 1], consumed by step j) lives in buffer set
  // (p0 + j) % 3 and was / is sampled with the neighbour-sampler streams of Sample[(p0 + j) % 2]
  auto choice = [&](uint32_t j) -> const ammsb_mb_choice& { return j == 0 ? *pending : next[j - 1]; };
  LOOP_HIP(lp, hipEventRecord(lp->ev_in, s));
  LOOP_HIP(lp, hipStreamWaitEvent(lp->main, lp->ev_in, 0));
  uint32_t done = done0;
  if (done0 == n_steps && avail0 <= n_steps) {
    // (resumption only) every step of the run has completed, but the mini-batch it leaves pending was never sampled:
    // its sampler chain was the one that gave up.  Sample it now, straight from a one-entry descriptor upload.
    Stage& st = lp->stage[lp->next_stage];
    lp->next_stage ^= 1;
    if (st.used) LOOP_HIP(lp, hipEventSynchronize(st.done));
    st.ring[0] = make_desc(lp, choice(n_steps), first_step_count + n_steps);
    LOOP_HIP(lp, hipMemcpyAsync(lp->d_ring, st.ring, sizeof(ammsb_step_desc), hipMemcpyHostToDevice, lp->main));
    LOOP_HIP(lp, hipEventRecord(st.done, lp->main));
    st.used = true;
    LOOP_HIP(lp, hipEventRecord(lp->ev_prime, lp->main));
    LOOP_HIP(lp, hipStreamWaitEvent(lp->side, lp->ev_prime, 0));
    LOOP_RC(record_sampler(lp, choice(n_steps).link ? 1 : 0, (int)((p0 + n_steps) % NBUF), (int)((p0 + n_steps) & 1u),
                           lp->d_ring, lp->side));
    if (ev) LOOP_HIP(lp, hipEventRecord(n_steps == 1 ? lp->ev_first : lp->ev_samp[(n_steps - 2) % NBUF], lp->side));
  }
  while (done < n_steps) {
    const uint32_t cnt = n_steps - done < CHUNK ? n_steps - done : CHUNK;
    const clk::time_point t_pro = clk::now();
    Stage& st = lp->stage[lp->next_stage];
    lp->next_stage ^= 1;
    if (st.used) LOOP_HIP(lp, hipEventSynchronize(st.done));  // the upload two chunks ago has long executed
    // ring[i] = descriptor of mini-batch done + i as step done + i (entries past the call's last batch repeat it:
    // they are handed over by the last steps' kernels and never used)
    for (uint32_t i = 0; i < cnt + 4; ++i) {
      const uint32_t j = done + i <= n_steps ? done + i : n_steps;
      st.ring[i] = make_desc(lp, choice(j), first_step_count + done + i);
    }
    LOOP_HIP(lp, hipMemcpyAsync(lp->d_ring, st.ring, sizeof(ammsb_step_desc) * (cnt + 4), hipMemcpyHostToDevice, lp->main));
    LOOP_HIP(lp, hipEventRecord(st.done, lp->main));
    st.used = true;
    const int dp0 = (int)((p0 + done) & 1u);
    loop_prime_kernel<<<1, 1, 0, lp->main>>>(lp->d_ring, lp->d_cursor, lp->d_cur[dp0], lp->d_nxt[(p0 + done + 2) % NBUF],
                                             ev ? nullptr : lp->d_hs + HS_TIMEOUTS);
    LOOP_HIP(lp, hipGetLastError());
    // the gradient constants of the theta this chunk starts from (ammsb_beta_step_d's theta step keeps them current
    // from here on; theta may have been stepped eagerly or restored from a checkpoint since the last run)
    LOOP_RC(ammsb_theta_coef_d(ctx, lp->c.theta, lp->c.beta, lp->main));
    LOOP_HIP(lp, hipEventRecord(lp->ev_prime, lp->main));
    if (done == done0) {
      // ramp-up: the mini-batches of steps done0 and done0 + 1 that are not there yet are sampled before the first
      // step starts (eagerly, straight from their ring entries); from then on the sampler chain of step i produces
      // mini-batch i + 2.  Normal entry (done0 = 0, the pending mini-batch available): mini-batch 1.
      const uint32_t hi = done0 + 2 < n_steps + 1 ? done0 + 2 : n_steps + 1;
      for (uint32_t j = avail0 > done0 ? avail0 : done0; j < hi; ++j) {
        if (j == 0) continue;  // (the pending mini-batch is the caller's)
        LOOP_HIP(lp, hipStreamWaitEvent(lp->side, lp->ev_prime, 0));
        LOOP_RC(record_sampler(lp, choice(j).link ? 1 : 0, (int)((p0 + j) % NBUF), (int)((p0 + j) & 1u),
                               lp->d_ring + (j - done0), lp->side));
        if (ev) LOOP_HIP(lp, hipEventRecord(j == 1 ? lp->ev_first : lp->ev_samp[(j - 2) % NBUF], lp->side));
      }
    }
    lp->t_pro_us += us_since(t_pro);
    bool threaded = !ev && cnt >= 32 && !lp->serial_launch && avail0 <= 1;
    if (threaded) {
      // With the device-side hand-shake nothing on the host orders the two chains any more: the sampler graphs of
      // the chunk are launched from a second thread while this one launches the main graphs (a launch costs
      // 8-18 us of host time, which is what a C1 step costs on the device).
      // The SUBMISSION order stays a valid order of the dependencies (a chain is submitted only after the chain its
      // wait kernel polls for): streams can share a hardware queue, where a polling kernel submitted ahead of its
      // producer would hold the queue until it gives up (seen: two learners in one process, every wait timing out).
      hipError_t side_err = hipSuccess;
      int side_rc = AMMSB_OK, main_rc = AMMSB_OK;
      std::atomic<uint32_t> main_sub(0), side_sub(0);  // chains of this chunk submitted so far
      std::atomic<bool> stop(false);                   // the other thread failed: submit nothing more
      const clk::time_point ts = clk::now();
      auto side_body = [&]() {
        side_err = hipSetDevice(ctx->device);
        if (side_err == hipSuccess) side_err = hipStreamWaitEvent(lp->side, lp->ev_prime, 0);
        for (uint32_t i = 0; i < cnt && side_err == hipSuccess && side_rc == AMMSB_OK && !stop.load(); ++i) {
          const uint32_t gi = done + i;
          if (gi + 2 > n_steps) break;
          while (main_sub.load(std::memory_order_acquire) < i) std::this_thread::yield();  // sampler(i) polls for main(i - 1)
          const int b = (int)((p0 + gi + 2) % NBUF), dp = (int)((p0 + gi) & 1u), nl = choice(gi + 2).link ? 1 : 0;
          if (lp->eager_launch) side_rc = record_sampler(lp, nl, b, dp, lp->d_nxt[b], lp->side);
          else side_err = hipGraphLaunch(lp->exec_samp[nl][b][dp], lp->side);
          side_sub.store(i + 1, std::memory_order_release);
        }
        if (side_err != hipSuccess || side_rc != AMMSB_OK) stop.store(true);
        side_sub.store(cnt + 2, std::memory_order_release);  // nothing more (or an error): never hold the main thread
      };
      std::thread helper;
      try {
        helper = std::thread(side_body);
      } catch (...) {  // no thread to be had: this chunk goes through the one-thread form below
        threaded = false;
      }
      if (threaded) {
        hipError_t main_err = hipSuccess;
        for (uint32_t i = 0; i < cnt && main_err == hipSuccess && main_rc == AMMSB_OK && !stop.load(); ++i) {
          const uint32_t gi = done + i;
          // main(i)'s last kernel polls for mini-batch i + 1, the product of sampler(i - 1)
          while (i >= 1 && side_sub.load(std::memory_order_acquire) < i) std::this_thread::yield();
          const int b = (int)((p0 + gi) % NBUF), dp = (int)((p0 + gi) & 1u), cl = choice(gi).link ? 1 : 0;
          if (lp->eager_launch) main_rc = record_main(lp, cl, b, dp, lp->main);
          else main_err = hipGraphLaunch(lp->exec_main[cl][b][dp], lp->main);
          main_sub.store(i + 1, std::memory_order_release);
        }
        if (main_err != hipSuccess || main_rc != AMMSB_OK) stop.store(true);
        main_sub.store(cnt + 2, std::memory_order_release);
        if (lp->host_prof) lp->t_main_us += us_since(ts);
        helper.join();
        if (lp->host_prof) lp->t_side_us += us_since(ts);  // (parallel form: main = this thread's loop, sampler = until joined)
        LOOP_HIP(lp, side_err);
        LOOP_HIP(lp, main_err);
        LOOP_RC(side_rc);
        LOOP_RC(main_rc);
      }
    }
    if (!threaded)
    for (uint32_t i = 0; i < cnt; ++i) {
      const uint32_t gi = done + i;  // step index within this call
      const int dp = (int)((p0 + gi) & 1u);
      // mini-batch gi + 2: sampled now, unless an earlier (interrupted) submission of this run already produced it.
      // The call ends with exactly ONE pending mini-batch (n_steps), like the eager loop.
      const bool sample = gi + 2 <= n_steps && gi + 2 >= avail0;
      if (sample) {
        // reads the descriptor the previous step's last kernel (or the prime kernel) handed over, overwrites the
        // buffer set step gi - 1 read
        // (the chain's own wait kernel holds it until main(gi - 1) is done; the first one of a chunk also needs
        // the prime kernel, which runs after that on the main stream)
        if (i == 0) LOOP_HIP(lp, hipStreamWaitEvent(lp->side, lp->ev_prime, 0));
        else if (ev) LOOP_HIP(lp, hipStreamWaitEvent(lp->side, lp->ev_main[(gi - 1) % NBUF], 0));
        const auto t0 = lp->host_prof ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
        const int nl = choice(gi + 2).link ? 1 : 0, b = (int)((p0 + gi + 2) % NBUF);
        if (lp->direct_only) LOOP_RC(record_sampler(lp, nl, b, dp, lp->d_nxt[b], lp->side));
        else LOOP_HIP(lp, hipGraphLaunch(lp->exec_samp[nl][b][dp], lp->side));
        if (lp->host_prof) lp->t_side_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (ev) LOOP_HIP(lp, hipEventRecord(lp->ev_samp[gi % NBUF], lp->side));
      }
      // main chain of step gi: its mini-batch was sampled during step gi - 2 (gi = 1: by the ramp-up above), or was
      // already there when this submission began (gi < avail0)
      if (ev && gi >= avail0) {
        if (gi == 1) LOOP_HIP(lp, hipStreamWaitEvent(lp->main, lp->ev_first, 0));
        else LOOP_HIP(lp, hipStreamWaitEvent(lp->main, lp->ev_samp[(gi - 2) % NBUF], 0));
      }
      const auto t1 = lp->host_prof ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
      const int cl = choice(gi).link ? 1 : 0, bm = (int)((p0 + gi) % NBUF);
      if (lp->direct_only) LOOP_RC(record_main(lp, cl, bm, dp, lp->main));
      else LOOP_HIP(lp, hipGraphLaunch(lp->exec_main[cl][bm][dp], lp->main));
      if (lp->host_prof) lp->t_main_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count();
      if (ev) LOOP_HIP(lp, hipEventRecord(lp->ev_main[gi % NBUF], lp->main));
    }
    lp->graphs_launched += cnt;
    done += cnt;
  }
  const clk::time_point t_epi = clk::now();
  // the pending mini-batch (n_steps): wait for its sampler, then move it into the caller's Sample[(p0 + n) % 2]
  if (ev && n_steps >= avail0) LOOP_HIP(lp, hipStreamWaitEvent(lp->main, n_steps == 1 ? lp->ev_first : lp->ev_samp[(n_steps - 2) % NBUF], 0));
  if (!ev) {
    loop_wait_kernel<<<1, 1, 0, lp->main>>>(lp->d_hs + HS_AVAIL, lp->d_hs + HS_MAIN, 1, lp->d_hs + HS_TIMEOUTS, lp->wait_ticks,
                                            nullptr, nullptr, 0u);
    LOOP_HIP(lp, hipGetLastError());
  }
  const uint32_t from = (p0 + n_steps) % NBUF, to = (p0 + n_steps) & 1u;
  if (from != to) {
    const ammsb_mb_choice& ch = choice(n_steps);
    const size_t ne = ch.link ? ch.n : lp->c.mini_batch, nv = ne + 1;
    const SampleBuf &a = lp->buf[from], &b = lp->buf[to];
    // (kernels, not copy commands: a poisoned run must leave the destination alone -- see loop_copy_kernel, which
    // copies exactly the bytes named)
    const uint32_t* poison = ev ? nullptr : lp->d_hs + HS_TIMEOUTS;
    auto copy = [&](void* dst, const void* src, size_t bytes) {
      const size_t n16 = bytes / 16;
      const unsigned blocks = (unsigned)(n16 / 256 + 1 < 1024 ? n16 / 256 + 1 : 1024);
      loop_copy_kernel<<<blocks, 256, 0, lp->main>>>(static_cast<uint4*>(dst), static_cast<const uint4*>(src), bytes, poison);
    };
    copy(b.edges, a.edges, sizeof(uint64_t) * ne);
    copy(b.nodes, a.nodes, sizeof(uint32_t) * nv);
    copy(b.neighbors, a.neighbors, sizeof(uint32_t) * nv * lp->n_nbr);
    copy(b.nbr_table, a.nbr_table, sizeof(uint32_t) * nv * 2 * lp->n_nbr);
    LOOP_HIP(lp, hipGetLastError());
  }
  LOOP_HIP(lp, hipEventRecord(lp->ev_out, lp->main));
  LOOP_HIP(lp, hipStreamWaitEvent(s, lp->ev_out, 0));
  lp->t_epi_us += us_since(t_epi);
  lp->t_run_us += us_since(t_run);
  lp->runs_done += 1;
  return AMMSB_OK;
}

}  // namespace

extern "C" int ammsb_loop_run(ammsb_loop* lp, const ammsb_mb_choice* pending, const ammsb_mb_choice* next,
                              uint32_t n_steps, uint32_t first_step_count, uint32_t parity, void* stream) {
  if (!lp) return AMMSB_EINVAL;
  ammsb_ctx* ctx = lp->ctx;
  AMMSB_CHECK_ARG(ctx, pending && (next || n_steps == 0), "null argument");
  AMMSB_CHECK_ARG(ctx, parity < 2 && first_step_count >= 1, "bad parity / step count");
  if (n_steps == 0) return AMMSB_OK;
  LOOP_RC(check_choice(lp, *pending));
  for (uint32_t i = 0; i < n_steps; ++i) LOOP_RC(check_choice(lp, next[i]));
  std::lock_guard<std::mutex> guard(g_submit_mutex);
  if (!lp->use_events) {  // what a resumption would need (ammsb_loop_check)
    try {
      ammsb_loop::Run r;
      r.pending = *pending;
      r.next.assign(next, next + n_steps);
      r.n_steps = n_steps;
      r.first_step_count = first_step_count;
      r.parity = parity;
      r.stream = as_stream(stream);
      r.main_start = lp->main_total;
      lp->runs->push_back(std::move(r));
    } catch (...) {
      return AMMSB_ENOMEM;
    }
  }
  lp->main_total += n_steps;
  return submit(lp, pending, next, n_steps, first_step_count, parity, as_stream(stream), 0u, 1u);
}

// Synchronises the loop's streams.  If a device-side wait gave up since the last call, the kernels after it were
// skipped (ammsb_step.h): the model, the streams and the sample buffers are those of the last completed step.  The
// loop then switches to the stream-event hand-over for good, re-enqueues the steps that did not run -- same kernels,
// same arguments, same order, so the trajectory is the one an undisturbed run has -- waits for them, and reports the
// fallback through ammsb_loop_status.  *wait_timeouts is the number of waits that gave up WITHOUT a successful
// resumption: 0 unless the recovery itself failed (callers treat that as an error).
extern "C" int ammsb_loop_check(ammsb_loop* lp, uint32_t* wait_timeouts) {
  if (!lp) return AMMSB_EINVAL;
  ammsb_ctx* ctx = lp->ctx;
  AMMSB_CHECK_ARG(ctx, wait_timeouts, "null argument");
  std::lock_guard<std::mutex> guard(g_submit_mutex);
  LOOP_HIP(lp, hipStreamSynchronize(lp->side));
  LOOP_HIP(lp, hipStreamSynchronize(lp->main));
  uint32_t hs[3] = {0, 0, 0};
  LOOP_HIP(lp, hipMemcpy(hs, lp->d_hs, sizeof hs, hipMemcpyDeviceToHost));
  *wait_timeouts = 0;
  if (!hs[HS_TIMEOUTS]) {
    lp->runs->clear();
    return AMMSB_OK;
  }
  // main chains completed over the loop's lifetime (the counter wraps at 2^32: take it relative to the total)
  const uint64_t done_total = lp->main_total - (uint64_t)(uint32_t)((uint32_t)lp->main_total - hs[HS_MAIN]);
  const uint32_t avail_ahead = hs[HS_AVAIL] - hs[HS_MAIN];  // mini-batches sampled and not yet consumed (1 .. 3)
  lp->use_events = true;
  lp->direct_only = true;
  lp->eager_launch = false;
  lp->fallbacks += 1;
  const uint32_t clear = 0;
  LOOP_HIP(lp, hipMemcpy(lp->d_hs + HS_TIMEOUTS, &clear, sizeof clear, hipMemcpyHostToDevice));
  std::vector<ammsb_loop::Run> runs;
  runs.swap(*lp->runs);
  // lifetime mini-batches [0, avail_total) have been sampled; a run is complete when its steps have run AND the
  // mini-batch it leaves pending (lifetime index main_start + n_steps) is among them
  const uint64_t avail_total = done_total + (avail_ahead <= 3 ? avail_ahead : 0);
  bool resumed = false;
  for (const ammsb_loop::Run& r : runs) {
    if (!resumed && r.main_start + r.n_steps <= done_total && r.main_start + r.n_steps < avail_total) continue;
    uint32_t done0 = 0, avail0 = 1;
    if (!resumed) {
      done0 = (uint32_t)(done_total > r.main_start ? done_total - r.main_start : 0);
      if (done0 > r.n_steps) done0 = r.n_steps;
      avail0 = (uint32_t)(avail_total > r.main_start ? avail_total - r.main_start : 0);
      if (avail0 == 0) avail0 = 1;  // a run's first mini-batch was the caller's: always there
      resumed = true;
    }
    const int rc = submit(lp, &r.pending, r.next.data(), r.n_steps, r.first_step_count, r.parity, r.stream, done0, avail0);
    if (rc != AMMSB_OK) {
      *wait_timeouts = hs[HS_TIMEOUTS];
      return rc;
    }
  }
  LOOP_HIP(lp, hipStreamSynchronize(lp->side));
  LOOP_HIP(lp, hipStreamSynchronize(lp->main));
  // the counters are not maintained by the event hand-over: leave them at a consistent boundary state
  const uint32_t fix[3] = {(uint32_t)lp->main_total, (uint32_t)lp->main_total + 1u, 0u};
  LOOP_HIP(lp, hipMemcpy(lp->d_hs, fix, sizeof fix, hipMemcpyHostToDevice));
  return AMMSB_OK;
}

// how the loop's two chains are ordered (0 = device-side polling, 1 = stream events) and how many times a run had to
// be finished on the event hand-over after a device-side wait gave up
extern "C" int ammsb_loop_status(const ammsb_loop* lp, uint32_t* event_handover, uint32_t* fallbacks) {
  if (!lp) return AMMSB_EINVAL;
  if (event_handover) *event_handover = lp->use_events ? 1u : 0u;
  if (fallbacks) *fallbacks = lp->fallbacks;
  return AMMSB_OK;
}

// all stamps of steps first_step .. first_step + n_steps - 1 in ns: out[i][AMMSB_STAMP_*] (AMMSB_STAMP_SLOTS doubles per step)
extern "C" int ammsb_loop_step_stamps(ammsb_loop* lp, uint32_t first_step, uint32_t n_steps, double* out_ns) {
  if (!lp) return AMMSB_EINVAL;
  ammsb_ctx* ctx = lp->ctx;
  AMMSB_CHECK_ARG(ctx, lp->d_stamps, "the loop was created without AMMSB_LOOP_TIMESTAMPS");
  AMMSB_CHECK_ARG(ctx, out_ns && n_steps <= STAMP_CAP, "bad argument (at most 8192 steps are kept)");
  LOOP_HIP(lp, hipStreamSynchronize(lp->main));
  const size_t words = (size_t)AMMSB_STAMP_SLOTS * STAMP_CAP;
  unsigned long long* host = new (std::nothrow) unsigned long long[words];
  if (!host) return AMMSB_ENOMEM;
  const hipError_t e = hipMemcpy(host, lp->d_stamps, sizeof(unsigned long long) * words, hipMemcpyDeviceToHost);
  if (e != hipSuccess) {
    delete[] host;
    snprintf(ctx->err, sizeof ctx->err, "ammsb_loop_step_stamps: hipMemcpy -> %s", hipGetErrorString(e));
    return AMMSB_EHIP;
  }
  const double ns_per_tick = 1.0e6 / static_cast<double>(lp->wall_khz > 0 ? lp->wall_khz : 100000);
  for (uint32_t i = 0; i < n_steps; ++i) {
    const uint32_t slot = (first_step + i) % STAMP_CAP;
    for (uint32_t w = 0; w < AMMSB_STAMP_SLOTS; ++w)
      out_ns[(size_t)i * AMMSB_STAMP_SLOTS + w] = static_cast<double>(host[(size_t)AMMSB_STAMP_SLOTS * slot + w]) * ns_per_tick;
  }
  delete[] host;
  return AMMSB_OK;
}

extern "C" int ammsb_loop_timestamps(ammsb_loop* lp, uint32_t first_step, uint32_t n_steps, double* begin_ns,
                                     double* end_ns) {
  if (!lp) return AMMSB_EINVAL;
  ammsb_ctx* ctx = lp->ctx;
  AMMSB_CHECK_ARG(ctx, begin_ns && end_ns && n_steps <= STAMP_CAP, "bad argument (at most 8192 steps are kept)");
  double* all = new (std::nothrow) double[(size_t)AMMSB_STAMP_SLOTS * (n_steps ? n_steps : 1)];
  if (!all) return AMMSB_ENOMEM;
  const int rc = ammsb_loop_step_stamps(lp, first_step, n_steps, all);
  for (uint32_t i = 0; rc == AMMSB_OK && i < n_steps; ++i) {
    begin_ns[i] = all[(size_t)i * AMMSB_STAMP_SLOTS + AMMSB_STAMP_PHI];
    end_ns[i] = all[(size_t)i * AMMSB_STAMP_SLOTS + AMMSB_STAMP_PI];
  }
  delete[] all;
  return rc;
}
