// ammsb_loop: whole learner iterations replayed as captured hipGraphs.
//
// The reference's loop (mcmc/learner.cc:222-247) launches one kernel at a time and waits for each
// (queue.Finish() in phi.cc:755-761, beta.cc:339-383): seven launch + wait pairs per iteration, which is what
// an iteration costs at the small configurations (N = 10k..100k: a few tens of microseconds of device work).
// Here one iteration is two hipGraphLaunch calls on two streams: the sampling chain of the NEXT mini-batch
// (sampler stream) runs beside update_phi, update_pi, the beta gradient, its partial-row sum and the theta/beta
// step of THIS mini-batch (main stream); two events per step order the double buffer.  (A fork/join inside one
// graph costs ~40 us per replay on this runtime and a single in-line chain exposes the sampler's latency; both
// were measured, tools/gpu_exp.sh.  AMMSB_LOOP_SERIAL keeps the single-stream form: one launch per step.)
//
// What changes from step to step (mini-batch sizes, eps_t, the sampler's vertex u) cannot be a kernel
// parameter of a captured graph.  It lives in device memory instead (ammsb_step_desc, ammsb_step.h): the host
// uploads the descriptors of the next <= CHUNK steps in one copy, and the last kernel of every step
// (update_theta) hands the following two descriptors to the graph that runs next.  Graphs are specialised by
// (link batch?, buffer parity) -- four main-chain and four sampler graphs, captured once.  A link
// batch has deg(u) edges, so its kernels are launched for the largest degree and read the real size from the
// descriptor; surplus blocks leave at once.
//
// Every kernel is the one the eager C ABI launches, with the same arguments in the same per-stream order, so
// the trajectory is bit-identical to the eager loop's (tests/test_gpu_graph_loop.py).
#include "ammsb_ctx.h"
#include "ammsb_step.h"

#include <string.h>

#include <new>

namespace {

constexpr uint32_t CHUNK = 1024;      // steps per descriptor upload
constexpr uint32_t STAMP_CAP = AMMSB_STAMP_CAP;  // steps whose update_phi timestamps are kept

__global__ void loop_prime_kernel(const ammsb_step_desc* ring, uint32_t* cursor, ammsb_step_desc* cur,
                                  ammsb_step_desc* nxt) {
  *cur = ring[0];
  *nxt = ring[1];
  *cursor = 0;
}

struct Stage {
  ammsb_step_desc* ring;  // pinned host staging, [CHUNK + 2]
  hipEvent_t done;        // the upload that read it has executed
  bool used;
};

}  // namespace

struct ammsb_loop {
  ammsb_ctx* ctx;
  ammsb_loop_config c;
  ammsb_rpm pi;
  ammsb_set training, heldout;
  bool has_heldout;
  hipStream_t main, side;
  hipEvent_t ev_in, ev_out, ev_prime;
  hipEvent_t ev_main[2], ev_samp[2];  // step i's main chain / sampler chain finished (alternating)
  ammsb_step_desc* d_ring;  // [CHUNK + 2]
  uint32_t* d_cursor;
  ammsb_step_desc* d_cur[2];
  ammsb_step_desc* d_nxt[2];
  unsigned long long* d_stamps;  // [STAMP_CAP][2] or null
  Stage stage[2];
  int next_stage;
  hipGraphExec_t exec[2][2][2];  // serial form: [this step link][next step link][parity]
  hipGraphExec_t exec_main[2][2];  // [this step link][parity]
  hipGraphExec_t exec_samp[2][2];  // [next step link][parity of THIS step]
  bool serial;
  uint32_t link_nodes_cap, link_edges_cap;
  float w_link, w_nonlink;
  int wall_khz;
  uint64_t graphs_launched;
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

// the sampling chain of the NEXT mini-batch into the other buffer pair (sample.cc:249-303 + learner.cc:162-194)
int record_sampler(ammsb_loop* lp, int nl, int p, hipStream_t st) {
  const ammsb_loop_config& c = lp->c;
  ammsb_ctx* ctx = lp->ctx;
  const int q = 1 - p;
  const ammsb_step_desc* nxt = lp->d_nxt[p];
  const uint32_t m = c.mini_batch;
  if (nl)
    LOOP_RC(ammsb_minibatch_link_d(ctx, c.csr_offsets, c.csr_targets, lp->link_edges_cap, c.edges[q], c.nodes[q], nxt, st));
  else
    LOOP_RC(ammsb_minibatch_nonlink_d(ctx, c.mb_seeds, c.mb_candidates, m, &lp->training,
                                      lp->has_heldout ? &lp->heldout : nullptr, c.mb_workspace, c.edges[q], c.nodes[q],
                                      c.mb_count, nxt, st));
  LOOP_RC(ammsb_sample_neighbors_d(ctx, c.nbr_seeds[q], c.nodes[q], nl ? lp->link_nodes_cap : m + 1, c.nbr_wg,
                                   c.nbr_table[q], c.neighbors[q], nxt, st));
  return AMMSB_OK;
}

// this step's chain (learner.cc:237-242)
int record_main(ammsb_loop* lp, int cl, int p, hipStream_t st) {
  const ammsb_loop_config& c = lp->c;
  ammsb_ctx* ctx = lp->ctx;
  const int q = 1 - p;
  const ammsb_step_desc* cur = lp->d_cur[p];
  const uint32_t m = c.mini_batch;
  const uint32_t cap_nodes = cl ? lp->link_nodes_cap : m + 1;
  const uint32_t cap_edges = cl ? lp->link_edges_cap : m;
  // (AMMSB_LOOP_TIMESTAMPS) block 0 of update_phi and block 0 of update_pi note the device time they start at
  LOOP_RC(ammsb_update_phi_d(ctx, c.beta, &lp->pi, c.phi_sum, &lp->training, c.nodes[p], c.neighbors[p], cap_nodes,
                             c.phi_seeds, c.phi_wg, c.phi_flags, c.phi_vec, cur, lp->d_stamps, st));
  LOOP_RC(ammsb_update_pi_d(ctx, &lp->pi, c.phi_sum, c.phi_vec, c.nodes[p], cap_nodes, c.phi_wg, cur, lp->d_stamps, st));
  const ammsb_step_advance adv = {lp->d_ring, lp->d_cursor, lp->d_cur[q], lp->d_nxt[q]};
  LOOP_RC(ammsb_beta_step_d(ctx, c.theta, c.beta, &lp->pi, &lp->training, c.edges[p], cap_edges, c.beta_wg, c.grads,
                            c.beta_seeds, c.beta_flags, cur, &adv, st));
  return AMMSB_OK;
}

// kind 0: sampler chain + main chain in line (serial form), 1: main chain, 2: sampler chain
int capture(ammsb_loop* lp, int kind, int cl, int nl, int p, hipGraphExec_t* out) {
  hipStream_t st = kind == 2 ? lp->side : lp->main;
  LOOP_HIP(lp, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  int rc = AMMSB_OK;
  if (kind == 0 || kind == 2) rc = record_sampler(lp, nl, p, st);
  if (rc == AMMSB_OK && (kind == 0 || kind == 1)) rc = record_main(lp, cl, p, st);
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
  if (lp->main) (void)hipStreamSynchronize(lp->main);
  if (lp->side) (void)hipStreamSynchronize(lp->side);
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b)
      for (int p = 0; p < 2; ++p)
        if (lp->exec[a][b][p]) (void)hipGraphExecDestroy(lp->exec[a][b][p]);
  for (int a = 0; a < 2; ++a)
    for (int p = 0; p < 2; ++p) {
      if (lp->exec_main[a][p]) (void)hipGraphExecDestroy(lp->exec_main[a][p]);
      if (lp->exec_samp[a][p]) (void)hipGraphExecDestroy(lp->exec_samp[a][p]);
    }
  for (Stage& st : lp->stage) {
    if (st.ring) (void)hipHostFree(st.ring);
    if (st.done) (void)hipEventDestroy(st.done);
  }
  for (hipEvent_t ev : {lp->ev_in, lp->ev_out, lp->ev_prime, lp->ev_main[0], lp->ev_main[1], lp->ev_samp[0], lp->ev_samp[1]})
    if (ev) (void)hipEventDestroy(ev);
  if (lp->d_ring) (void)hipFree(lp->d_ring);  // one allocation: ring, cursor, cur/nxt
  if (lp->d_stamps) (void)hipFree(lp->d_stamps);
  if (lp->side) (void)hipStreamDestroy(lp->side);
  if (lp->main) (void)hipStreamDestroy(lp->main);
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
  AMMSB_CHECK_ARG(ctx, c.csr_offsets && c.csr_targets && c.mb_seeds && c.mb_workspace && c.mb_count,
                  "null mini-batch sampler buffer");
  AMMSB_CHECK_ARG(ctx, c.mini_batch > 0 && c.max_fan_out > 0, "mini_batch / max_fan_out must be positive");
  AMMSB_CHECK_ARG(ctx, c.mb_candidates >= c.mini_batch && c.mb_candidates % 256 == 0, "bad candidate capacity");
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
  CREATE_HIP(hipStreamCreateWithFlags(&lp->main, hipStreamNonBlocking));
  CREATE_HIP(hipStreamCreateWithFlags(&lp->side, hipStreamNonBlocking));
  for (hipEvent_t* ev : {&lp->ev_in, &lp->ev_out, &lp->ev_prime, &lp->ev_main[0], &lp->ev_main[1], &lp->ev_samp[0],
                         &lp->ev_samp[1]})
    CREATE_HIP(hipEventCreateWithFlags(ev, hipEventDisableTiming));
  lp->serial = (c.flags & AMMSB_LOOP_SERIAL) != 0;
  // one device allocation: ring [CHUNK + 2], cur[2], nxt[2], cursor
  const size_t n_desc = CHUNK + 2 + 4;
  CREATE_HIP(hipMalloc(&lp->d_ring, sizeof(ammsb_step_desc) * n_desc + 64));
  CREATE_HIP(hipMemset(lp->d_ring, 0, sizeof(ammsb_step_desc) * n_desc + 64));
  lp->d_cur[0] = lp->d_ring + CHUNK + 2;
  lp->d_cur[1] = lp->d_ring + CHUNK + 3;
  lp->d_nxt[0] = lp->d_ring + CHUNK + 4;
  lp->d_nxt[1] = lp->d_ring + CHUNK + 5;
  lp->d_cursor = reinterpret_cast<uint32_t*>(lp->d_ring + n_desc);
  for (Stage& st : lp->stage) {
    CREATE_HIP(hipHostMalloc(&st.ring, sizeof(ammsb_step_desc) * (CHUNK + 2), hipHostMallocDefault));
    CREATE_HIP(hipEventCreateWithFlags(&st.done, hipEventDisableTiming));
    st.used = false;
  }
  if (c.flags & AMMSB_LOOP_TIMESTAMPS) {
    CREATE_HIP(hipMalloc(&lp->d_stamps, sizeof(unsigned long long) * 2 * STAMP_CAP));
    CREATE_HIP(hipMemset(lp->d_stamps, 0, sizeof(unsigned long long) * 2 * STAMP_CAP));
    CREATE_HIP(hipDeviceGetAttribute(&lp->wall_khz, hipDeviceAttributeWallClockRate, ctx->device));
  }
#undef CREATE_HIP
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b)
      for (int p = 0; p < 2; ++p) {
        int rc = AMMSB_OK;
        if (lp->serial) {
          rc = capture(lp, 0, a, b, p, &lp->exec[a][b][p]);
        } else if (b == 0) {
          rc = capture(lp, 1, a, 0, p, &lp->exec_main[a][p]);
          if (rc == AMMSB_OK) rc = capture(lp, 2, 0, a, p, &lp->exec_samp[a][p]);
        }
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

extern "C" int ammsb_loop_run(ammsb_loop* lp, const ammsb_mb_choice* pending, const ammsb_mb_choice* next,
                              uint32_t n_steps, uint32_t first_step_count, uint32_t parity, void* stream) {
  if (!lp) return AMMSB_EINVAL;
  ammsb_ctx* ctx = lp->ctx;
  AMMSB_CHECK_ARG(ctx, pending && (next || n_steps == 0), "null argument");
  AMMSB_CHECK_ARG(ctx, parity < 2 && first_step_count >= 1, "bad parity / step count");
  if (n_steps == 0) return AMMSB_OK;
  LOOP_RC(check_choice(lp, *pending));
  for (uint32_t i = 0; i < n_steps; ++i) LOOP_RC(check_choice(lp, next[i]));
  hipStream_t s = as_stream(stream);
  LOOP_HIP(lp, hipEventRecord(lp->ev_in, s));
  LOOP_HIP(lp, hipStreamWaitEvent(lp->main, lp->ev_in, 0));
  uint32_t done = 0;
  int p = static_cast<int>(parity);
  while (done < n_steps) {
    const uint32_t cnt = n_steps - done < CHUNK ? n_steps - done : CHUNK;
    Stage& st = lp->stage[lp->next_stage];
    lp->next_stage ^= 1;
    if (st.used) LOOP_HIP(lp, hipEventSynchronize(st.done));  // the upload two chunks ago has long executed
    // ring[i] = descriptor of step done + i (i = 0 .. cnt; entry cnt is only the last step's look-ahead)
    for (uint32_t i = 0; i <= cnt; ++i) {
      const ammsb_mb_choice& ch = (done + i == 0) ? *pending : next[done + i - 1];
      st.ring[i] = make_desc(lp, ch, first_step_count + done + i);
    }
    st.ring[cnt + 1] = st.ring[cnt];  // read by the last advance, never used
    LOOP_HIP(lp, hipMemcpyAsync(lp->d_ring, st.ring, sizeof(ammsb_step_desc) * (cnt + 2), hipMemcpyHostToDevice, lp->main));
    LOOP_HIP(lp, hipEventRecord(st.done, lp->main));
    st.used = true;
    loop_prime_kernel<<<1, 1, 0, lp->main>>>(lp->d_ring, lp->d_cursor, lp->d_cur[p], lp->d_nxt[p]);
    LOOP_HIP(lp, hipGetLastError());
    if (!lp->serial) LOOP_HIP(lp, hipEventRecord(lp->ev_prime, lp->main));
    for (uint32_t i = 0; i < cnt; ++i) {
      const int cl = st.ring[i].link ? 1 : 0, nl = st.ring[i + 1].link ? 1 : 0;
      if (lp->serial) {
        LOOP_HIP(lp, hipGraphLaunch(lp->exec[cl][nl][p], lp->main));
      } else {
        const uint32_t gi = done + i;  // step index within this call
        // sampler of the mini-batch for step gi + 1: reads the descriptors the previous step's theta kernel handed
        // over (or the prime kernel wrote) and overwrites the buffer pair that step read
        LOOP_HIP(lp, hipStreamWaitEvent(lp->side, i == 0 ? lp->ev_prime : lp->ev_main[(gi - 1) & 1], 0));
        LOOP_HIP(lp, hipGraphLaunch(lp->exec_samp[nl][p], lp->side));
        LOOP_HIP(lp, hipEventRecord(lp->ev_samp[gi & 1], lp->side));
        // main chain of step gi: its mini-batch was sampled during step gi - 1
        if (gi > 0) LOOP_HIP(lp, hipStreamWaitEvent(lp->main, lp->ev_samp[(gi - 1) & 1], 0));
        LOOP_HIP(lp, hipGraphLaunch(lp->exec_main[cl][p], lp->main));
        LOOP_HIP(lp, hipEventRecord(lp->ev_main[gi & 1], lp->main));
      }
      p ^= 1;
    }
    lp->graphs_launched += cnt;
    done += cnt;
  }
  if (!lp->serial) LOOP_HIP(lp, hipStreamWaitEvent(lp->main, lp->ev_samp[(n_steps - 1) & 1], 0));  // the pending sample
  LOOP_HIP(lp, hipEventRecord(lp->ev_out, lp->main));
  LOOP_HIP(lp, hipStreamWaitEvent(s, lp->ev_out, 0));
  return AMMSB_OK;
}

extern "C" int ammsb_loop_timestamps(ammsb_loop* lp, uint32_t first_step, uint32_t n_steps, double* begin_ns,
                                     double* end_ns) {
  if (!lp) return AMMSB_EINVAL;
  ammsb_ctx* ctx = lp->ctx;
  AMMSB_CHECK_ARG(ctx, lp->d_stamps, "the loop was created without AMMSB_LOOP_TIMESTAMPS");
  AMMSB_CHECK_ARG(ctx, begin_ns && end_ns && n_steps <= STAMP_CAP, "bad argument (at most 8192 steps are kept)");
  LOOP_HIP(lp, hipStreamSynchronize(lp->main));
  unsigned long long* host = new (std::nothrow) unsigned long long[2 * STAMP_CAP];
  if (!host) return AMMSB_ENOMEM;
  const hipError_t e = hipMemcpy(host, lp->d_stamps, sizeof(unsigned long long) * 2 * STAMP_CAP, hipMemcpyDeviceToHost);
  if (e != hipSuccess) {
    delete[] host;
    snprintf(ctx->err, sizeof ctx->err, "ammsb_loop_timestamps: hipMemcpy -> %s", hipGetErrorString(e));
    return AMMSB_EHIP;
  }
  const double ns_per_tick = 1.0e6 / static_cast<double>(lp->wall_khz > 0 ? lp->wall_khz : 100000);
  for (uint32_t i = 0; i < n_steps; ++i) {
    const uint32_t slot = (first_step + i) % STAMP_CAP;
    begin_ns[i] = static_cast<double>(host[2 * slot]) * ns_per_tick;
    end_ns[i] = static_cast<double>(host[2 * slot + 1]) * ns_per_tick;
  }
  delete[] host;
  return AMMSB_OK;
}
