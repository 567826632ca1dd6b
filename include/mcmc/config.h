// mcmc::Config: drop-in for the reference's mcmc/config.h:25-102 (same field names, same defaults).
#ifndef MCMC_AMD_CONFIG_H_
#define MCMC_AMD_CONFIG_H_

#include <iosfwd>
#include <memory>
#include <string>
#include <vector>

#include "ammsb.h"
#include "mcmc/data.h"
#include "mcmc/sample.h"

namespace mcmc {

enum PhiUpdaterMode {
  PHI_NODE_PER_THREAD,
  PHI_NODE_PER_WORKGROUP_NAIVE,
  PHI_NODE_PER_WORKGROUP_SHARED,
  PHI_NODE_PER_WORKGROUP_CODE_GEN
};

std::istream& operator>>(std::istream& in, PhiUpdaterMode& mode);
std::string to_string(const PhiUpdaterMode& mode);

struct Config {
  Float heldout_ratio;
  Float alpha;
  Float a, b, c;
  Float epsilon;
  Float eta0, eta1;
  uint64_t K;
  uint64_t mini_batch_size;
  uint64_t num_node_sample;
  uint64_t N;
  uint64_t E;
  std::vector<Edge> training_edges;
  std::vector<Edge> heldout_edges;
  std::unique_ptr<mcmc::Set> training;
  std::unique_ptr<mcmc::Set> heldout;
  std::unique_ptr<mcmc::Graph> trainingGraph;
  std::unique_ptr<mcmc::Graph> heldoutGraph;

  uint32_t ppx_wg_size;
  uint32_t ppx_interval;
  uint32_t neighbor_sampler_wg_size;
  uint32_t phi_wg_size;
  uint32_t beta_wg_size;

  ulong2 phi_seed;
  ulong2 beta_seed;
  ulong2 neighbor_seed;

  bool phi_disable_noise;

  // MCMC_CALC_TRAIN_PPX (CMakeLists.txt:41, config.h:26-28,71-73, learner.cc:47-75) as a run-time switch instead of
  // a build option: a second PerplexityCalculator over a subset of the training edges plus sampled non-links.
  bool calc_train_ppx;        // default false = the reference's build default (OFF)
  Float training_ppx_ratio;   // subset of training edges used for training ppx (default 0.01)
  unsigned int training_ppx_seed;  // new: rand_r seed of the sampled non-links (the reference draws them from the
                                   // process-global rand(), which the HIP runtime's start-up also consumes)

  SampleStrategy strategy;

  // The three work-group modes run the same arithmetic on the same lane / column / stream map (phi.cc:214-606) and
  // select the one HIP kernel family (a warning names the mode when it is not NAIVE).  PHI_NODE_PER_THREAD -- the
  // reference's CPU-device shape, one RNG stream per node -- is refused by PhiUpdater (std::invalid_argument).
  // phi_vector_width > 1 changes results in the reference (Floatn column ownership) and is NOT reproduced: the
  // operators warn and compute width 1.  sum_grads_vector_width has no numerical effect (beta.cc:39-49).
  PhiUpdaterMode phi_mode;
  bool phi_probs_shared;
  bool phi_grads_shared;
  bool phi_pi_shared;
  uint32_t phi_vector_width;
  uint32_t sum_grads_vector_width;

  // new: draw mini-batches on the device instead of with the host samplers above
  bool device_sampling;
  // new: enqueue-only operators (no Finish() after a launch) and, with device_sampling, a two-stream loop ordered by
  // events instead of host joins.  Results are identical.  The per-kernel times of PrintStats come from event pairs
  // recorded around every launch and read back at the next synchronisation point (loop_timers below).
  bool async_launch;
  // new: with async_launch + device_sampling, enqueue whole iterations as captured hipGraphs (ammsb_loop,
  // include/ammsb.h): one hipGraphLaunch per iteration instead of a dozen kernel launches.  Results are identical.
  bool graph_launch;
  // new: per-kernel device times for PrintStats (learner.cc:252-299) in the enqueue-only loops: event pairs around
  // each launch (async_launch), device time stamps written by the kernels themselves (graph_launch).  On by default;
  // off removes two event records per launch / a few stores per step.
  bool loop_timers;
  // new: multi-GPU.  Non-null with world() > 1 => mcmc::Learner shards every iteration over the ranks (one process
  // per GPU, same Config and data on every rank) and exchanges through it (include/mcmc/exchange.h).
  std::shared_ptr<class Exchange> exchange;
  ulong2 device_sampling_seed;  // new: streams of the device sampler's candidate draws
  uint64_t device_sampling_host_seed;  // new: host generator choosing (link?, u) per mini-batch
  // new: rand_r seeds of the two Sample buffers.  The reference takes them from the process-global
  // rand() (sample.cc:132), which is not reproducible once the HIP runtime shares the process (its
  // start-up consumes rand() too); the defaults are what rand() returns first after srand(1).
  unsigned int sample_seed[2];
  // new (multi-GPU, with `exchange`): the sharded update_phi as in mcmc-ammsb-gpu_amd/learner.py -- the first
  // phi_replicate * 65535 virtual groups are computed by every rank (nothing to send), the rest are cut into
  // world * phi_chunks blocks whose exchange overlaps the next block's update_phi.  phi_replicate < 0: measured at
  // start-up (one timed update_phi and one timed all-gather; rank 0's answer is adopted by every rank); 0 = every
  // group exchanged.
  uint32_t phi_chunks;
  Float phi_replicate;
  // new (multi-GPU): a mini-batch with at most this many edges is not cut over the ranks -- every rank computes its
  // whole gradient (a link mini-batch has a few dozen edges: the all-gather of the partial sums would cost more than
  // the gradient, and the result is then the single-GPU one bit for bit)
  uint32_t beta_shard_min_edges;
  // new (multi-GPU): 0 = "sharded" (edge slices, all-gather of the partial sums, added in rank order), 1 =
  // "replicated" (every rank the whole gradient: no collective, theta bit-identical to one GPU's), -1 = auto:
  // replicated where update_pi can be folded into the gradient launch (device-sampled Node mini-batches, K <= 1024)
  int beta_grads;
  // new: where pi lands in HBM moves update_phi's launch time by up to 10 % (profiles/README.md, round 4): at start-up
  // this many allocations of pi are timed under update_phi and the fastest is kept (only when pi is >= 1 GB and the
  // candidates fit in a third of the free HBM; 0 or 1: off; AMMSB_PI_CANDIDATES overrides)
  uint32_t pi_placement_candidates;

  Config();
};

std::ostream& operator<<(std::ostream& out, const Config& cfg);

// The -D constants of MakeCompileFlags (config.cc:66-83) as the POD the C ABI takes, including the
// "%e" round trip every float goes through there.
ammsb_params MakeKernelParams(const Config& cfg);

// kept for callers that log them; the strings match config.cc:66-83
std::vector<std::string> MakeCompileFlags(const Config& cfg);

}  // namespace mcmc

#endif  // MCMC_AMD_CONFIG_H_
