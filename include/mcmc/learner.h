// mcmc::Learner: drop-in for the reference's mcmc/learner.h:18-88 (same constructor, same methods).
#ifndef MCMC_AMD_LEARNER_H_
#define MCMC_AMD_LEARNER_H_

#include <signal.h>

#include <future>
#include <map>
#include <istream>
#include <memory>
#include <ostream>
#include <random>

#include "mcmc/config.h"
#include "mcmc/data.h"
#include "mcmc/operators.h"

namespace mcmc {

class Learner {
 public:
  Learner(const Config& cfg, clcuda::Queue queue);
  ~Learner();

  void Run(uint32_t max_iters, sig_atomic_t* signaled = nullptr);  // learner.cc:214-250
  Float HeldoutPerplexity();                                       // learner.cc:196-203
  Float TrainingPerplexity();                                      // learner.cc:204-212 (Config::calc_train_ppx)
  void PrintStats(std::ostream& out);                              // learner.cc:252-299
  void PrintStats();
  bool Serialize(std::ostream* out);  // learner.cc:301-330
  bool Parse(std::istream* in);       // learner.cc:332-363

  // read-back helpers for tests / drivers (not in the reference API)
  std::vector<Float> GetBeta();
  std::vector<Float> GetTheta();
  std::vector<Float> GetPiRow(Vertex v);
  uint64_t MiniBatchEdges() const { return edges_done_; }

 private:
  Float DoSample(Sample* sample);        // learner.cc:175-194
  Float DoSampleDevice(Sample* sample);  // Config::device_sampling: csrc/ammsb_minibatch.hip instead of sample.cc
  bool SerializeDeviceSampler(std::ostream* out);
  void RunAsync(uint32_t max_iters, sig_atomic_t* signaled);  // Config::async_launch + device_sampling
  void RunGraph(uint32_t max_iters, sig_atomic_t* signaled);  // Config::graph_launch: iterations as captured graphs
  void AccountLoopStamps(uint32_t first_step, uint32_t n_steps);  // in-kernel time stamps -> PrintStats categories
  ammsb_mb_choice ChooseDevice();                             // the next mini-batch: (link?, u, deg(u), candidates)
  Float EnqueueDevice(Sample* sample, const ammsb_mb_choice& choice);
  void DrainAsync();
  bool ParseDeviceSampler(std::istream* in);
  uint32_t CandidatesFor(uint64_t u);  // candidate draws of a non-link mini-batch of vertex u
  uint32_t CandidatesForExcluded(uint32_t excluded);
  void CheckDeviceSampler();           // throws if a mini-batch came up short since the last check
  // multi-GPU (Config::exchange with world() > 1; new, see include/mcmc/exchange.h): every rank runs the same
  // learner on the same data and seeds, computes its block of update_phi's virtual groups / its slice of the
  // gradient's and the perplexity's edges, and exchanges phi_vec rows, [2K] gradient partials and the 4 sums.
  bool Sharded() const;
  void Step(Sample& s, Float weight);  // phi, pi, beta of one iteration (sharded or not)
  void StepSharded(Sample& s, Float weight);
  Float Perplexity(PerplexityCalculator* calc);
  void GatherShardedState();  // before a checkpoint: owners hand out their phi streams and running means

  const Config& cfg_;
  clcuda::Queue queue_;
  clcuda::Buffer<Float> beta_;
  clcuda::Buffer<Float> theta_;
  std::shared_ptr<RowPartitionedMatrixFactory<Float>> allocFactory_;
  std::unique_ptr<RowPartitionedMatrix<Float>> pi_;
  clcuda::Buffer<Float> phi_;
  std::shared_ptr<OpenClSetFactory> setFactory_;
  std::unique_ptr<OpenClSet> trainingSet_;
  std::unique_ptr<OpenClSet> heldoutSet_;
  clcuda::Buffer<Edge> heldoutEdges_;
  // Config::calc_train_ppx (the reference's MCMC_CALC_TRAIN_PPX members, learner.h:65-69)
  std::vector<Edge> trainingPerplexityEdges_;
  std::unique_ptr<clcuda::Buffer<Edge>> devTrainingPerplexityEdges_;
  std::unique_ptr<PerplexityCalculator> trainingPerplexity_;
  PerplexityCalculator heldoutPerplexity_;
  PhiUpdater phiUpdater_;
  BetaUpdater betaUpdater_;
  SamplerFn sampler_;
  uint32_t stepCount_;
  uint64_t time_, samplingTime_, edges_done_;
  // device sampler state (only with Config::device_sampling)
  std::shared_ptr<ammsb_ctx> ctx_;
  std::unique_ptr<clcuda::Buffer<uint64_t>> csr_offsets_;
  std::unique_ptr<clcuda::Buffer<Vertex>> csr_targets_;
  std::vector<uint32_t> degree_;
  std::vector<uint32_t> excluded_;  // 1 + training degree + held-out link degree
  std::map<uint32_t, uint32_t> cand_cache_;
  uint32_t candidates_ = 0;
  std::unique_ptr<random::OpenClRandom> mb_rand_;
  std::unique_ptr<clcuda::Buffer<uint8_t>> mb_workspace_;
  std::unique_ptr<clcuda::Buffer<uint32_t>> mb_count_;
  std::mt19937_64 host_rng_;
  // async loop: per sample, `ready` (sampling done, recorded on the sample's stream) and `consumed` (the
  // iteration that used it is done, recorded on the main stream); weights of the enqueued samples
  void* ev_ready_[2] = {nullptr, nullptr};
  void* ev_consumed_[2] = {nullptr, nullptr};
  bool consumed_valid_[2] = {false, false};
  void* ev_sampler_ = nullptr;  // the device sampler's shared streams / workspace: one call at a time
  bool sampler_valid_ = false;
  uint32_t chunks_since_check_ = 0;  // RunGraph: chunks enqueued since the last ammsb_loop_check
  bool enqueued_[2] = {false, false};
  Float weights_[2] = {0, 0};
  ammsb_mb_choice choice_[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};  // what sits in each sample's buffers (device sampling)
  ammsb_loop* loop_ = nullptr;
  // Ownership map of the sharded update_phi: groups [0, g_rep_) are replicated (every rank computes them); the rest are
  // cut into world * nch_ blocks of cc_ groups, block b owned by rank b % world and exchanged in chunk b / world.
  uint32_t cc_ = 0, g_rep_ = 0, nch_ = 1;
  void SetSplit(uint32_t g_rep);
  void CalibrateSplit();              // Config::phi_replicate < 0: balance recomputing a group against receiving it
  void PlacePi();                     // Config::pi_placement_candidates: keep the allocation of pi update_phi runs fastest over
  void* xstream_ = nullptr;           // the exchanges of a step run here, beside the next block's update_phi
  void* ev_block_ = nullptr;          // main stream: a block's update_phi has been enqueued (the exchange waits for it)
  void* ev_xdone_ = nullptr;          // exchange stream: the step's exchanges are done (the main stream waits for it)
  double calib_phi_ms_ = 0, calib_xchg_ms_ = 0;
  std::unique_ptr<clcuda::Buffer<Float>> all_grads_, grads_sum_, tail_buf_;
  std::unique_ptr<clcuda::Buffer<ammsb_ppx_sums>> all_sums_;
  std::unique_ptr<Sample> samples_[2];  // MCMC_SAMPLE_PARALLEL (CMakeLists.txt:42, default ON)
  std::future<Float> futures_[2];
  int phase_;
};

// learner.cc:47-75: the first training_ppx_ratio * |training| training edges, then links * (N(N-1)/2) / E random
// pairs (u != v, MakeEdge(u, v) as drawn -- NOT canonicalised, as in the reference) that are in neither set.
std::vector<Edge> MakeEdgesForTrainingPerplexity(const Config& cfg);

}  // namespace mcmc

#endif  // MCMC_AMD_LEARNER_H_
