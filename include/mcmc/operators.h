// Device-side operators with the reference's class names and call signatures:
//   RowPartitionedMatrix<Float> / Factory   mcmc/partitioned-alloc.h:73-187
//   OpenClSet / OpenClSetFactory            mcmc/cuckoo.h:69-105   (device image of a cuckoo Set)
//   random::OpenClRandom / Factory          mcmc/random.h:22-62
//   random::RandomAndNormalize, RandomGammaAndNormalize  mcmc/random.h:64-79
//   NeighborSampler, Sample                 mcmc/sample.h:16-92
//   PhiUpdater                              mcmc/phi.h:10-61
//   BetaUpdater                             mcmc/beta.h:15-72
//   PerplexityCalculator                    mcmc/perplexity.h:23-114
// Each forwards to one C-ABI call of include/ammsb.h.  The compileFlags / baseFuncs constructor
// arguments of the reference (JIT inputs) are accepted and ignored.  Fatal conditions throw
// std::runtime_error with the reference's message where it LOG(FATAL)s.
#ifndef MCMC_AMD_OPERATORS_H_
#define MCMC_AMD_OPERATORS_H_

#include <functional>
#include <istream>
#include <memory>
#include <ostream>
#include <string>
#include <vector>

#include "ammsb.h"
#include "mcmc/config.h"
#include "mcmc/device.h"

namespace mcmc {

// One ammsb context per (device, Config constants); shared by the operators built from one Config.
std::shared_ptr<ammsb_ctx> AcquireContext(const Config& cfg, const clcuda::Queue& queue);
void ThrowIfError(ammsb_ctx* ctx, int rc, const char* what);

template <class T>
class RowPartitionedMatrix {
 public:
  RowPartitionedMatrix(clcuda::Queue queue, uint32_t rows, uint32_t cols, uint32_t rows_in_block = 0);
  uint32_t Rows() const { return rows_; }
  uint32_t Cols() const { return cols_; }
  uint32_t RowsPerBlock() const { return rows_per_alloc_; }
  std::vector<clcuda::Buffer<T>>& Blocks() { return blocks_; }
  const ammsb_rpm& Get() const { return desc_; }
  // Exchange storage with `other` (same shape): every holder of a pointer to this matrix then works on other's blocks
  // (Learner::PlacePi keeps the allocation of pi that update_phi runs fastest over).
  void SwapStorage(RowPartitionedMatrix& other);

 private:
  clcuda::Queue queue_;
  uint32_t rows_, cols_, rows_per_alloc_;
  std::vector<clcuda::Buffer<T>> blocks_;
  ammsb_rpm desc_;
};

template <class T>
class RowPartitionedMatrixFactory {
 public:
  static std::shared_ptr<RowPartitionedMatrixFactory> New(clcuda::Queue queue) {
    return std::shared_ptr<RowPartitionedMatrixFactory>(new RowPartitionedMatrixFactory(queue));
  }
  RowPartitionedMatrix<T>* CreateMatrix(uint32_t rows, uint32_t cols, uint32_t rows_in_block = 0) {
    return new RowPartitionedMatrix<T>(queue_, rows, cols, rows_in_block);
  }

 private:
  explicit RowPartitionedMatrixFactory(clcuda::Queue queue) : queue_(queue) {}
  clcuda::Queue queue_;
};

class OpenClSet {
 public:
  OpenClSet(clcuda::Queue queue, const Set& set);
  const ammsb_set& Get() const { return desc_; }

 private:
  clcuda::Buffer<Edge> data_;
  ammsb_set desc_;
};

class OpenClSetFactory {
 public:
  static std::shared_ptr<OpenClSetFactory> New(clcuda::Queue queue) {
    return std::shared_ptr<OpenClSetFactory>(new OpenClSetFactory(queue));
  }
  OpenClSet* CreateSet(const Set& set) { return new OpenClSet(queue_, set); }

 private:
  explicit OpenClSetFactory(clcuda::Queue queue) : queue_(queue) {}
  clcuda::Queue queue_;
};

namespace random {

typedef ::mcmc::ulong2 random_seed_t;

class OpenClRandom {
 public:
  OpenClRandom(clcuda::Queue queue, uint64_t size, random_seed_t seed);
  void SetSeed(random_seed_t seed);
  clcuda::Buffer<ammsb_seed>& GetSeeds() { return data_; }
  ammsb_seed* Get() { return data_.data(); }
  bool Serialize(std::ostream* out);  // random.cc:71-77
  bool Parse(std::istream* in);

 private:
  clcuda::Queue queue_;
  clcuda::Buffer<ammsb_seed> data_;
};

class OpenClRandomFactory {
 public:
  static std::shared_ptr<OpenClRandomFactory> New(clcuda::Queue queue) {
    return std::shared_ptr<OpenClRandomFactory>(new OpenClRandomFactory(queue));
  }
  OpenClRandom* CreateRandom(uint64_t size, random_seed_t seed) { return new OpenClRandom(queue_, size, seed); }

 private:
  explicit OpenClRandomFactory(clcuda::Queue queue) : queue_(queue) {}
  clcuda::Queue queue_;
};

// random.h:70-79: fill `base` from a host generator, copy to `norm`, normalise rows of `cols`
void RandomAndNormalize(clcuda::Queue* queue, const std::function<Float()>& gen, clcuda::Buffer<Float>* base,
                        clcuda::Buffer<Float>* norm, uint32_t cols);
// random.cc:159-167
void RandomGammaAndNormalize(clcuda::Queue* queue, Float eta0, Float eta1, RowPartitionedMatrix<Float>* norm,
                             clcuda::Buffer<Float>* sum);

}  // namespace random

class NeighborSampler {
 public:
  NeighborSampler(const Config& cfg, clcuda::Queue queue);
  void operator()(uint32_t num_samples, clcuda::Buffer<Vertex>* nodes);
  clcuda::Buffer<Vertex>& GetHash() { return hash_; }
  clcuda::Buffer<Vertex>& GetData() { return data_; }
  uint32_t HashCapacityPerSample() { return capacity_; }
  uint32_t DataSizePerSample() { return n_; }
  random::OpenClRandom& Rand() { return rand_; }  // stream states / work-group size, for the captured-graph loop
  uint32_t Local() const { return local_; }
  bool Serialize(std::ostream* out);  // sample.h:30-36
  bool Parse(std::istream* in);

 private:
  std::shared_ptr<ammsb_ctx> ctx_;
  clcuda::Queue queue_;
  uint32_t n_, capacity_, local_;
  bool async_;
  uint64_t max_nodes_;
  clcuda::Buffer<Vertex> hash_, data_;
  random::OpenClRandom rand_;
};

struct Sample {  // sample.h:51-92
  clcuda::Queue queue;
  std::vector<Edge> edges;
  clcuda::Buffer<Edge> dev_edges;
  std::vector<Vertex> nodes_vec;
  clcuda::Buffer<Vertex> dev_nodes;
  unsigned int seed;
  NeighborSampler neighbor_sampler;
  // sizes of the mini-batch in dev_edges / dev_nodes (== edges.size() / nodes_vec.size() with the host samplers;
  // the device sampler fills the device buffers only)
  uint32_t num_edges = 0, num_nodes = 0;
  Sample(const Config& cfg, clcuda::Queue queue);                     // seed = rand(), as sample.cc:132
  Sample(const Config& cfg, clcuda::Queue queue, unsigned int seed);  // reproducible
  bool Serialize(std::ostream* out);  // sample.h:62-91
  bool Parse(std::istream* in);
};

// Device times of enqueue-only launches (Config::async_launch): an event pair is recorded around a launch and read
// back at the caller's next synchronisation point; nothing waits inside the loop.
class DeferredTimer {
 public:
  DeferredTimer() = default;
  DeferredTimer(const DeferredTimer&) = delete;
  DeferredTimer& operator=(const DeferredTimer&) = delete;
  ~DeferredTimer();
  void Start(void* stream);                // start event of the launch that follows
  void Stop(void* stream, uint64_t* acc);  // stop event; *acc += elapsed ns once resolved
  void Resolve(bool all = true);           // the work is known to have finished (all) / the older half has (else)

 private:
  struct Rec {
    void* start;
    void* stop;
    uint64_t* acc;
  };
  void* Take();
  std::vector<Rec> pending_;
  std::vector<void*> free_;
  void* cur_ = nullptr;
};

class PhiUpdater {
 public:
  PhiUpdater(const Config& cfg, clcuda::Queue queue, clcuda::Buffer<Float>& beta, RowPartitionedMatrix<Float>* pi,
             clcuda::Buffer<Float>& phi, OpenClSet* trainingSet, const std::vector<std::string>& compileFlags = {},
             const std::string& baseFuncs = "");
  void operator()(clcuda::Buffer<Vertex>& mini_batch_nodes, clcuda::Buffer<Vertex>& neighbors,
                  uint32_t num_mini_batch_nodes);
  uint64_t UpdatePhiTime() const { return t_update_phi_; }  // ns of device time (hip events)
  uint64_t UpdatePiTime() const { return t_update_pi_; }
  clcuda::Buffer<Float>& GetPhiVec() { return phi_vec_; }
  // the two halves of operator() for a sharding caller (mcmc::Learner with an Exchange): BeginCall() counts the
  // iteration (phi.cc:739), UpdatePhi enqueues the virtual groups [group_begin, group_end) only, UpdatePi applies
  // phi_vec rows [0, n) once they are complete everywhere.  Enqueue-only: the caller synchronises.
  void BeginCall() { ++count_calls_; }
  void UpdatePhi(clcuda::Buffer<Vertex>& nodes, clcuda::Buffer<Vertex>& neighbors, uint32_t n, uint32_t group_begin,
                 uint32_t group_end);
  void UpdatePi(clcuda::Buffer<Vertex>& nodes, uint32_t n);
  random::OpenClRandom& Rand() { return rand_; }  // for the captured-graph loop (ammsb_loop)
  uint32_t& CountCalls() { return count_calls_; }
  uint32_t Local() const { return local_; }
  uint32_t Flags() const { return flags_; }
  // enqueue-only loops: read back the pending event pairs (the stream has been drained) / add device times measured
  // elsewhere (the captured-graph loop's in-kernel stamps)
  void ResolveTimers() { timers_.Resolve(); }
  void AddTimes(uint64_t phi_ns, uint64_t pi_ns) {
    t_update_phi_ += phi_ns;
    t_update_pi_ += pi_ns;
  }
  bool Serialize(std::ostream* out);  // phi.cc:765-784
  bool Parse(std::istream* in);

 private:
  std::shared_ptr<ammsb_ctx> ctx_;
  clcuda::Queue queue_;
  clcuda::Buffer<Float>& beta_;
  RowPartitionedMatrix<Float>* pi_;
  clcuda::Buffer<Float>& phi_;
  OpenClSet* trainingSet_;
  uint64_t max_nodes_;
  clcuda::Buffer<Float> phi_vec_;
  random::OpenClRandom rand_;
  uint32_t count_calls_, local_, flags_;
  bool async_, timed_;
  uint64_t t_update_phi_, t_update_pi_;
  DeferredTimer timers_;
};

class BetaUpdater {
 public:
  enum Mode { EDGE_PER_THREAD, EDGE_PER_WORKGROUP };
  BetaUpdater(Mode mode, const Config& cfg, clcuda::Queue queue, clcuda::Buffer<Float>& theta,
              clcuda::Buffer<Float>& beta, RowPartitionedMatrix<Float>* pi, OpenClSet* trainingSet,
              const std::vector<std::string>& compileFlags = {}, const std::string& baseFuncs = "");
  void operator()(clcuda::Buffer<Edge>* edges, uint32_t num_edges, Float scale);
  clcuda::Buffer<Float>& GetGrads() { return grads_; }
  clcuda::Buffer<Float>& GetThetaSum();  // beta.h:27: as of the last operator()
  // halves of operator() for a sharding caller: the gradient over edges [edge_begin, edge_end) into `out` ([2K]),
  // then the theta / beta step from an externally reduced gradient.  Enqueue-only.
  void BeginCall() { ++count_calls_; }
  void CalculateGrads(clcuda::Buffer<Edge>* edges, uint32_t num_edges, uint32_t edge_begin, uint32_t edge_end, Float* out);
  void UpdateTheta(Float scale, const Float* grads);
  // PhiUpdater::UpdatePi over nodes[0 .. num_edges] and CalculateGrads over the whole mini-batch as one launch
  // (ammsb_update_pi_beta_grads: node-stratified mini-batches, shapes CanFuseUpdatePi accepts).  Enqueue-only.
  bool CanFuseUpdatePi(uint32_t phi_local) const;
  void UpdatePiAndGrads(clcuda::Buffer<Float>& phi_sum, clcuda::Buffer<Float>& phi_vec, clcuda::Buffer<Vertex>& nodes,
                        clcuda::Buffer<Edge>* edges, uint32_t num_edges, Float* out);
  random::OpenClRandom& Rand() { return rand_; }  // for the captured-graph loop (ammsb_loop)
  uint32_t& CountCalls() { return count_calls_; }
  uint32_t Local() const { return local_; }
  // device time in ns.  The reference's five stages (beta.h:30-34) are two launches here: theta_sum +
  // partial gradients + their sum are one call, update_theta + theta->beta normalisation the other.
  uint64_t ThetaSumTime() const { return 0; }
  uint64_t GradsPartialTime() const { return t_grads_; }
  uint64_t GradsSumTime() const { return 0; }
  uint64_t UpdateThetaTime() const { return t_update_theta_; }
  uint64_t NormalizeTime() const { return 0; }
  void ResolveTimers() { timers_.Resolve(); }
  void AddTimes(uint64_t grads_ns, uint64_t update_theta_ns) {
    t_grads_ += grads_ns;
    t_update_theta_ += update_theta_ns;
  }
  bool Serialize(std::ostream* out);  // beta.cc:386-413
  bool Parse(std::istream* in);

 private:
  std::shared_ptr<ammsb_ctx> ctx_;
  clcuda::Queue queue_;
  clcuda::Buffer<Float>& theta_;
  clcuda::Buffer<Float>& beta_;
  RowPartitionedMatrix<Float>* pi_;
  OpenClSet* trainingSet_;
  random::OpenClRandom rand_;
  uint32_t count_calls_, local_;
  clcuda::Buffer<Float> grads_;
  uint64_t t_grads_ = 0, t_update_theta_ = 0;
  clcuda::Buffer<Float> theta_sum_;
  bool async_ = false, timed_ = false;
  DeferredTimer timers_;
};

class PerplexityCalculator {
 public:
  enum Mode { EDGE_PER_THREAD, EDGE_PER_WORKGROUP };
  PerplexityCalculator(Mode mode, const Config& cfg, clcuda::Queue queue, clcuda::Buffer<Float>& beta,
                       RowPartitionedMatrix<Float>* pi, clcuda::Buffer<Edge>& edges, OpenClSet* edgeSet,
                       const std::vector<std::string>& compileFlags = {}, const std::string& baseFuncs = "");
  Float operator()();  // returns -average log-likelihood (perplexity.cc:273)
  // for a sharding caller: BeginCall() counts the evaluation (perplexity.cc:252), Partial enqueues edges
  // [edge_begin, edge_end) and returns the DEVICE sums record of this calculator (valid once the stream is drained)
  void BeginCall() { ++count_calls_; }
  ammsb_ppx_sums* Partial(uint32_t edge_begin, uint32_t edge_end);
  uint32_t NumEdges() { return static_cast<uint32_t>(edges_.Count()); }
  clcuda::Buffer<Float>& PerEdge() { return ppx_per_edge_; }  // the running means, indexed by edge position
  uint64_t PerplexityTime() const { return t_ppx_; }  // ns; the four reductions are inside the same launch
  uint64_t AccumulateTime() const { return 0; }
  bool Serialize(std::ostream* out);  // perplexity.cc:276-293
  bool Parse(std::istream* in);

 private:
  std::shared_ptr<ammsb_ctx> ctx_;
  clcuda::Queue queue_;
  clcuda::Buffer<Float>& beta_;
  RowPartitionedMatrix<Float>* pi_;
  clcuda::Buffer<Edge>& edges_;
  OpenClSet* edgeSet_;
  clcuda::Buffer<Float> ppx_per_edge_;
  clcuda::Buffer<ammsb_ppx_sums> sums_;
  std::shared_ptr<ammsb_ppx_sums> host_sums_;  // host-mapped pinned memory the kernel writes operator()()'s result into
  uint32_t count_calls_, local_;
  uint64_t t_ppx_ = 0;
};

}  // namespace mcmc

#endif  // MCMC_AMD_OPERATORS_H_
