// The reference's operator classes as thin forwards to the C ABI (include/mcmc/operators.h).
#include "mcmc/operators.h"
#include "mcmc/exchange.h"
#include "mcmc/serialize.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <iostream>
#include <cstring>
#include <map>
#include <mutex>
#include <stdexcept>
#include <tuple>

namespace mcmc {

void ThrowIfError(ammsb_ctx* ctx, int rc, const char* what) {
  if (rc == AMMSB_OK) return;
  throw std::runtime_error(std::string(what) + ": " + ammsb_strerror(rc) + " (" + ammsb_last_error(ctx) + ")");
}

std::shared_ptr<ammsb_ctx> AcquireContext(const Config& cfg, const clcuda::Queue& queue) {
  const ammsb_params p = MakeKernelParams(cfg);
  ammsb_ctx* ctx = nullptr;
  const int rc = ammsb_ctx_create(queue.device(), &p, &ctx);
  if (rc != AMMSB_OK) throw std::runtime_error(std::string("ammsb_ctx_create: ") + ammsb_strerror(rc));
  return std::shared_ptr<ammsb_ctx>(ctx, [](ammsb_ctx* c) { ammsb_ctx_destroy(c); });
}

namespace {
uint64_t MaxNodes(const Config& cfg) {  // phi.cc:620-622
  return std::max<uint64_t>(2 * cfg.mini_batch_size, 1 + cfg.trainingGraph->MaxFanOut());
}
// rows of phi_vec: with an Exchange the in-place all-gather region is world blocks of ceil(MAX_GROUPS / world) rows
uint64_t PhiVecRows(const Config& cfg) {
  uint64_t rows = MaxNodes(cfg);
  if (cfg.exchange && cfg.exchange->world() > 1)
    rows = std::max<uint64_t>(rows, AMMSB_MAX_GROUPS + static_cast<uint64_t>(cfg.exchange->world()) *
                                                              std::max<uint32_t>(1u, cfg.phi_chunks));
  return rows;
}
uint64_t MaxEdges(const Config& cfg) {  // sample.cc:129
  return std::max<uint64_t>(cfg.mini_batch_size, cfg.trainingGraph->MaxFanOut());
}
}  // namespace

// ------------------------------------------------------------------ RowPartitionedMatrix

template <class T>
RowPartitionedMatrix<T>::RowPartitionedMatrix(clcuda::Queue queue, uint32_t rows, uint32_t cols, uint32_t rows_in_block)
    : queue_(queue), rows_(rows), cols_(cols) {
  // partitioned-alloc.h:122-131 sizes blocks by the device's maximum allocation; on MI355X that is the
  // whole HBM, so the default is a single block.
  rows_per_alloc_ = rows_in_block ? rows_in_block : rows;
  const clcuda::Context ctx = queue_.GetContext();
  for (uint32_t i = 0; i < rows_ / rows_per_alloc_; ++i) blocks_.emplace_back(ctx, (size_t)rows_per_alloc_ * cols_);
  if (rows_ % rows_per_alloc_) blocks_.emplace_back(ctx, (size_t)(rows_ % rows_per_alloc_) * cols_);
  if (blocks_.size() > AMMSB_RPM_MAX_BLOCKS) throw std::runtime_error("more than 32 blocks");
  std::memset(&desc_, 0, sizeof desc_);
  for (size_t i = 0; i < blocks_.size(); ++i) desc_.blocks[i] = blocks_[i].data();
  desc_.rows_in_block = rows_per_alloc_;
  desc_.num_rows = rows_;
  desc_.num_cols = cols_;
  desc_.num_blocks = static_cast<uint32_t>(blocks_.size());
}
template <class T>
void RowPartitionedMatrix<T>::SwapStorage(RowPartitionedMatrix<T>& other) {
  if (rows_ != other.rows_ || cols_ != other.cols_ || rows_per_alloc_ != other.rows_per_alloc_)
    throw std::invalid_argument("RowPartitionedMatrix::SwapStorage: shapes differ");
  blocks_.swap(other.blocks_);
  std::swap(desc_, other.desc_);
}

template class RowPartitionedMatrix<Float>;
template class RowPartitionedMatrix<uint32_t>;

// ------------------------------------------------------------------------------ OpenClSet

OpenClSet::OpenClSet(clcuda::Queue queue, const Set& set) : data_(queue.GetContext(), set.Capacity()) {
  data_.Write(queue, set.Capacity(), set.Data());  // cuckoo.cc:230-231
  desc_.slots = data_.data();
  desc_.num_bins = set.BinsPerBucket();
  desc_.prime_idx = set.PrimeIdx();
}

// ---------------------------------------------------------------------------------- random

namespace random {

OpenClRandom::OpenClRandom(clcuda::Queue queue, uint64_t size, random_seed_t seed)
    : queue_(queue), data_(queue.GetContext(), size) {
  SetSeed(seed);
}

void OpenClRandom::SetSeed(random_seed_t seed) {
  // ammsb_rng_init needs no context state; a null ctx is rejected, so use a throw-away one-element params
  static std::mutex mu;
  static std::map<int, std::shared_ptr<ammsb_ctx>> ctxs;
  std::lock_guard<std::mutex> lock(mu);
  auto& c = ctxs[queue_.device()];
  if (!c) {
    ammsb_params p = {2, 1, 0, 1, 1.f, 1.f, 1.f, 1.f, 1e-7f, 1.f, 1.f};
    ammsb_ctx* raw = nullptr;
    ThrowIfError(nullptr, ammsb_ctx_create(queue_.device(), &p, &raw), "ammsb_ctx_create");
    c.reset(raw, [](ammsb_ctx* x) { ammsb_ctx_destroy(x); });
  }
  ThrowIfError(c.get(), ammsb_rng_init(c.get(), data_.data(), data_.Count(), seed[0], seed[1], queue_.stream()),
               "ammsb_rng_init");
  queue_.Finish();  // random.cc:67-68
}

bool OpenClRandom::Serialize(std::ostream* out) { return ::mcmc::Serialize(out, &data_, &queue_); }
bool OpenClRandom::Parse(std::istream* in) { return ::mcmc::Parse(in, &data_, &queue_); }

void RandomAndNormalize(clcuda::Queue* queue, const std::function<Float()>& gen, clcuda::Buffer<Float>* base,
                        clcuda::Buffer<Float>* norm, uint32_t cols) {
  if (cols != 2) throw std::runtime_error("RandomAndNormalize: only the (theta, beta) pair form is used");
  std::vector<Float> host(base->Count());
  for (Float& v : host) v = gen();
  base->Write(*queue, host.size(), host.data());
  // Normalizer(slice = 2, wg = 1), normalize.cc:13-32: lsum = (0 + a) + b
  for (size_t k = 0; k + 1 < host.size(); k += 2) {
    Float lsum = 0;
    lsum += host[k];
    lsum += host[k + 1];
    host[k] = host[k] / lsum;
    host[k + 1] = host[k + 1] / lsum;
  }
  norm->Write(*queue, host.size(), host.data());
}

void RandomGammaAndNormalize(clcuda::Queue* queue, Float eta0, Float eta1, RowPartitionedMatrix<Float>* norm,
                             clcuda::Buffer<Float>* sum) {
  OpenClRandom randv(*queue, (uint64_t)norm->Rows() * 32, random_seed_t{11, 113});  // random.cc:163-164
  Config dummy;
  dummy.N = norm->Rows();
  dummy.K = norm->Cols();
  std::shared_ptr<ammsb_ctx> ctx = AcquireContext(dummy, *queue);
  ThrowIfError(ctx.get(),
               ammsb_pi_init_gamma(ctx.get(), &norm->Get(), sum->data(), eta0, eta1, randv.Get(), queue->stream()),
               "ammsb_pi_init_gamma");
  queue->Finish();
}

}  // namespace random

// ------------------------------------------------------------------------- NeighborSampler

NeighborSampler::NeighborSampler(const Config& cfg, clcuda::Queue queue)
    : ctx_(AcquireContext(cfg, queue)),
      queue_(queue),
      n_(static_cast<uint32_t>(cfg.num_node_sample)),
      capacity_(2 * n_),
      local_(cfg.neighbor_sampler_wg_size),
      async_(cfg.async_launch),
      max_nodes_(MaxNodes(cfg)),
      hash_(queue.GetContext(), max_nodes_ * capacity_),
      data_(queue.GetContext(), max_nodes_ * n_),
      rand_(queue, max_nodes_ * capacity_, cfg.neighbor_seed) {}

void NeighborSampler::operator()(uint32_t num_samples, clcuda::Buffer<Vertex>* nodes) {
  if (num_samples > max_nodes_) throw std::runtime_error("NeighborSampler: more samples than buffer rows");
  ThrowIfError(ctx_.get(),
               ammsb_sample_neighbors(ctx_.get(), rand_.Get(), nodes->data(), num_samples, local_, hash_.data(),
                                      data_.data(), queue_.stream()),
               "ammsb_sample_neighbors");
  if (!async_) queue_.Finish();  // sample.cc:120
}

bool NeighborSampler::Serialize(std::ostream* out) { return rand_.Serialize(out) && ::mcmc::Serialize(out, &data_, &queue_); }
bool NeighborSampler::Parse(std::istream* in) { return rand_.Parse(in) && ::mcmc::Parse(in, &data_, &queue_); }

Sample::Sample(const Config& cfg, clcuda::Queue q, unsigned int s)
    : queue(q.GetContext(), q.GetDevice()),
      dev_edges(q.GetContext(), MaxEdges(cfg)),
      dev_nodes(q.GetContext(), MaxNodes(cfg)),
      seed(s),
      // async: one stream per sample carries its mini-batch kernels and its neighbour sampler in order
      neighbor_sampler(cfg, cfg.async_launch ? queue : clcuda::Queue(q.GetContext(), q.GetDevice())) {
  // zeros: whatever a kernel reads from a not yet sampled buffer is a valid vertex id
  clcuda::Check(hipMemset(dev_edges.data(), 0, dev_edges.Count() * sizeof(Edge)), "hipMemset");
  clcuda::Check(hipMemset(dev_nodes.data(), 0, dev_nodes.Count() * sizeof(Vertex)), "hipMemset");
}

Sample::Sample(const Config& cfg, clcuda::Queue q) : Sample(cfg, q, static_cast<unsigned int>(rand())) {}

bool Sample::Serialize(std::ostream* out) {
  SampleStorage storage;
  storage.edges.assign(reinterpret_cast<const char*>(edges.data()), edges.size() * sizeof(Edge));
  storage.nodes_vec.assign(reinterpret_cast<const char*>(nodes_vec.data()), nodes_vec.size() * sizeof(Vertex));
  storage.seed = seed;
  return SerializeMessage(out, storage) && ::mcmc::Serialize(out, &dev_edges, &queue) &&
         ::mcmc::Serialize(out, &dev_nodes, &queue) && neighbor_sampler.Serialize(out);
}

bool Sample::Parse(std::istream* in) {
  SampleStorage storage;
  if (!(ParseMessage(in, &storage) && ::mcmc::Parse(in, &dev_edges, &queue) && ::mcmc::Parse(in, &dev_nodes, &queue) &&
        neighbor_sampler.Parse(in)))
    return false;
  edges.resize(storage.edges.size() / sizeof(Edge));
  memcpy(edges.data(), storage.edges.data(), edges.size() * sizeof(Edge));
  nodes_vec.resize(storage.nodes_vec.size() / sizeof(Vertex));
  memcpy(nodes_vec.data(), storage.nodes_vec.data(), nodes_vec.size() * sizeof(Vertex));
  seed = storage.seed;
  num_edges = static_cast<uint32_t>(edges.size());
  num_nodes = static_cast<uint32_t>(nodes_vec.size());
  return true;
}

// ------------------------------------------------------------------------------ PhiUpdater

namespace {
struct EventTimer {  // device time of one enqueue, like clcuda::Event::GetElapsedTime()
  hipEvent_t a, b;
  hipStream_t s;
  explicit EventTimer(void* stream) : s(static_cast<hipStream_t>(stream)) {
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    (void)hipEventRecord(a, s);
  }
  uint64_t StopNs() {
    (void)hipEventRecord(b, s);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    return static_cast<uint64_t>(ms * 1e6);
  }
};
}  // namespace

// ---------------------------------------------------------------------------- DeferredTimer

DeferredTimer::~DeferredTimer() {
  for (Rec& r : pending_) {
    (void)hipEventDestroy(static_cast<hipEvent_t>(r.start));
    (void)hipEventDestroy(static_cast<hipEvent_t>(r.stop));
  }
  for (void* e : free_) (void)hipEventDestroy(static_cast<hipEvent_t>(e));
  if (cur_) (void)hipEventDestroy(static_cast<hipEvent_t>(cur_));
}

void* DeferredTimer::Take() {
  if (!free_.empty()) {
    void* e = free_.back();
    free_.pop_back();
    return e;
  }
  hipEvent_t e;
  clcuda::Check(hipEventCreate(&e), "hipEventCreate");
  return e;
}

void DeferredTimer::Start(void* stream) {
  if (pending_.size() >= 4096) Resolve(false);  // a very long Run(): keep the pool bounded
  if (!cur_) cur_ = Take();
  clcuda::Check(hipEventRecord(static_cast<hipEvent_t>(cur_), static_cast<hipStream_t>(stream)), "hipEventRecord");
}

void DeferredTimer::Stop(void* stream, uint64_t* acc) {
  void* stop = Take();
  clcuda::Check(hipEventRecord(static_cast<hipEvent_t>(stop), static_cast<hipStream_t>(stream)), "hipEventRecord");
  pending_.push_back(Rec{cur_, stop, acc});
  cur_ = nullptr;
}

void DeferredTimer::Resolve(bool all) {
  const size_t n = all ? pending_.size() : pending_.size() / 2;
  if (n == 0) return;
  clcuda::Check(hipEventSynchronize(static_cast<hipEvent_t>(pending_[n - 1].stop)), "hipEventSynchronize");
  for (size_t i = 0; i < n; ++i) {
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, static_cast<hipEvent_t>(pending_[i].start), static_cast<hipEvent_t>(pending_[i].stop)) ==
        hipSuccess)
      *pending_[i].acc += static_cast<uint64_t>(static_cast<double>(ms) * 1.0e6);
    free_.push_back(pending_[i].start);
    free_.push_back(pending_[i].stop);
  }
  pending_.erase(pending_.begin(), pending_.begin() + static_cast<std::ptrdiff_t>(n));
}

// What the reference's kernel-variant switches mean here (phi.cc:608-700, main.cc:71-76).  The three work-group modes
// run the same arithmetic on the same lane <-> column <-> stream map (phi.cc:214-275 / :318-368 / :420-606 differ only
// in where pi_a / probs / grads live), so they all map onto the one HIP kernel family.  Two settings change RESULTS in
// the reference and are not reproduced: PHI_NODE_PER_THREAD draws a node's noise from ONE stream per node
// (phi.cc:124-152, rand_ sized max_nodes * 1) -- refused; phi_vector_width > 1 hands lane l the column vectors
// l, l + wg, ... of K / width (Floatn ownership, v_accn partial sums, `width` draws per vector: phi.cc:214-275) --
// computed as width 1, with a warning.  sum_grads_vector_width only widens the loads of sum_grads (beta.cc:39-49:
// the same per-column additions in the same order), which has no numerical effect to reproduce.
static const Config& CheckedPhiConfig(const Config& cfg) {
  if (cfg.phi_mode == PHI_NODE_PER_THREAD)
    throw std::invalid_argument(
        "phi_mode PHI_NODE_PER_THREAD is the reference's CPU-device kernel (one RNG stream per node, phi.cc:124-152): "
        "no MI355X form; use a PHI_NODE_PER_WORKGROUP_* mode");
  if (cfg.phi_mode != PHI_NODE_PER_WORKGROUP_NAIVE)
    std::cerr << "W phi_mode " << to_string(cfg.phi_mode) << ": runs the PHI_NODE_PER_WORKGROUP_NAIVE kernel family "
              << "(same arithmetic, lane-to-column map and RNG streams; the modes differ only in operand placement)"
              << std::endl;
  if (cfg.phi_probs_shared != true || cfg.phi_grads_shared != true || cfg.phi_pi_shared != true)
    std::cerr << "W phi-probs-shared / phi-grads-shared / phi-pi-shared: operand placement is the library's choice "
              << "(registers / LDS ring); results do not depend on it" << std::endl;
  if (cfg.phi_vector_width != 1)
    std::cerr << "W phi_vector_width " << cfg.phi_vector_width << ": NOT reproduced -- results follow phi_vector_width 1 "
              << "(lane l owns columns l, l + wg, ...; one draw per column in ascending order).  The reference's width-"
              << cfg.phi_vector_width << " kernels give lane l the column vectors l, l + wg, ... of K / width and sum "
              << "a vector's components before the lane partial (phi.cc:214-275), which changes WG_SUM's association "
              << "and the stream-to-column map -- and draw a vector's noise as VLn(randn) = (Floatn)(randn, ..., randn) "
              << "(types.cc:327-328), whose evaluation order OpenCL C leaves unspecified: the reference's own width-"
              << cfg.phi_vector_width << " results are implementation-defined" << std::endl;
  if (cfg.sum_grads_vector_width != 1)
    std::cerr << "W sum_grads_vector_width " << cfg.sum_grads_vector_width << ": accepted; sum_grads' vector width only "
              << "widens its loads (beta.cc:39-49), the sums are the same" << std::endl;
  return cfg;
}

PhiUpdater::PhiUpdater(const Config& cfg, clcuda::Queue queue, clcuda::Buffer<Float>& beta,
                       RowPartitionedMatrix<Float>* pi, clcuda::Buffer<Float>& phi, OpenClSet* trainingSet,
                       const std::vector<std::string>&, const std::string&)
    : ctx_(AcquireContext(CheckedPhiConfig(cfg), queue)),
      queue_(queue),
      beta_(beta),
      pi_(pi),
      phi_(phi),
      trainingSet_(trainingSet),
      max_nodes_(MaxNodes(cfg)),
      phi_vec_(queue.GetContext(), PhiVecRows(cfg) * cfg.K),
      rand_(queue, max_nodes_ * cfg.phi_wg_size, cfg.phi_seed),  // phi.cc:625-629
      count_calls_(0),
      local_(cfg.phi_wg_size),
      flags_(cfg.phi_disable_noise ? AMMSB_NOISE_OFF : 0u),
      async_(cfg.async_launch),
      timed_(cfg.async_launch && cfg.loop_timers && !cfg.graph_launch),
      t_update_phi_(0),
      t_update_pi_(0) {}

void PhiUpdater::operator()(clcuda::Buffer<Vertex>& nodes, clcuda::Buffer<Vertex>& neighbors, uint32_t n) {
  if (n == 0) throw std::runtime_error("mini-batch nodes size = 0!");  // phi.cc:732
  if (n > max_nodes_) throw std::runtime_error("grads too small");     // phi.cc:734-737
  ++count_calls_;
  if (async_) {
    if (timed_) timers_.Start(queue_.stream());
    ThrowIfError(ctx_.get(),
                 ammsb_update_phi(ctx_.get(), beta_.data(), &pi_->Get(), phi_.data(), &trainingSet_->Get(), nodes.data(),
                                  neighbors.data(), n, count_calls_, rand_.Get(), local_, flags_, 0, 0xFFFFFFFFu,
                                  phi_vec_.data(), queue_.stream()),
                 "ammsb_update_phi");
    if (timed_) {
      timers_.Stop(queue_.stream(), &t_update_phi_);
      timers_.Start(queue_.stream());
    }
    ThrowIfError(ctx_.get(),
                 ammsb_update_pi(ctx_.get(), &pi_->Get(), phi_.data(), phi_vec_.data(), nodes.data(), n, local_,
                                 queue_.stream()),
                 "ammsb_update_pi");
    if (timed_) timers_.Stop(queue_.stream(), &t_update_pi_);
    return;
  }
  {
    EventTimer t(queue_.stream());
    ThrowIfError(ctx_.get(),
                 ammsb_update_phi(ctx_.get(), beta_.data(), &pi_->Get(), phi_.data(), &trainingSet_->Get(), nodes.data(),
                                  neighbors.data(), n, count_calls_, rand_.Get(), local_, flags_, 0, 0xFFFFFFFFu,
                                  phi_vec_.data(), queue_.stream()),
                 "ammsb_update_phi");
    t_update_phi_ += t.StopNs();  // phi.cc:755-757 (launch, Finish, elapsed)
  }
  {
    EventTimer t(queue_.stream());
    ThrowIfError(ctx_.get(),
                 ammsb_update_pi(ctx_.get(), &pi_->Get(), phi_.data(), phi_vec_.data(), nodes.data(), n, local_,
                                 queue_.stream()),
                 "ammsb_update_pi");
    t_update_pi_ += t.StopNs();
  }
}

void PhiUpdater::UpdatePhi(clcuda::Buffer<Vertex>& nodes, clcuda::Buffer<Vertex>& neighbors, uint32_t n,
                           uint32_t group_begin, uint32_t group_end) {
  if (n == 0) throw std::runtime_error("mini-batch nodes size = 0!");  // phi.cc:732
  if (n > max_nodes_) throw std::runtime_error("grads too small");     // phi.cc:734-737
  ThrowIfError(ctx_.get(),
               ammsb_update_phi(ctx_.get(), beta_.data(), &pi_->Get(), phi_.data(), &trainingSet_->Get(), nodes.data(),
                                neighbors.data(), n, count_calls_, rand_.Get(), local_, flags_, group_begin, group_end,
                                phi_vec_.data(), queue_.stream()),
               "ammsb_update_phi");
}

void PhiUpdater::UpdatePi(clcuda::Buffer<Vertex>& nodes, uint32_t n) {
  ThrowIfError(ctx_.get(),
               ammsb_update_pi(ctx_.get(), &pi_->Get(), phi_.data(), phi_vec_.data(), nodes.data(), n, local_,
                               queue_.stream()),
               "ammsb_update_pi");
}

bool PhiUpdater::Serialize(std::ostream* out) {
  PhiProperties props;
  props.count_calls = count_calls_;
  props.update_phi_time = static_cast<double>(t_update_phi_);
  props.update_pi_time = static_cast<double>(t_update_pi_);
  return rand_.Serialize(out) && SerializeMessage(out, props);
}

bool PhiUpdater::Parse(std::istream* in) {
  PhiProperties props;
  if (!(rand_.Parse(in) && ParseMessage(in, &props))) return false;
  count_calls_ = props.count_calls;
  t_update_phi_ = static_cast<uint64_t>(props.update_phi_time);
  t_update_pi_ = static_cast<uint64_t>(props.update_pi_time);
  return true;
}

// ----------------------------------------------------------------------------- BetaUpdater

// EDGE_PER_THREAD is what the reference picks on a CPU device (learner.cc:105-114): the same terms added in another
// order (beta.cc:87-136, perplexity.cc:16-84).  There is one (work-group) kernel family here; say so instead of
// silently returning work-group-ordered sums to a caller who asked for the other mode.
static void WarnPerThreadMode(const char* who, bool per_thread) {
  if (per_thread)
    std::cerr << "W " << who << ": EDGE_PER_THREAD requested; the MI355X build runs the EDGE_PER_WORKGROUP form (same "
              << "terms, work-group summation order)" << std::endl;
}

BetaUpdater::BetaUpdater(Mode mode, const Config& cfg, clcuda::Queue queue, clcuda::Buffer<Float>& theta,
                         clcuda::Buffer<Float>& beta, RowPartitionedMatrix<Float>* pi, OpenClSet* trainingSet,
                         const std::vector<std::string>&, const std::string&)
    : ctx_((WarnPerThreadMode("BetaUpdater", mode == EDGE_PER_THREAD), AcquireContext(cfg, queue))),
      queue_(queue),
      theta_(theta),
      beta_(beta),
      pi_(pi),
      trainingSet_(trainingSet),
      rand_(queue, cfg.K, cfg.beta_seed),  // beta.cc:251-252
      count_calls_(0),
      local_(cfg.beta_wg_size),
      grads_(queue.GetContext(), 2 * cfg.K),
      theta_sum_(queue.GetContext(), cfg.K) {
  async_ = cfg.async_launch;
  timed_ = cfg.async_launch && cfg.loop_timers && !cfg.graph_launch;
}

clcuda::Buffer<Float>& BetaUpdater::GetThetaSum() {
  ThrowIfError(ctx_.get(), ammsb_theta_sum(ctx_.get(), theta_sum_.data(), queue_.stream()), "ammsb_theta_sum");
  queue_.Finish();
  return theta_sum_;
}

void BetaUpdater::operator()(clcuda::Buffer<Edge>* edges, uint32_t num_edges, Float scale) {
  ++count_calls_;  // beta.cc:336
  if (async_) {
    if (timed_) timers_.Start(queue_.stream());
    ThrowIfError(ctx_.get(),
                 ammsb_beta_grads(ctx_.get(), theta_.data(), beta_.data(), &pi_->Get(), &trainingSet_->Get(), edges->data(),
                                  num_edges, 0, num_edges, local_, grads_.data(), queue_.stream()),
                 "ammsb_beta_grads");
    if (timed_) {
      timers_.Stop(queue_.stream(), &t_grads_);
      timers_.Start(queue_.stream());
    }
    ThrowIfError(ctx_.get(),
                 ammsb_update_theta(ctx_.get(), theta_.data(), beta_.data(), grads_.data(), count_calls_, scale,
                                    rand_.Get(), 0, queue_.stream()),
                 "ammsb_update_theta");
    if (timed_) timers_.Stop(queue_.stream(), &t_update_theta_);
    return;
  }
  EventTimer tg(queue_.stream());
  ThrowIfError(ctx_.get(),
               ammsb_beta_grads(ctx_.get(), theta_.data(), beta_.data(), &pi_->Get(), &trainingSet_->Get(), edges->data(),
                                num_edges, 0, num_edges, local_, grads_.data(), queue_.stream()),
               "ammsb_beta_grads");
  t_grads_ += tg.StopNs();
  EventTimer tu(queue_.stream());
  ThrowIfError(ctx_.get(),
               ammsb_update_theta(ctx_.get(), theta_.data(), beta_.data(), grads_.data(), count_calls_, scale, rand_.Get(),
                                  0, queue_.stream()),
               "ammsb_update_theta");
  t_update_theta_ += tu.StopNs();
  queue_.Finish();
}

void BetaUpdater::CalculateGrads(clcuda::Buffer<Edge>* edges, uint32_t num_edges, uint32_t edge_begin, uint32_t edge_end,
                                 Float* out) {
  ThrowIfError(ctx_.get(),
               ammsb_beta_grads(ctx_.get(), theta_.data(), beta_.data(), &pi_->Get(), &trainingSet_->Get(), edges->data(),
                                num_edges, edge_begin, edge_end, local_, out, queue_.stream()),
               "ammsb_beta_grads");
}

bool BetaUpdater::CanFuseUpdatePi(uint32_t phi_local) const {
  return ammsb_can_fuse_pi_beta(ctx_.get(), phi_local, local_) != 0;
}

void BetaUpdater::UpdatePiAndGrads(clcuda::Buffer<Float>& phi_sum, clcuda::Buffer<Float>& phi_vec,
                                   clcuda::Buffer<Vertex>& nodes, clcuda::Buffer<Edge>* edges, uint32_t num_edges,
                                   Float* out) {
  ThrowIfError(ctx_.get(),
               ammsb_update_pi_beta_grads(ctx_.get(), theta_.data(), beta_.data(), &pi_->Get(), phi_sum.data(),
                                          phi_vec.data(), nodes.data(), &trainingSet_->Get(), edges->data(), num_edges,
                                          local_, out, queue_.stream()),
               "ammsb_update_pi_beta_grads");
}

void BetaUpdater::UpdateTheta(Float scale, const Float* grads) {
  ThrowIfError(ctx_.get(),
               ammsb_update_theta(ctx_.get(), theta_.data(), beta_.data(), grads, count_calls_, scale, rand_.Get(), 0,
                                  queue_.stream()),
               "ammsb_update_theta");
}

bool BetaUpdater::Serialize(std::ostream* out) {
  // theta_sum_ (beta.cc:20-28) is recomputed by every launch; the record is produced from theta
  std::vector<Float> theta(theta_.Count());
  theta_.Read(queue_, theta.size(), theta.data());
  VectorStorage sum;
  sum.storage.resize(theta.size() / 2 * sizeof(Float));
  for (size_t k = 0; k + 1 < theta.size(); k += 2) {
    const Float ts = theta[k] + theta[k + 1];
    memcpy(&sum.storage[k / 2 * sizeof(Float)], &ts, sizeof(Float));
  }
  BetaProperties props;
  props.count_calls = count_calls_;
  props.grads_partial_time = static_cast<double>(t_grads_);
  props.update_theta_time = static_cast<double>(t_update_theta_);
  return rand_.Serialize(out) && SerializeMessage(out, sum) && SerializeMessage(out, props);
}

bool BetaUpdater::Parse(std::istream* in) {
  VectorStorage sum;
  BetaProperties props;
  if (!(rand_.Parse(in) && ParseMessage(in, &sum) && ParseMessage(in, &props))) return false;
  if (sum.storage.size() != theta_.Count() / 2 * sizeof(Float)) return false;
  count_calls_ = props.count_calls;
  t_grads_ = static_cast<uint64_t>(props.grads_partial_time);
  t_update_theta_ = static_cast<uint64_t>(props.update_theta_time);
  return true;
}

// -------------------------------------------------------------------- PerplexityCalculator

PerplexityCalculator::PerplexityCalculator(Mode mode, const Config& cfg, clcuda::Queue queue, clcuda::Buffer<Float>& beta,
                                           RowPartitionedMatrix<Float>* pi, clcuda::Buffer<Edge>& edges,
                                           OpenClSet* edgeSet, const std::vector<std::string>&, const std::string&)
    : ctx_((WarnPerThreadMode("PerplexityCalculator", mode == EDGE_PER_THREAD), AcquireContext(cfg, queue))),
      queue_(queue),
      beta_(beta),
      pi_(pi),
      edges_(edges),
      edgeSet_(edgeSet),
      ppx_per_edge_(queue.GetContext(), edges.Count()),
      sums_(queue.GetContext(), 1),
      count_calls_(0),
      local_(cfg.ppx_wg_size) {
  // operator()() has its 32-byte result written straight into pinned host memory (one launch + one Finish, no copy)
  ammsb_ppx_sums* hs = nullptr;
  if (hipHostMalloc(reinterpret_cast<void**>(&hs), sizeof(ammsb_ppx_sums), hipHostMallocDefault) != hipSuccess || !hs)
    throw std::runtime_error("PerplexityCalculator: hipHostMalloc failed");
  std::memset(hs, 0, sizeof *hs);
  host_sums_.reset(hs, [](ammsb_ppx_sums* p) { (void)hipHostFree(p); });
  std::vector<Float> zero(edges.Count(), 0);  // perplexity.cc:204-205
  ppx_per_edge_.Write(queue_, zero.size(), zero.data());
}

Float PerplexityCalculator::operator()() {
  ++count_calls_;  // perplexity.cc:252
  const uint32_t H = static_cast<uint32_t>(edges_.Count());
  EventTimer t(queue_.stream());
  ThrowIfError(ctx_.get(),
               ammsb_perplexity(ctx_.get(), beta_.data(), &pi_->Get(), &edgeSet_->Get(), edges_.data(), H, 0, H,
                                count_calls_, local_, ppx_per_edge_.data(), host_sums_.get(), queue_.stream()),
               "ammsb_perplexity");
  t_ppx_ += t.StopNs();  // (waits for the launch: the sums are in host memory)
  const ammsb_ppx_sums s = *host_sums_;
  double avg = 0.0;  // perplexity.cc:264-268
  if (s.link_cnt + s.nonlink_cnt != 0) avg = (s.link_ll + s.nonlink_ll) / static_cast<double>(s.link_cnt + s.nonlink_cnt);
  return static_cast<Float>(-avg);
}

ammsb_ppx_sums* PerplexityCalculator::Partial(uint32_t edge_begin, uint32_t edge_end) {
  const uint32_t H = static_cast<uint32_t>(edges_.Count());
  ThrowIfError(ctx_.get(),
               ammsb_perplexity(ctx_.get(), beta_.data(), &pi_->Get(), &edgeSet_->Get(), edges_.data(), H, edge_begin,
                                edge_end, count_calls_, local_, ppx_per_edge_.data(), sums_.data(), queue_.stream()),
               "ammsb_perplexity");
  return sums_.data();
}

bool PerplexityCalculator::Serialize(std::ostream* out) {
  PerplexityProperties props;
  props.count_calls = count_calls_;
  props.ppx_time = static_cast<double>(t_ppx_);
  return SerializeMessage(out, props) && ::mcmc::Serialize(out, &ppx_per_edge_, &queue_);
}

bool PerplexityCalculator::Parse(std::istream* in) {
  PerplexityProperties props;
  if (!ParseMessage(in, &props)) return false;
  count_calls_ = props.count_calls;
  t_ppx_ = static_cast<uint64_t>(props.ppx_time);
  return ::mcmc::Parse(in, &ppx_per_edge_, &queue_);
}

}  // namespace mcmc
