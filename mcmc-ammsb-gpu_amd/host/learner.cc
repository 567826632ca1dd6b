// mcmc::Learner (include/mcmc/learner.h): the reference's orchestration (mcmc/learner.cc:77-299) on
// top of the C-ABI-backed operators.  Same allocation order, same initialisation (theta_0 from
// std::mt19937(6342455113) + std::gamma_distribution, pi_0 from the device gamma streams {11,113}),
// same loop: join the sample produced in the background, start the next one, phi, pi, beta.
#include "mcmc/learner.h"
#include "mcmc/exchange.h"
#include "mcmc/serialize.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <functional>
#include <iostream>
#include <random>
#include <sstream>
#include <stdexcept>
#include <string>
#include <tuple>

using namespace std::chrono;

namespace mcmc {

namespace {
clcuda::Buffer<Edge> Upload(const clcuda::Queue& q, const std::vector<Edge>& v) {
  return clcuda::Buffer<Edge>(q.GetContext(), q, v.begin(), v.end());
}
}  // namespace

std::vector<Edge> MakeEdgesForTrainingPerplexity(const Config& cfg) {
  const uint64_t total = (cfg.N * (cfg.N - 1)) / 2;
  const uint64_t num_links = static_cast<uint64_t>(cfg.training_ppx_ratio * cfg.training_edges.size());
  const uint64_t num_non_links = static_cast<uint64_t>(num_links * total / static_cast<double>(cfg.E));
  if (num_links + num_non_links >= (1ull << 31))
    throw std::runtime_error("training perplexity: " + std::to_string(num_links + num_non_links) +
                             " edges (links * N(N-1)/2 / E non-links, learner.cc:50-53) do not fit one launch; lower "
                             "training_ppx_ratio");
  std::vector<Edge> ret(num_links + num_non_links);
  std::copy(cfg.training_edges.begin(), cfg.training_edges.begin() + num_links, ret.begin());
  unsigned int seed = cfg.training_ppx_seed;
  for (uint64_t i = num_links; i < ret.size(); ++i) {
    Vertex u, v;
    Edge e;
    do {
      u = static_cast<Vertex>(rand_r(&seed) % cfg.N);
      do {
        v = static_cast<Vertex>(rand_r(&seed) % cfg.N);
      } while (u == v);
      e = MakeEdge(u, v);
    } while (cfg.training->Has(e) || (cfg.heldout && cfg.heldout->Has(e)));
    ret[i] = e;
  }
  return ret;
}

Learner::Learner(const Config& cfg, clcuda::Queue queue)
    : cfg_(cfg),
      queue_(queue),
      beta_(queue_.GetContext(), 2 * cfg_.K),
      theta_(queue_.GetContext(), 2 * cfg_.K),
      allocFactory_(RowPartitionedMatrixFactory<Float>::New(queue_)),
      pi_(allocFactory_->CreateMatrix(static_cast<uint32_t>(cfg_.N), static_cast<uint32_t>(cfg_.K))),
      phi_(queue_.GetContext(), cfg_.N),
      setFactory_(OpenClSetFactory::New(queue_)),
      trainingSet_(setFactory_->CreateSet(*cfg_.training)),
      heldoutSet_(setFactory_->CreateSet(*cfg_.heldout)),
      heldoutEdges_(Upload(queue_, cfg_.heldout_edges)),
      heldoutPerplexity_(PerplexityCalculator::EDGE_PER_WORKGROUP, cfg_, queue_, beta_, pi_.get(), heldoutEdges_,
                         heldoutSet_.get()),
      phiUpdater_(cfg_, queue_, beta_, pi_.get(), phi_, trainingSet_.get()),
      betaUpdater_(BetaUpdater::EDGE_PER_WORKGROUP, cfg_, queue_, theta_, beta_, pi_.get(), trainingSet_.get()),
      sampler_(GetSampler(cfg_.strategy)),
      stepCount_(1),
      time_(0),
      samplingTime_(0),
      edges_done_(0),
      phase_(0) {
  if (!sampler_) throw std::runtime_error("Unkown sample strategy");  // learner.cc:146
  if (cfg_.calc_train_ppx) {  // learner.cc:92-104
    trainingPerplexityEdges_ = MakeEdgesForTrainingPerplexity(cfg_);
    if (trainingPerplexityEdges_.empty()) throw std::runtime_error("training perplexity: no edges (raise training_ppx_ratio)");
    devTrainingPerplexityEdges_.reset(new clcuda::Buffer<Edge>(queue_.GetContext(), queue_, trainingPerplexityEdges_.begin(),
                                                               trainingPerplexityEdges_.end()));
    trainingPerplexity_.reset(new PerplexityCalculator(PerplexityCalculator::EDGE_PER_WORKGROUP, cfg_, queue_, beta_,
                                                       pi_.get(), *devTrainingPerplexityEdges_, trainingSet_.get()));
  }
  samples_[0].reset(new Sample(cfg_, queue_, cfg_.sample_seed[0]));
  samples_[1].reset(new Sample(cfg_, queue_, cfg_.sample_seed[1]));
  if (cfg_.device_sampling) {
    if (cfg_.strategy != Node && cfg_.strategy != NodeLink && cfg_.strategy != NodeNonLink)
      throw std::runtime_error("device sampling implements Node / NodeLink / NodeNonLink only");
    ctx_ = AcquireContext(cfg_, queue_);
    std::vector<uint64_t> off;
    std::vector<Vertex> tgt;
    cfg_.trainingGraph->ExportCSR(&off, &tgt);
    if (tgt.empty()) throw std::runtime_error("training graph has no edges");
    degree_.resize(cfg_.N);
    for (uint64_t u = 0; u < cfg_.N; ++u) degree_[u] = static_cast<uint32_t>(off[u + 1] - off[u]);
    csr_offsets_.reset(new clcuda::Buffer<uint64_t>(queue_.GetContext(), queue_, off.begin(), off.end()));
    csr_targets_.reset(new clcuda::Buffer<Vertex>(queue_.GetContext(), queue_, tgt.begin(), tgt.end()));
    // invalid non-link partners of u: u itself and its neighbours in the training and held-out graphs
    // (sample.cc:283-285); the candidate draws of a mini-batch follow its vertex, the capacity the largest
    excluded_.assign(cfg_.N, 1);
    uint32_t max_excluded = 1;
    for (uint64_t u = 0; u < cfg_.N; ++u) excluded_[u] += degree_[u];
    if (cfg_.heldout)
      for (Edge e : cfg_.heldout_edges)
        if (cfg_.heldout->Has(e)) {
          Vertex a, b;
          std::tie(a, b) = Vertices(e);
          ++excluded_[a];
          ++excluded_[b];
        }
    for (uint64_t u = 0; u < cfg_.N; ++u) max_excluded = std::max(max_excluded, excluded_[u]);
    candidates_ = CandidatesForExcluded(max_excluded);
    if (candidates_ == 0)
      throw std::runtime_error("device sampling needs N >= 2 * mini_batch with room for the largest degree");
    mb_rand_.reset(new random::OpenClRandom(queue_, candidates_, cfg_.device_sampling_seed));
    // scrambled states: the reference's {s+i, s'+i} layout makes the streams' first draws collide far too often
    ThrowIfError(ctx_.get(),
                 ammsb_rng_init_mixed(ctx_.get(), mb_rand_->Get(), candidates_, cfg_.device_sampling_seed[0],
                                      cfg_.device_sampling_seed[1], queue_.stream()),
                 "ammsb_rng_init_mixed");
    queue_.Finish();
    mb_workspace_.reset(new clcuda::Buffer<uint8_t>(queue_.GetContext(), ammsb_minibatch_workspace_bytes(candidates_)));
    // 0xFF once: every call leaves the de-duplication table empty again (include/ammsb.h)
    clcuda::Check(hipMemset(mb_workspace_->data(), 0xFF, ammsb_minibatch_workspace_bytes(candidates_)), "hipMemset");
    mb_count_.reset(new clcuda::Buffer<uint32_t>(queue_.GetContext(), 2));  // [0] last count, [1] sticky shortfalls
    const uint32_t zero2[2] = {0, 0};
    mb_count_->Write(queue_, 2, zero2);
    host_rng_.seed(cfg_.device_sampling_host_seed);
  }
  if (cfg_.async_launch) {
    if (!cfg_.device_sampling) throw std::runtime_error("async_launch needs device_sampling (the host samplers block)");
    for (int i = 0; i < 2; ++i) {
      hipEvent_t a, b;
      clcuda::Check(hipEventCreateWithFlags(&a, hipEventDisableTiming), "hipEventCreate");
      clcuda::Check(hipEventCreateWithFlags(&b, hipEventDisableTiming), "hipEventCreate");
      ev_ready_[i] = a;
      ev_consumed_[i] = b;
    }
    hipEvent_t e;
    clcuda::Check(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate");
    ev_sampler_ = e;
  }
  if (Sharded()) {
    if (cfg_.graph_launch) throw std::runtime_error("graph_launch is single-rank (the exchange is not captured)");
    if (!ctx_) ctx_ = AcquireContext(cfg_, queue_);
    const uint32_t R = static_cast<uint32_t>(cfg_.exchange->world());
    SetSplit(cfg_.phi_replicate > 0 ? static_cast<uint32_t>(std::min<double>(cfg_.phi_replicate, 1.0) * AMMSB_MAX_GROUPS + 0.5) : 0u);
    {
      hipStream_t xs;
      clcuda::Check(hipStreamCreateWithFlags(&xs, hipStreamNonBlocking), "hipStreamCreate");
      xstream_ = xs;
      hipEvent_t a, b;
      clcuda::Check(hipEventCreateWithFlags(&a, hipEventDisableTiming), "hipEventCreate");
      clcuda::Check(hipEventCreateWithFlags(&b, hipEventDisableTiming), "hipEventCreate");
      ev_block_ = a;
      ev_xdone_ = b;
    }
    const uint64_t max_nodes = samples_[0]->dev_nodes.Count();
    all_grads_.reset(new clcuda::Buffer<Float>(queue_.GetContext(), static_cast<uint64_t>(R) * 2 * cfg_.K));
    grads_sum_.reset(new clcuda::Buffer<Float>(queue_.GetContext(), 2 * cfg_.K));
    tail_buf_.reset(new clcuda::Buffer<Float>(
        queue_.GetContext(), std::max<uint64_t>(max_nodes > AMMSB_MAX_GROUPS ? max_nodes - AMMSB_MAX_GROUPS : 0, 1) * cfg_.K));
    all_sums_.reset(new clcuda::Buffer<ammsb_ppx_sums>(queue_.GetContext(), R));
  }
  std::mt19937 mt19937(6342455113);  // learner.cc:150-153
  std::gamma_distribution<Float> gamma_distribution(cfg_.eta0, cfg_.eta1);
  auto gamma = std::bind(gamma_distribution, mt19937);
  random::RandomAndNormalize(&queue_, gamma, &theta_, &beta_, 2);
  random::RandomGammaAndNormalize(&queue_, cfg_.eta0, cfg_.eta1, pi_.get(), &phi_);  // learner.cc:154-155
  PlacePi();
  if (Sharded() && cfg_.phi_replicate < 0) CalibrateSplit();
  if (cfg_.graph_launch) {
    if (!cfg_.async_launch || !cfg_.device_sampling)
      throw std::runtime_error("graph_launch needs async_launch and device_sampling");
    ammsb_loop_config lc;
    std::memset(&lc, 0, sizeof lc);
    lc.theta = theta_.data();
    lc.beta = beta_.data();
    lc.pi = &pi_->Get();
    lc.phi_sum = phi_.data();
    lc.training_set = &trainingSet_->Get();
    lc.heldout_set = heldoutSet_ ? &heldoutSet_->Get() : nullptr;
    lc.phi_seeds = phiUpdater_.Rand().Get();
    lc.phi_vec = phiUpdater_.GetPhiVec().data();
    lc.phi_wg = phiUpdater_.Local();
    lc.phi_flags = phiUpdater_.Flags();
    lc.beta_seeds = betaUpdater_.Rand().Get();
    lc.grads = betaUpdater_.GetGrads().data();
    lc.beta_wg = betaUpdater_.Local();
    lc.beta_flags = 0;
    for (int i = 0; i < 2; ++i) {
      lc.edges[i] = samples_[i]->dev_edges.data();
      lc.nodes[i] = samples_[i]->dev_nodes.data();
      lc.neighbors[i] = samples_[i]->neighbor_sampler.GetData().data();
      lc.nbr_table[i] = samples_[i]->neighbor_sampler.GetHash().data();
      lc.nbr_seeds[i] = samples_[i]->neighbor_sampler.Rand().Get();
    }
    lc.nbr_wg = samples_[0]->neighbor_sampler.Local();
    lc.csr_offsets = csr_offsets_->data();
    lc.csr_targets = csr_targets_->data();
    lc.mb_seeds = mb_rand_->Get();
    lc.mb_candidates = candidates_;
    lc.mb_workspace = mb_workspace_->data();
    lc.mb_count = mb_count_->data();
    lc.mini_batch = static_cast<uint32_t>(cfg_.mini_batch_size);
    lc.max_fan_out = static_cast<uint32_t>(cfg_.trainingGraph->MaxFanOut());
    lc.max_nodes = static_cast<uint32_t>(samples_[0]->dev_nodes.Count());
    lc.max_edges = static_cast<uint32_t>(samples_[0]->dev_edges.Count());
    lc.flags = cfg_.loop_timers ? AMMSB_LOOP_TIMESTAMPS : 0u;
    queue_.Finish();
    ThrowIfError(ctx_.get(), ammsb_loop_create(ctx_.get(), &lc, &loop_), "ammsb_loop_create");
  }
}

uint32_t Learner::CandidatesForExcluded(uint32_t excluded) {
  const uint32_t key = (excluded + 31u) / 32u * 32u;  // few distinct values
  auto it = cand_cache_.find(key);
  if (it != cand_cache_.end()) return it->second;
  const uint32_t c = ammsb_minibatch_candidates_for(cfg_.N, static_cast<uint32_t>(cfg_.mini_batch_size), key);
  cand_cache_[key] = c;
  return c;
}

uint32_t Learner::CandidatesFor(uint64_t u) { return CandidatesForExcluded(excluded_[u]); }

// A non-link mini-batch that found fewer than m distinct partners repeats entries to stay memory-safe; it is not a
// valid sample.  The device counts them (sticky); this reads the counter at a synchronisation point.
void Learner::CheckDeviceSampler() {
  if (!cfg_.device_sampling) return;
  if (loop_) {
    uint32_t timeouts = 0;
    ThrowIfError(ctx_.get(), ammsb_loop_check(loop_, &timeouts), "ammsb_loop_check");
    if (timeouts)
      throw std::runtime_error("graph loop: " + std::to_string(timeouts) +
                               " device-side wait(s) timed out (set AMMSB_LOOP_HANDSHAKE=event under kernel-serialising tools)");
  }
  uint32_t cnt[2] = {0, 0};
  mb_count_->Read(queue_, 2, cnt);
  if (cnt[1] != 0) {
    const uint32_t zero2[2] = {0, 0};
    mb_count_->Write(queue_, 2, zero2);
    throw std::runtime_error("device mini-batch sampler: " + std::to_string(cnt[1]) +
                             " mini-batch(es) found fewer than mini_batch_size distinct non-links");
  }
}

Learner::~Learner() {
  for (auto& f : futures_)
    if (f.valid()) f.wait();
  if (loop_) ammsb_loop_destroy(loop_);
  if (cfg_.async_launch) {
    (void)hipDeviceSynchronize();
    for (int i = 0; i < 2; ++i) {
      if (ev_ready_[i]) (void)hipEventDestroy(static_cast<hipEvent_t>(ev_ready_[i]));
      if (ev_consumed_[i]) (void)hipEventDestroy(static_cast<hipEvent_t>(ev_consumed_[i]));
    }
    if (ev_sampler_) (void)hipEventDestroy(static_cast<hipEvent_t>(ev_sampler_));
  }
  if (xstream_) {
    (void)hipDeviceSynchronize();
    (void)hipStreamDestroy(static_cast<hipStream_t>(xstream_));
    if (ev_block_) (void)hipEventDestroy(static_cast<hipEvent_t>(ev_block_));
    if (ev_xdone_) (void)hipEventDestroy(static_cast<hipEvent_t>(ev_xdone_));
  }
}

// Device-side replacement for sampleNode + ExtractNodesFromMiniBatch (sample.cc:249-303, learner.cc:162-173):
// the coin flip and the choice of u stay on the host, so the sizes are known without a read-back; the m
// distinct non-links / the edges of u are produced on the device straight into the sample's buffers.
ammsb_mb_choice Learner::ChooseDevice() {
  bool link = cfg_.strategy == NodeLink;
  if (cfg_.strategy == Node) link = (host_rng_() & 1u) != 0;  // rand_r(seed) % 2, sample.cc:297
  const uint64_t N = cfg_.N;
  ammsb_mb_choice ch = {0, 0, 0, 0};
  if (link) {
    uint64_t u;
    do {  // sampleNodeLink retries until the vertex has an edge, sample.cc:254-263
      u = host_rng_() % N;
    } while (degree_[u] == 0);
    ch.link = 1;
    ch.u = static_cast<uint32_t>(u);
    ch.n = degree_[u];
    return ch;
  }
  ch.u = static_cast<uint32_t>(host_rng_() % N);
  ch.n_candidates = CandidatesFor(ch.u);
  return ch;
}

Float Learner::EnqueueDevice(Sample* sample, const ammsb_mb_choice& ch) {
  const uint64_t N = cfg_.N;
  void* stream = sample->queue.stream();
  // async: the candidate streams, workspace and counter are shared by the two samples' streams
  if (cfg_.async_launch && sampler_valid_)
    clcuda::Check(hipStreamWaitEvent(static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(ev_sampler_), 0),
                  "hipStreamWaitEvent");
  const uint32_t m = static_cast<uint32_t>(cfg_.mini_batch_size);
  if (ch.link)
    ThrowIfError(ctx_.get(),
                 ammsb_minibatch_link(ctx_.get(), csr_offsets_->data(), csr_targets_->data(), ch.u, ch.n,
                                      sample->dev_edges.data(), sample->dev_nodes.data(), stream),
                 "ammsb_minibatch_link");
  else
    ThrowIfError(ctx_.get(),
                 ammsb_minibatch_nonlink(ctx_.get(), mb_rand_->Get(), ch.n_candidates, candidates_, ch.u, m,
                                         &trainingSet_->Get(), heldoutSet_ ? &heldoutSet_->Get() : nullptr,
                                         mb_workspace_->data(), sample->dev_edges.data(), sample->dev_nodes.data(),
                                         mb_count_->data(), stream),
                 "ammsb_minibatch_nonlink");
  if (cfg_.async_launch) {
    clcuda::Check(hipEventRecord(static_cast<hipEvent_t>(ev_sampler_), static_cast<hipStream_t>(stream)), "hipEventRecord");
    sampler_valid_ = true;
  } else {
    sample->queue.Finish();  // the neighbour sampler runs on its own queue
  }
  sample->num_edges = ch.link ? ch.n : m;
  sample->num_nodes = sample->num_edges + 1;
  sample->neighbor_sampler(sample->num_nodes, &sample->dev_nodes);
  return ch.link ? static_cast<Float>(N)                                        // sample.cc:268
                 : static_cast<Float>(2 * cfg_.E) / static_cast<Float>(m);      // sample.cc:292
}

Float Learner::DoSampleDevice(Sample* sample) {
  const int idx = sample == samples_[0].get() ? 0 : 1;
  choice_[idx] = ChooseDevice();
  return EnqueueDevice(sample, choice_[idx]);
}

Float Learner::DoSample(Sample* sample) {
  if (cfg_.device_sampling) return DoSampleDevice(sample);
  sample->edges.clear();
  const Float weight = sampler_(cfg_, &sample->edges, &sample->seed);
  ExtractNodesFromMiniBatch(sample->edges, &sample->nodes_vec);
  if (sample->nodes_vec.empty()) throw std::runtime_error("mini-batch size = 0!");
  if (sample->edges.size() > sample->dev_edges.Count() || sample->nodes_vec.size() > sample->dev_nodes.Count())
    throw std::runtime_error("mini-batch larger than its device buffers");  // learner.cc:184-189
  sample->dev_edges.Write(sample->queue, sample->edges.size(), sample->edges.data());
  sample->dev_nodes.Write(sample->queue, sample->nodes_vec.size(), sample->nodes_vec.data());
  sample->num_edges = static_cast<uint32_t>(sample->edges.size());
  sample->num_nodes = static_cast<uint32_t>(sample->nodes_vec.size());
  sample->neighbor_sampler(sample->num_nodes, &sample->dev_nodes);
  return weight;
}

bool Learner::Sharded() const { return cfg_.exchange && cfg_.exchange->world() > 1; }

void Learner::Step(Sample& s, Float weight) {
  if (Sharded()) return StepSharded(s, weight);
  phiUpdater_(s.dev_nodes, s.neighbor_sampler.GetData(), s.num_nodes);
  betaUpdater_(&s.dev_edges, s.num_edges, weight);
}

// Where pi lands in HBM (mcmc-ammsb-gpu_amd/learner.py, _place_pi): update_phi gathers 4 KiB rows of pi at random, and
// the same launch over allocations of pi made one after the other in one process differs by up to 10 %, stably per
// allocation.  Once pi is initialised: it is copied into a few more allocations, full-size update_phi launches are
// timed over each (warm clocks first, then round-robin), the fastest is kept, the rest released.  Every candidate holds
// the same pi, the streams and the call counter are restored: results do not depend on it.
void Learner::PlacePi() {
  uint32_t want = cfg_.pi_placement_candidates;
  if (const char* e = getenv("AMMSB_PI_CANDIDATES")) want = static_cast<uint32_t>(std::max(0, atoi(e)));
  const uint64_t bytes = static_cast<uint64_t>(cfg_.N) * cfg_.K * sizeof(Float);
  if (want < 2 || pi_->Blocks().size() != 1 || bytes < (1ull << 30)) return;
  size_t free_b = 0, total_b = 0;
  clcuda::Check(hipMemGetInfo(&free_b, &total_b), "hipMemGetInfo");
  want = std::min<uint32_t>(want, 1u + static_cast<uint32_t>(free_b / 3 / bytes));
  if (want < 2) return;
  hipStream_t stream = static_cast<hipStream_t>(queue_.stream());
  const uint32_t K = static_cast<uint32_t>(cfg_.K), nn = static_cast<uint32_t>(cfg_.num_node_sample);
  Sample& s = *samples_[0];
  const uint32_t n = static_cast<uint32_t>(std::min<uint64_t>(s.dev_nodes.Count(), AMMSB_MAX_GROUPS));
  std::mt19937_64 gen(7);
  std::vector<Vertex> nodes(n), nbrs(static_cast<size_t>(n) * nn);
  for (Vertex& v : nodes) v = static_cast<Vertex>(gen() % cfg_.N);
  for (Vertex& v : nbrs) v = static_cast<Vertex>(gen() % cfg_.N);
  // (the sample's buffers are borrowed: what they held is put back, so that a checkpoint does not depend on this)
  std::vector<Vertex> nodes_was(n), nbrs_was(nbrs.size());
  s.dev_nodes.Read(queue_, n, nodes_was.data());
  s.neighbor_sampler.GetData().Read(queue_, nbrs_was.size(), nbrs_was.data());
  s.dev_nodes.Write(queue_, n, nodes.data());
  s.neighbor_sampler.GetData().Write(queue_, nbrs.size(), nbrs.data());
  std::vector<ammsb_seed> keep(phiUpdater_.Rand().GetSeeds().Count());
  phiUpdater_.Rand().GetSeeds().Read(queue_, keep.size(), keep.data());
  const uint32_t calls = phiUpdater_.CountCalls();
  phiUpdater_.CountCalls() = 1;
  // candidate 0 is pi itself; a candidate is tried by exchanging its storage with pi's (the operators hold pi)
  std::vector<std::unique_ptr<RowPartitionedMatrix<Float>>> cands;
  for (uint32_t c = 1; c < want; ++c)
    cands.emplace_back(allocFactory_->CreateMatrix(static_cast<uint32_t>(cfg_.N), K));
  for (auto& c : cands)  // the initialised pi: whichever candidate is kept holds it
    clcuda::Check(hipMemcpyAsync(c->Blocks()[0].data(), pi_->Blocks()[0].data(), bytes, hipMemcpyDeviceToDevice, stream),
                  "hipMemcpyAsync");
  auto three = [&] {
    for (int i = 0; i < 3; ++i) phiUpdater_.UpdatePhi(s.dev_nodes, s.neighbor_sampler.GetData(), n, 0, 0xFFFFFFFFu);
  };
  auto with = [&](uint32_t c, const std::function<void()>& fn) {  // run fn with candidate c in pi's place
    if (c) pi_->SwapStorage(*cands[c - 1]);
    fn();
    if (c) pi_->SwapStorage(*cands[c - 1]);
  };
  const auto t_warm = std::chrono::steady_clock::now();
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_warm).count() < 0.4) {
    for (uint32_t c = 0; c < want; ++c) with(c, three);
    queue_.Finish();
  }
  std::vector<std::vector<double>> ms(want);
  hipEvent_t a, b;
  clcuda::Check(hipEventCreate(&a), "hipEventCreate");
  clcuda::Check(hipEventCreate(&b), "hipEventCreate");
  for (int rnd = 0; rnd < 3; ++rnd)
    for (uint32_t c = 0; c < want; ++c) {
      queue_.Finish();
      with(c, [&] {
        clcuda::Check(hipEventRecord(a, stream), "hipEventRecord");
        three();
        clcuda::Check(hipEventRecord(b, stream), "hipEventRecord");
      });
      clcuda::Check(hipEventSynchronize(b), "hipEventSynchronize");
      float t = 0;
      clcuda::Check(hipEventElapsedTime(&t, a, b), "hipEventElapsedTime");
      ms[c].push_back(t / 3.0);
    }
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  phiUpdater_.Rand().GetSeeds().Write(queue_, keep.size(), keep.data());
  phiUpdater_.CountCalls() = calls;
  uint32_t best = 0;
  std::vector<double> med(want);
  for (uint32_t c = 0; c < want; ++c) {
    std::sort(ms[c].begin(), ms[c].end());
    med[c] = ms[c][ms[c].size() / 2];
    if (med[c] < med[best]) best = c;
  }
  if (best) pi_->SwapStorage(*cands[best - 1]);
  queue_.Finish();
  s.dev_nodes.Write(queue_, n, nodes_was.data());
  s.neighbor_sampler.GetData().Write(queue_, nbrs_was.size(), nbrs_was.data());
  cands.clear();  // releases every allocation but the kept one
  std::cerr << "I pi placement: update_phi over " << want << " allocations of pi:";
  for (double v : med) std::cerr << " " << v;
  std::cerr << " ms -> kept #" << best << std::endl;
}

// Ownership map (as in mcmc-ammsb-gpu_amd/learner.py, _set_split): groups [0, g_rep) replicated, the rest in
// world * nch blocks of cc groups; a rank's block of a chunk stays near or above one chip-load of one-wave nodes.
void Learner::SetSplit(uint32_t g_rep) {
  const uint32_t R = static_cast<uint32_t>(cfg_.exchange->world());
  g_rep_ = std::min<uint32_t>(g_rep, AMMSB_MAX_GROUPS);
  const uint32_t own = (AMMSB_MAX_GROUPS - g_rep_) / R;
  nch_ = std::max<uint32_t>(1u, std::min<uint32_t>(std::max<uint32_t>(1u, cfg_.phi_chunks), own / 3072u));
  const uint32_t blocks = R * nch_;
  cc_ = std::max<uint32_t>(1u, (AMMSB_MAX_GROUPS - g_rep_ + blocks - 1) / blocks);
}

// Choose g_rep so that computing (replicated + own) groups takes as long as receiving the others' rows: T = one
// update_phi over a full synthetic mini-batch, X = one all-gather of a full phi_vec; a fraction rho replicated costs
// T (rho + (1 - rho) / R) of compute and X (1 - rho) of exchange.  The phi streams are restored, pi is not touched;
// every rank adopts rank 0's answer.  (learner.py, _calibrate_split; the choice of exchange form stays with
// AMMSB_EXCHANGE_FORM here.)
void Learner::CalibrateSplit() {
  Exchange& x = *cfg_.exchange;
  const uint32_t R = static_cast<uint32_t>(x.world());
  const uint32_t K = static_cast<uint32_t>(cfg_.K), nn = static_cast<uint32_t>(cfg_.num_node_sample);
  hipStream_t stream = static_cast<hipStream_t>(queue_.stream());
  Sample& s = *samples_[0];
  const uint32_t n = static_cast<uint32_t>(std::min<uint64_t>(s.dev_nodes.Count(), AMMSB_MAX_GROUPS));
  std::mt19937_64 gen(1);
  std::vector<Vertex> nodes(n), nbrs(static_cast<size_t>(n) * nn);
  for (Vertex& v : nodes) v = static_cast<Vertex>(gen() % cfg_.N);
  for (Vertex& v : nbrs) v = static_cast<Vertex>(gen() % cfg_.N);
  std::vector<Vertex> nodes_was(n), nbrs_was(nbrs.size());  // (borrowed buffers: put back below)
  s.dev_nodes.Read(queue_, n, nodes_was.data());
  s.neighbor_sampler.GetData().Read(queue_, nbrs_was.size(), nbrs_was.data());
  s.dev_nodes.Write(queue_, n, nodes.data());
  s.neighbor_sampler.GetData().Write(queue_, nbrs.size(), nbrs.data());
  std::vector<ammsb_seed> keep(phiUpdater_.Rand().GetSeeds().Count());
  phiUpdater_.Rand().GetSeeds().Read(queue_, keep.size(), keep.data());
  const uint32_t calls = phiUpdater_.CountCalls();
  phiUpdater_.CountCalls() = 1;
  auto timed = [&](const std::function<void()>& fn) {
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
      hipEvent_t a, b;
      clcuda::Check(hipEventCreate(&a), "hipEventCreate");
      clcuda::Check(hipEventCreate(&b), "hipEventCreate");
      queue_.Finish();
      x.Barrier();
      clcuda::Check(hipEventRecord(a, stream), "hipEventRecord");
      fn();
      clcuda::Check(hipEventRecord(b, stream), "hipEventRecord");
      clcuda::Check(hipEventSynchronize(b), "hipEventSynchronize");
      float ms = 0;
      clcuda::Check(hipEventElapsedTime(&ms, a, b), "hipEventElapsedTime");
      best = std::min<double>(best, ms);
      (void)hipEventDestroy(a);
      (void)hipEventDestroy(b);
    }
    return best;
  };
  const double T = timed([&] { phiUpdater_.UpdatePhi(s.dev_nodes, s.neighbor_sampler.GetData(), n, 0, 0xFFFFFFFFu); });
  const uint32_t per = (n + R - 1) / R;
  const double X = timed([&] { x.AllGatherInPlace(phiUpdater_.GetPhiVec().data(), static_cast<size_t>(per) * K * sizeof(Float), stream); }) *
                   (static_cast<double>(n) / (static_cast<double>(per) * R));
  phiUpdater_.Rand().GetSeeds().Write(queue_, keep.size(), keep.data());
  phiUpdater_.CountCalls() = calls;
  auto cost = [&](double r) { return std::max(T * (r + (1.0 - r) / R), X * (1.0 - r)); };
  double rho = X > T / R ? (X - T / R) / (T - T / R + X) : 0.0;
  rho = std::min(std::max(rho, 0.0), 1.0);
  if (cost(1.0) <= cost(rho) * 1.02) rho = 1.0;  // links so slow that exchanging anything loses
  x.BroadcastHost(&rho, sizeof rho, 0);           // one answer for the whole job: rank 0's
  calib_phi_ms_ = T;
  calib_xchg_ms_ = X;
  SetSplit(static_cast<uint32_t>(rho * AMMSB_MAX_GROUPS));
  queue_.Finish();
  s.dev_nodes.Write(queue_, n, nodes_was.data());
  s.neighbor_sampler.GetData().Write(queue_, nbrs_was.size(), nbrs_was.data());
  if (x.rank() == 0)
    std::cerr << "I exchange split: update_phi " << T << " ms, full exchange " << X << " ms -> " << g_rep_ << " of "
              << AMMSB_MAX_GROUPS << " groups replicated, " << nch_ << " chunk(s) of " << cc_ << " groups per rank"
              << std::endl;
}

// One iteration over `world` ranks (the schedule of learner.py's _phi_sharded).  Ownership is fixed -- groups
// [0, g_rep) run on every rank, block b of cc groups behind them on rank b % world -- because a group's RNG streams
// advance only where the group runs; phi_vec row i < G belongs to group i, row G + t to group t (its second node).
// Per chunk: update_phi over the own block on the main stream, then that chunk's in-place all-gather (region
// [base, base + world * cc), which can reach past G: owners park their tail rows first and hand them out afterwards)
// on the exchange stream, beside the next block and the replicated groups.  A chunk whose live groups all sit in rank
// 0's block (a link mini-batch of a low-degree vertex) is a broadcast.  pi is bit-identical to a single rank's; so is
// theta when every rank computes the whole gradient (Config::beta_grads), else it differs by the association of the
// gradient sum (per-rank partials summed in rank order).
void Learner::StepSharded(Sample& s, Float weight) {
  Exchange& x = *cfg_.exchange;
  const uint32_t R = static_cast<uint32_t>(x.world()), r = static_cast<uint32_t>(x.rank()), Cc = cc_, g0 = g_rep_;
  const uint32_t n = s.num_nodes, K = static_cast<uint32_t>(cfg_.K);
  hipStream_t stream = static_cast<hipStream_t>(queue_.stream()), xs = static_cast<hipStream_t>(xstream_);
  clcuda::Buffer<Vertex>& nbrs = s.neighbor_sampler.GetData();
  Float* pv = phiUpdater_.GetPhiVec().data();
  Float* tb = tail_buf_->data();
  const size_t row = static_cast<size_t>(K) * sizeof(Float);
  if (n == 0) throw std::runtime_error("mini-batch nodes size = 0!");  // phi.cc:732
  phiUpdater_.BeginCall();
  const uint32_t G = std::min<uint32_t>(n, AMMSB_MAX_GROUPS), tail = n - G;
  const uint32_t rep_hi = std::min(g0, G), rep_tail = std::min(tail, rep_hi);
  // replicated groups that own tail rows go first: the last all-gather region can reach past row G, and every sender
  // must already hold the final value of whatever it sends from there
  if (rep_tail > 0) phiUpdater_.UpdatePhi(s.dev_nodes, nbrs, n, 0, rep_tail);
  bool exchanged = false;
  if (G > g0) {
    const uint32_t live = (G - g0 + R * Cc - 1) / (R * Cc);  // chunks with at least one live exchanged group
    for (uint32_t c = 0; c < live; ++c) {
      const uint32_t base = g0 + c * R * Cc, lo = base + r * Cc, hi = std::min(lo + Cc, G);
      if (lo < hi) phiUpdater_.UpdatePhi(s.dev_nodes, nbrs, n, lo, hi);
      if (c == live - 1 && tail > g0)  // exchanged groups with a tail row: owners park theirs before the region is overwritten
        for (uint32_t b0 = g0; b0 < tail; b0 += Cc)
          if (((b0 - g0) / Cc) % R == r)
            clcuda::Check(hipMemcpyAsync(tb + static_cast<size_t>(b0) * K, pv + static_cast<size_t>(G + b0) * K,
                                         (std::min(b0 + Cc, tail) - b0) * row, hipMemcpyDeviceToDevice, stream),
                          "hipMemcpyAsync");
      clcuda::Check(hipEventRecord(static_cast<hipEvent_t>(ev_block_), stream), "hipEventRecord");
      clcuda::Check(hipStreamWaitEvent(xs, static_cast<hipEvent_t>(ev_block_), 0), "hipStreamWaitEvent");
      if (G - base <= Cc)
        x.Broadcast(pv + static_cast<size_t>(base) * K, (G - base) * row, 0, xs);
      else
        x.AllGatherInPlace(pv + static_cast<size_t>(base) * K, Cc * row, xs);
      exchanged = true;
    }
  }
  if (rep_hi > rep_tail) phiUpdater_.UpdatePhi(s.dev_nodes, nbrs, n, rep_tail, rep_hi);  // overlaps the exchanges in flight
  if (exchanged) {
    clcuda::Check(hipEventRecord(static_cast<hipEvent_t>(ev_xdone_), xs), "hipEventRecord");
    clcuda::Check(hipStreamWaitEvent(stream, static_cast<hipEvent_t>(ev_xdone_), 0), "hipStreamWaitEvent");
  }
  if (tail > g0) {
    for (uint32_t b0 = g0; b0 < tail; b0 += Cc)  // owners hand out their parked tail rows
      x.Broadcast(tb + static_cast<size_t>(b0) * K, (std::min(b0 + Cc, tail) - b0) * row, static_cast<int>(((b0 - g0) / Cc) % R), stream);
    clcuda::Check(hipMemcpyAsync(pv + static_cast<size_t>(G + g0) * K, tb + static_cast<size_t>(g0) * K, (tail - g0) * row,
                                 hipMemcpyDeviceToDevice, stream),
                  "hipMemcpyAsync");
  }
  // The gradient (mcmc-ammsb-gpu_amd/learner.py, _run): cut over the ranks only in "sharded" mode and only for
  // mini-batches worth a collective; otherwise every rank computes all of it -- with update_pi folded into the same
  // launch where the shape and the mini-batch (edge t = (nodes[0], nodes[t + 1]): the device sampler's) allow.
  const uint32_t ne = s.num_edges, per = (ne + R - 1) / R;
  const bool fusable = cfg_.device_sampling && betaUpdater_.CanFuseUpdatePi(phiUpdater_.Local());
  const bool replicated = cfg_.beta_grads == 1 || (cfg_.beta_grads < 0 && fusable);
  const bool shard = !replicated && ne > cfg_.beta_shard_min_edges;
  const bool fuse = !shard && fusable && n == ne + 1;
  if (!fuse) phiUpdater_.UpdatePi(s.dev_nodes, n);
  betaUpdater_.BeginCall();
  Float* local = betaUpdater_.GetGrads().data();
  if (shard) {
    betaUpdater_.CalculateGrads(&s.dev_edges, ne, std::min(r * per, ne), std::min((r + 1) * per, ne), local);
    x.AllGather(local, all_grads_->data(), 2 * row, stream);
    ThrowIfError(ctx_.get(), ammsb_sum_rows_f32(ctx_.get(), all_grads_->data(), R, 2 * K, grads_sum_->data(), stream),
                 "ammsb_sum_rows_f32");
    betaUpdater_.UpdateTheta(weight, grads_sum_->data());
  } else {
    if (fuse) betaUpdater_.UpdatePiAndGrads(phi_, phiUpdater_.GetPhiVec(), s.dev_nodes, &s.dev_edges, ne, local);
    else betaUpdater_.CalculateGrads(&s.dev_edges, ne, 0, ne, local);
    betaUpdater_.UpdateTheta(weight, local);
  }
  if (!cfg_.async_launch) queue_.Finish();
}

Float Learner::Perplexity(PerplexityCalculator* calc) {
  if (!Sharded()) return (*calc)();
  Exchange& x = *cfg_.exchange;
  const uint32_t R = static_cast<uint32_t>(x.world()), r = static_cast<uint32_t>(x.rank());
  calc->BeginCall();
  const uint32_t H = calc->NumEdges(), per = (H + R - 1) / R;
  ammsb_ppx_sums* mine = calc->Partial(std::min(r * per, H), std::min((r + 1) * per, H));
  x.AllGather(mine, all_sums_->data(), sizeof(ammsb_ppx_sums), queue_.stream());
  std::vector<ammsb_ppx_sums> parts(R);
  all_sums_->Read(queue_, R, parts.data());
  ammsb_ppx_sums t = {0.0, 0.0, 0, 0};
  for (const ammsb_ppx_sums& p : parts) {  // rank order: the same value on every rank
    t.link_ll += p.link_ll;
    t.nonlink_ll += p.nonlink_ll;
    t.link_cnt += p.link_cnt;
    t.nonlink_cnt += p.nonlink_cnt;
  }
  double avg = 0.0;  // perplexity.cc:264-268
  if (t.link_cnt + t.nonlink_cnt != 0) avg = (t.link_ll + t.nonlink_ll) / static_cast<double>(t.link_cnt + t.nonlink_cnt);
  return static_cast<Float>(-avg);
}

// Two pieces of state advance only at their owner: the phi streams of each rank's group block and the running-mean
// perplexity of each rank's edge slice.  A collective: every rank calls Serialize, after which any rank's stream is
// the checkpoint (and restarts with any world size).
void Learner::GatherShardedState() {
  if (!Sharded()) return;
  Exchange& x = *cfg_.exchange;
  const uint32_t R = static_cast<uint32_t>(x.world());
  DrainAsync();
  queue_.Finish();
  const uint64_t L = phiUpdater_.Local(), count = phiUpdater_.Rand().GetSeeds().Count();
  ammsb_seed* seeds = phiUpdater_.Rand().Get();
  for (uint32_t b = 0; b < R * nch_; ++b) {  // block b of cc_ groups behind the replicated prefix belongs to rank b % R
    const uint64_t glo = static_cast<uint64_t>(g_rep_) + static_cast<uint64_t>(b) * cc_;
    const uint64_t lo = glo * L, hi = std::min<uint64_t>(std::min<uint64_t>(glo + cc_, AMMSB_MAX_GROUPS) * L, count);
    if (lo >= hi) break;
    x.Broadcast(seeds + lo, (hi - lo) * sizeof(ammsb_seed), static_cast<int>(b % R), queue_.stream());
  }
  std::vector<PerplexityCalculator*> calcs;
  if (trainingPerplexity_) calcs.push_back(trainingPerplexity_.get());
  calcs.push_back(&heldoutPerplexity_);
  for (PerplexityCalculator* c : calcs) {
    const uint32_t H = c->NumEdges(), per = (H + R - 1) / R;
    for (uint32_t q = 0; q < R; ++q) {
      const uint32_t lo = std::min(q * per, H), hi = std::min((q + 1) * per, H);
      if (lo < hi) x.Broadcast(c->PerEdge().data() + lo, (hi - lo) * sizeof(Float), static_cast<int>(q), queue_.stream());
    }
  }
  queue_.Finish();
}

Float Learner::TrainingPerplexity() {  // learner.cc:204-212
  if (!trainingPerplexity_) throw std::runtime_error("TrainingPerplexity() needs Config::calc_train_ppx");
  const auto t1 = high_resolution_clock::now();
  const Float ppx = Perplexity(trainingPerplexity_.get());
  time_ += duration_cast<nanoseconds>(high_resolution_clock::now() - t1).count();
  return std::exp(ppx);
}

Float Learner::HeldoutPerplexity() {
  const auto t1 = high_resolution_clock::now();
  const Float ppx = Perplexity(&heldoutPerplexity_);
  time_ += duration_cast<nanoseconds>(high_resolution_clock::now() - t1).count();
  return std::exp(ppx);
}

// Enqueue-only loop (Config::async_launch, device sampling): the next mini-batch is produced on the other sample's
// stream while the main stream runs phi / pi / beta; `ready` and `consumed` events order the two, the host never
// waits inside the loop.  Same launches in the same per-stream order as the synchronous loop => identical results.
void Learner::RunAsync(uint32_t max_iters, sig_atomic_t* signaled) {
  hipStream_t main = static_cast<hipStream_t>(queue_.stream());
  auto enqueue_sample = [&](int idx) {
    Sample* s = samples_[idx].get();
    hipStream_t st = static_cast<hipStream_t>(s->queue.stream());
    if (consumed_valid_[idx])  // do not overwrite buffers a running iteration still reads
      clcuda::Check(hipStreamWaitEvent(st, static_cast<hipEvent_t>(ev_consumed_[idx]), 0), "hipStreamWaitEvent");
    weights_[idx] = DoSampleDevice(s);
    clcuda::Check(hipEventRecord(static_cast<hipEvent_t>(ev_ready_[idx]), st), "hipEventRecord");
    enqueued_[idx] = true;
  };
  if (!enqueued_[phase_]) enqueue_sample(phase_);
  for (uint64_t i = 0; i < max_iters && (signaled ? !*signaled : true); ++i, ++stepCount_) {
    const Float weight = weights_[phase_];
    enqueue_sample(1 - phase_);
    Sample& s = *samples_[phase_];
    clcuda::Check(hipStreamWaitEvent(main, static_cast<hipEvent_t>(ev_ready_[phase_]), 0), "hipStreamWaitEvent");
    Step(s, weight);
    clcuda::Check(hipEventRecord(static_cast<hipEvent_t>(ev_consumed_[phase_]), main), "hipEventRecord");
    consumed_valid_[phase_] = true;
    enqueued_[phase_] = false;
    edges_done_ += s.num_edges;
    phase_ = 1 - phase_;
  }
}

// The same iterations as RunAsync, enqueued as captured graphs (ammsb_loop): the host only chooses the mini-batches
// and hands them over; sizes, eps_t and weights travel in device descriptors.  Bit-identical to RunAsync.
void Learner::RunGraph(uint32_t max_iters, sig_atomic_t* signaled) {
  hipStream_t main = static_cast<hipStream_t>(queue_.stream());
  if (phiUpdater_.CountCalls() != betaUpdater_.CountCalls()) throw std::runtime_error("graph_launch: step counters differ");
  if (!enqueued_[phase_]) {  // the first mini-batch is sampled eagerly, as in RunAsync
    Sample* s = samples_[phase_].get();
    weights_[phase_] = DoSampleDevice(s);
    clcuda::Check(hipEventRecord(static_cast<hipEvent_t>(ev_ready_[phase_]), static_cast<hipStream_t>(s->queue.stream())),
                  "hipEventRecord");
    enqueued_[phase_] = true;
  }
  const uint32_t kChunk = 512;  // iterations between two looks at `signaled`
  std::vector<ammsb_mb_choice> next;
  uint32_t done = 0;
  while (done < max_iters && (signaled ? !*signaled : true)) {
    const uint32_t n = std::min(kChunk, max_iters - done);
    clcuda::Check(hipStreamWaitEvent(main, static_cast<hipEvent_t>(ev_ready_[phase_]), 0), "hipStreamWaitEvent");
    next.resize(n);
    for (uint32_t i = 0; i < n; ++i) next[i] = ChooseDevice();
    ThrowIfError(ctx_.get(),
                 ammsb_loop_run(loop_, &choice_[phase_], next.data(), n, phiUpdater_.CountCalls() + 1,
                                static_cast<uint32_t>(phase_), main),
                 "ammsb_loop_run");
    const uint32_t m = static_cast<uint32_t>(cfg_.mini_batch_size);
    edges_done_ += choice_[phase_].link ? choice_[phase_].n : m;
    for (uint32_t i = 0; i + 1 < n; ++i) edges_done_ += next[i].link ? next[i].n : m;
    phiUpdater_.CountCalls() += n;
    betaUpdater_.CountCalls() += n;
    stepCount_ += n;
    phase_ ^= static_cast<int>(n & 1u);
    const ammsb_mb_choice& last = next.back();
    choice_[phase_] = last;
    samples_[phase_]->num_edges = last.link ? last.n : m;
    samples_[phase_]->num_nodes = samples_[phase_]->num_edges + 1;
    weights_[phase_] = last.link ? static_cast<Float>(cfg_.N) : static_cast<Float>(2 * cfg_.E) / static_cast<Float>(m);
    enqueued_[phase_] = true;
    enqueued_[1 - phase_] = false;
    // everything queued on the main stream so far orders whatever the eager path does next
    clcuda::Check(hipEventRecord(static_cast<hipEvent_t>(ev_ready_[phase_]), main), "hipEventRecord");
    for (int i = 0; i < 2; ++i) {
      clcuda::Check(hipEventRecord(static_cast<hipEvent_t>(ev_consumed_[i]), main), "hipEventRecord");
      consumed_valid_[i] = true;
    }
    clcuda::Check(hipEventRecord(static_cast<hipEvent_t>(ev_sampler_), main), "hipEventRecord");
    sampler_valid_ = true;
    done += n;
    // every 64 chunks (32 768 iterations) the loop's record of what it has enqueued is released and a device-side
    // wait that gave up is noticed and resumed (ammsb_loop_check synchronises): a long Run() neither grows that
    // record without bound nor learns about a fallback only at its very end
    if (++chunks_since_check_ >= 64) {
      chunks_since_check_ = 0;
      CheckDeviceSampler();
    }
  }
}

void Learner::DrainAsync() {
  if (!cfg_.async_launch) return;
  for (auto& smp : samples_) smp->queue.Finish();
  queue_.Finish();
  phiUpdater_.ResolveTimers();  // event pairs of the enqueue-only launches (Config::loop_timers)
  betaUpdater_.ResolveTimers();
}

// The captured-graph loop's kernels note the device time they start at (include/ammsb.h, ammsb_loop_step_stamps):
// the differences are the categories of PrintStats.  The sum of the partial rows and the theta / beta step are one
// kernel here and are counted under UPDATE THETA; the time the main chain waited for the next mini-batch is SAMPLING
// (the reference's SAMPLING is the host's wait for the sampler thread, learner.cc:225-235).  The last 8192 steps of
// a longer call stand for all of them.
void Learner::AccountLoopStamps(uint32_t first_step, uint32_t n_steps) {
  if (!loop_ || !cfg_.loop_timers || n_steps == 0) return;
  const uint32_t kept = std::min<uint32_t>(n_steps, 8192);
  std::vector<double> st(static_cast<size_t>(kept) * AMMSB_LOOP_STAMP_SLOTS);
  ThrowIfError(ctx_.get(), ammsb_loop_step_stamps(loop_, first_step + (n_steps - kept), kept, st.data()),
               "ammsb_loop_step_stamps");
  double phi = 0, pi = 0, grads = 0, theta = 0, wait = 0;
  for (uint32_t i = 0; i < kept; ++i) {
    const double* s = &st[static_cast<size_t>(i) * AMMSB_LOOP_STAMP_SLOTS];
    if (!(s[0] > 0 && s[1] >= s[0] && s[2] >= s[1] && s[3] >= s[2] && s[4] >= s[3] && s[5] >= s[4])) continue;
    phi += s[1] - s[0];
    pi += s[2] - s[1];
    grads += s[3] - s[2];
    theta += s[4] - s[3];
    wait += s[5] - s[4];
  }
  const double scale = static_cast<double>(n_steps) / kept;
  phiUpdater_.AddTimes(static_cast<uint64_t>(phi * scale), static_cast<uint64_t>(pi * scale));
  betaUpdater_.AddTimes(static_cast<uint64_t>(grads * scale), static_cast<uint64_t>(theta * scale));
  samplingTime_ += static_cast<uint64_t>(wait * scale);
}

void Learner::Run(uint32_t max_iters, sig_atomic_t* signaled) {
  const auto t1 = high_resolution_clock::now();
  if (cfg_.async_launch) {
    const uint32_t first_step = phiUpdater_.CountCalls() + 1;
    if (loop_)
      RunGraph(max_iters, signaled);
    else
      RunAsync(max_iters, signaled);
    DrainAsync();  // Run() returns with the work done, like the reference's
    CheckDeviceSampler();
    if (loop_) AccountLoopStamps(first_step, phiUpdater_.CountCalls() + 1 - first_step);
    time_ += duration_cast<nanoseconds>(high_resolution_clock::now() - t1).count();
    return;
  }
  if (stepCount_ == 1 && !futures_[phase_].valid())
    futures_[phase_] = std::async(std::launch::async, &Learner::DoSample, this, samples_[phase_].get());
  for (uint64_t i = 0; i < max_iters && (signaled ? !*signaled : true); ++i, ++stepCount_) {
    const auto ts = high_resolution_clock::now();
    const Float weight = futures_[phase_].get();
    futures_[1 - phase_] = std::async(std::launch::async, &Learner::DoSample, this, samples_[1 - phase_].get());
    samplingTime_ += duration_cast<nanoseconds>(high_resolution_clock::now() - ts).count();
    Sample& s = *samples_[phase_];
    Step(s, weight);
    edges_done_ += s.num_edges;
    phase_ = 1 - phase_;
  }
  if (cfg_.device_sampling) {
    if (futures_[phase_].valid()) futures_[phase_].wait();  // the pending sample's kernels are enqueued and finished
    CheckDeviceSampler();
  }
  time_ += duration_cast<nanoseconds>(high_resolution_clock::now() - t1).count();
}

bool Learner::Serialize(std::ostream* out) {
  LearnerProperties props;
  props.stepCount = stepCount_;
  props.time = time_;
  props.samplingTime = samplingTime_;
  props.phase = phase_;
  Float weight;
  if (cfg_.async_launch) {
    if (!enqueued_[phase_]) {
      weights_[phase_] = DoSampleDevice(samples_[phase_].get());
      clcuda::Check(hipEventRecord(static_cast<hipEvent_t>(ev_ready_[phase_]),
                                   static_cast<hipStream_t>(samples_[phase_]->queue.stream())), "hipEventRecord");
      enqueued_[phase_] = true;
    }
    DrainAsync();
    weight = weights_[phase_];
  } else {
    if (!futures_[phase_].valid())  // the reference's constructor has the first sample in flight already
      futures_[phase_] = std::async(std::launch::async, &Learner::DoSample, this, samples_[phase_].get());
    weight = futures_[phase_].get();  // learner.cc:307-311: take the value, then re-arm the future
    futures_[phase_] = std::async(std::launch::deferred, [weight]() -> Float { return weight; });
  }
  props.weight = weight;
  queue_.Finish();
  GatherShardedState();
  return ::mcmc::Serialize(out, &beta_, &queue_) && ::mcmc::Serialize(out, &theta_, &queue_) &&
         ::mcmc::Serialize(out, pi_.get(), &queue_) && ::mcmc::Serialize(out, &phi_, &queue_) &&
         phiUpdater_.Serialize(out) && betaUpdater_.Serialize(out) &&
         (!trainingPerplexity_ || trainingPerplexity_->Serialize(out)) &&  // learner.cc:321-323
         heldoutPerplexity_.Serialize(out) &&
         SerializeMessage(out, props) && samples_[0]->Serialize(out) && samples_[1]->Serialize(out) &&
         SerializeDeviceSampler(out);
}

// Trailing extension (the reference's Parse stops after the two samples): the device sampler's host generator,
// the sizes of the mini-batches sitting in the sample buffers, and its candidate streams.
bool Learner::SerializeDeviceSampler(std::ostream* out) {
  if (!cfg_.device_sampling) return true;
  std::ostringstream st;
  st << host_rng_ << " " << samples_[0]->num_edges << " " << samples_[0]->num_nodes << " " << samples_[1]->num_edges
     << " " << samples_[1]->num_nodes << " " << edges_done_;
  for (const ammsb_mb_choice& ch : choice_) st << " " << ch.link << " " << ch.u << " " << ch.n << " " << ch.n_candidates;
  SampleStorage ext;  // reused as a two-bytes-field container: edges = magic, nodes_vec = text state
  ext.edges = "AMMSB-DEVSAMPLER-CPP-1";
  ext.nodes_vec = st.str();
  return SerializeMessage(out, ext) && mb_rand_->Serialize(out);
}

bool Learner::ParseDeviceSampler(std::istream* in) {
  if (!cfg_.device_sampling) return true;
  SampleStorage ext;
  if (!ParseMessage(in, &ext) || ext.edges != "AMMSB-DEVSAMPLER-CPP-1") return false;
  std::istringstream st(ext.nodes_vec);
  st >> host_rng_ >> samples_[0]->num_edges >> samples_[0]->num_nodes >> samples_[1]->num_edges >>
      samples_[1]->num_nodes >> edges_done_;
  for (ammsb_mb_choice& ch : choice_) st >> ch.link >> ch.u >> ch.n >> ch.n_candidates;
  return !st.fail() && mb_rand_->Parse(in);
}

bool Learner::Parse(std::istream* in) {
  for (auto& f : futures_)
    if (f.valid()) f.wait();
  LearnerProperties props;
  if (!(::mcmc::Parse(in, &beta_, &queue_) && ::mcmc::Parse(in, &theta_, &queue_) &&
        ::mcmc::Parse(in, pi_.get(), &queue_) && ::mcmc::Parse(in, &phi_, &queue_) && phiUpdater_.Parse(in) &&
        betaUpdater_.Parse(in) && (!trainingPerplexity_ || trainingPerplexity_->Parse(in)) &&  // learner.cc:339-341
        heldoutPerplexity_.Parse(in) && ParseMessage(in, &props)))
    return false;
  stepCount_ = props.stepCount;
  time_ = props.time;
  samplingTime_ = props.samplingTime;
  phase_ = props.phase & 1;
  if (!(samples_[0]->Parse(in) && samples_[1]->Parse(in) && ParseDeviceSampler(in))) return false;
  const Float weight = static_cast<Float>(props.weight);
  if (cfg_.async_launch) {
    DrainAsync();
    weights_[phase_] = weight;
    enqueued_[phase_] = true;   // the restored sample buffers hold the pending mini-batch
    enqueued_[1 - phase_] = false;
    consumed_valid_[0] = consumed_valid_[1] = false;
    clcuda::Check(hipEventRecord(static_cast<hipEvent_t>(ev_ready_[phase_]),
                                 static_cast<hipStream_t>(samples_[phase_]->queue.stream())), "hipEventRecord");
    return true;
  }
  futures_[1 - phase_] = std::future<Float>();
  futures_[phase_] = std::async(std::launch::deferred, [weight]() -> Float { return weight; });
  return true;
}

void Learner::PrintStats(std::ostream& out) {  // learner.cc:252-299: same categories, seconds and share of TOTAL
  const double total = time_ / 1.0e9;
  auto line = [&](const char* name, uint64_t ns) {
    out << name << ": " << ns / 1.0e9 << " (%" << (total > 0 ? 100 * (ns / 1.0e9) / total : 0.0) << ")\n";
  };
  out << "TOTAL    : " << total << "\n";
  line("PPX CALC ", heldoutPerplexity_.PerplexityTime());
  line("PPX ACCUM", heldoutPerplexity_.AccumulateTime());
  if (trainingPerplexity_) {  // learner.cc:262-271
    line("TRAIN PPX CALC ", trainingPerplexity_->PerplexityTime());
    line("TRAIN PPX ACCUM", trainingPerplexity_->AccumulateTime());
  }
  line("SAMPLING ", samplingTime_);
  line("PHI      ", phiUpdater_.UpdatePhiTime());
  line("PI       ", phiUpdater_.UpdatePiTime());
  line("THETA SUM   ", betaUpdater_.ThetaSumTime());
  line("GRADS PAR   ", betaUpdater_.GradsPartialTime());
  line("GRADS SUM   ", betaUpdater_.GradsSumTime());
  line("UPDATE THETA", betaUpdater_.UpdateThetaTime());
  line("NORM THETA  ", betaUpdater_.NormalizeTime());
  // not in the reference: the headline rate of this build's benchmark
  out << "MINI-BATCH EDGES: " << edges_done_ << " (" << (total > 0 ? edges_done_ / total : 0.0) << " edges/s)\n";
}

void Learner::PrintStats() { PrintStats(std::cerr); }

std::vector<Float> Learner::GetBeta() {
  std::vector<Float> v(2 * cfg_.K);
  beta_.Read(queue_, v.size(), v.data());
  return v;
}
std::vector<Float> Learner::GetTheta() {
  std::vector<Float> v(2 * cfg_.K);
  theta_.Read(queue_, v.size(), v.data());
  return v;
}
std::vector<Float> Learner::GetPiRow(Vertex v) {
  std::vector<Float> row(cfg_.K);
  const uint32_t blk = v / pi_->RowsPerBlock();
  pi_->Blocks()[blk].Read(queue_, cfg_.K, row.data(), (size_t)(v % pi_->RowsPerBlock()) * cfg_.K);
  return row;
}

}  // namespace mcmc
