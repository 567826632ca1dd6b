// extern "C" view of the host data structures (include/ammsb_host.h).
#include "ammsb_host.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <new>
#include <random>

#include "mcmc/config.h"
#include "mcmc/data.h"
#include "mcmc/learner.h"
#include "mcmc/sample.h"

struct ammsb_host_set {
  mcmc::Set* set;
  bool owned;
};

struct ammsb_host_dataset {
  mcmc::Config cfg;  // only the data half is populated
  ammsb_host_set training{nullptr, false}, heldout{nullptr, false};
};

namespace {
uint64_t* CopyOut(const std::vector<mcmc::Edge>& v) {
  uint64_t* p = static_cast<uint64_t*>(malloc(sizeof(uint64_t) * (v.size() ? v.size() : 1)));
  if (p && !v.empty()) memcpy(p, v.data(), sizeof(uint64_t) * v.size());
  return p;
}
}  // namespace

extern "C" {

ammsb_host_set* ammsb_host_set_create(const uint64_t* keys, uint64_t n) {
  std::unique_ptr<mcmc::Set> s(new (std::nothrow) mcmc::Set(n));
  if (!s || !s->SetContents(keys, keys + n)) return nullptr;
  return new ammsb_host_set{s.release(), true};
}

void ammsb_host_set_destroy(ammsb_host_set* s) {
  if (!s) return;
  if (s->owned) delete s->set;
  delete s;
}

uint64_t ammsb_host_set_bins(const ammsb_host_set* s) { return s->set->BinsPerBucket(); }
uint32_t ammsb_host_set_prime_idx(const ammsb_host_set* s) { return s->set->PrimeIdx(); }
uint64_t ammsb_host_set_size(const ammsb_host_set* s) { return s->set->Size(); }
const uint64_t* ammsb_host_set_data(const ammsb_host_set* s) { return s->set->Data(); }

int ammsb_host_set_has(const ammsb_host_set* s, const uint64_t* keys, uint64_t n, uint8_t* out) {
  if (!s || !keys || !out) return -1;
  for (uint64_t i = 0; i < n; ++i) out[i] = s->set->Has(keys[i]) ? 1 : 0;
  return 0;
}

int64_t ammsb_host_generate_graph(uint64_t N, uint32_t K_true, double avg_degree, uint64_t seed, uint64_t** edges) {
  if (!edges || N < 2 || K_true == 0) return -1;
  const std::vector<mcmc::Edge> e = mcmc::GenerateSyntheticGraph(N, K_true, avg_degree, seed);
  *edges = CopyOut(e);
  return *edges ? static_cast<int64_t>(e.size()) : -1;
}

void ammsb_host_free(void* p) { free(p); }

int64_t ammsb_host_load_snap(const char* path, uint64_t* N, uint64_t** edges) {
  std::vector<mcmc::Edge> e;
  if (!path || !N || !edges || !mcmc::GetUniqueEdgesFromFile(path, N, &e)) return -1;
  *edges = CopyOut(e);
  return *edges ? static_cast<int64_t>(e.size()) : -1;
}

int ammsb_host_dump_dataset(const char* path, uint64_t N, float heldout_ratio, const uint64_t* edges, uint64_t n) {
  if (!path || (!edges && n)) return -1;
  return mcmc::DumpDataset(path, N, heldout_ratio, std::vector<mcmc::Edge>(edges, edges + n)) ? 0 : -1;
}

int64_t ammsb_host_load_dataset(const char* path, uint64_t* N, float* heldout_ratio, uint64_t** edges) {
  std::vector<mcmc::Edge> e;
  if (!path || !N || !heldout_ratio || !edges || !mcmc::LoadDataset(path, N, heldout_ratio, &e)) return -1;
  *edges = CopyOut(e);
  return *edges ? static_cast<int64_t>(e.size()) : -1;
}

ammsb_host_dataset* ammsb_host_dataset_create(uint64_t N, const uint64_t* edges, uint64_t n, double heldout_ratio,
                                              unsigned rand_seed) {
  if (!edges || N < 2) return nullptr;
  std::unique_ptr<ammsb_host_dataset> d(new ammsb_host_dataset);
  const std::vector<mcmc::Edge> vals(edges, edges + n);
  srand(rand_seed);
  d->cfg.N = N;
  d->cfg.E = n;
  d->cfg.heldout_ratio = static_cast<mcmc::Float>(heldout_ratio);
  if (!mcmc::GenerateSetsFromEdges(N, vals, heldout_ratio, &d->cfg.training_edges, &d->cfg.heldout_edges,
                                   &d->cfg.training, &d->cfg.heldout))
    return nullptr;
  d->cfg.trainingGraph.reset(new mcmc::Graph(N, d->cfg.training_edges));
  d->cfg.heldoutGraph.reset(new mcmc::Graph(N, d->cfg.heldout_edges));
  d->training = {d->cfg.training.get(), false};
  d->heldout = {d->cfg.heldout.get(), false};
  return d.release();
}

void ammsb_host_dataset_destroy(ammsb_host_dataset* d) { delete d; }
uint64_t ammsb_host_dataset_num_training(const ammsb_host_dataset* d) { return d->cfg.training_edges.size(); }
uint64_t ammsb_host_dataset_num_heldout(const ammsb_host_dataset* d) { return d->cfg.heldout_edges.size(); }
const uint64_t* ammsb_host_dataset_training_edges(const ammsb_host_dataset* d) { return d->cfg.training_edges.data(); }
const uint64_t* ammsb_host_dataset_heldout_edges(const ammsb_host_dataset* d) { return d->cfg.heldout_edges.data(); }
const ammsb_host_set* ammsb_host_dataset_training_set(const ammsb_host_dataset* d) { return &d->training; }
const ammsb_host_set* ammsb_host_dataset_heldout_set(const ammsb_host_dataset* d) {
  return d->heldout.set ? &d->heldout : nullptr;
}
uint64_t ammsb_host_dataset_max_fan_out(const ammsb_host_dataset* d) { return d->cfg.trainingGraph->MaxFanOut(); }

int ammsb_host_dataset_training_csr(const ammsb_host_dataset* d, uint64_t* offsets, uint32_t* targets) {
  if (!d || !offsets || !targets) return -1;
  std::vector<uint64_t> off;
  std::vector<mcmc::Vertex> tgt;
  d->cfg.trainingGraph->ExportCSR(&off, &tgt);
  memcpy(offsets, off.data(), sizeof(uint64_t) * off.size());
  memcpy(targets, tgt.data(), sizeof(uint32_t) * tgt.size());
  return 0;
}

// theta_0 as Learner's constructor draws it (learner.cc:150-153): std::mt19937(6342455113) feeding
// std::gamma_distribution<float>(eta0, eta1), 2K consecutive draws.  libstdc++-defined stream.
int ammsb_host_theta_init(uint64_t K, float eta0, float eta1, float* theta_out) {
  if (!theta_out || K == 0) return -1;
  std::mt19937 engine(6342455113);
  std::gamma_distribution<mcmc::Float> dist(eta0, eta1);
  auto gamma = std::bind(dist, engine);
  std::generate(theta_out, theta_out + 2 * K, gamma);
  return 0;
}

int64_t ammsb_host_train_ppx_edges(const ammsb_host_dataset* d, uint64_t N, uint64_t E, float ratio, unsigned seed,
                                   uint64_t** out) {
  if (!d || !out || N < 2 || E == 0) return -1;
  mcmc::Config& cfg = const_cast<mcmc::Config&>(d->cfg);
  cfg.N = N;
  cfg.E = E;
  cfg.training_ppx_ratio = ratio;
  cfg.training_ppx_seed = seed;
  try {
    const std::vector<mcmc::Edge> e = mcmc::MakeEdgesForTrainingPerplexity(cfg);
    *out = static_cast<uint64_t*>(malloc(sizeof(uint64_t) * (e.empty() ? 1 : e.size())));
    if (!*out) return -1;
    memcpy(*out, e.data(), sizeof(uint64_t) * e.size());
    return static_cast<int64_t>(e.size());
  } catch (const std::exception&) {
    return -1;
  }
}

int ammsb_host_sample(const ammsb_host_dataset* d, uint64_t N, uint64_t E, uint64_t mini_batch, int strategy,
                      unsigned* seed, uint64_t* edges_out, uint64_t edges_cap, uint64_t* n_edges, uint32_t* nodes_out,
                      uint64_t nodes_cap, uint64_t* n_nodes, float* weight) {
  if (!d || !seed || !edges_out || !n_edges || !nodes_out || !n_nodes || !weight) return -1;
  if (strategy < 0 || strategy > 5) return -1;
  // the samplers read N, E, mini_batch_size and the data half of the Config
  mcmc::Config& cfg = const_cast<mcmc::Config&>(d->cfg);
  cfg.N = N;
  cfg.E = E;
  cfg.mini_batch_size = mini_batch;
  std::vector<mcmc::Edge> edges;
  *weight = mcmc::GetSampler(static_cast<mcmc::SampleStrategy>(strategy))(cfg, &edges, seed);
  std::vector<mcmc::Vertex> nodes;
  mcmc::ExtractNodesFromMiniBatch(edges, &nodes);
  *n_edges = edges.size();
  *n_nodes = nodes.size();
  if (edges.size() > edges_cap || nodes.size() > nodes_cap) return -2;  // learner.cc:184-189 "N | cap"
  memcpy(edges_out, edges.data(), sizeof(uint64_t) * edges.size());
  memcpy(nodes_out, nodes.data(), sizeof(uint32_t) * nodes.size());
  return 0;
}

}  // extern "C"
