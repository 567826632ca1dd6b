// C++ drop-in check: a caller written against the reference's API (main.cc:94-170 shape) compiles and
// runs against this library.  Mirrors serialize-test.cc:90-134's determinism contract without the
// checkpoint: same seeds => identical perplexities; and the sampler must actually learn.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "mcmc/data.h"
#include "mcmc/learner.h"

namespace clcuda = mcmc::clcuda;

static int fails = 0;
#define EXPECT(cond)                                          \
  do {                                                        \
    if (!(cond)) {                                            \
      printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond);   \
      ++fails;                                                \
    }                                                         \
  } while (0)

struct RunResult {
  mcmc::Float p0, p1, p2;
  std::vector<mcmc::Float> beta, row;
};

static RunResult RunOnce(uint64_t N, const std::vector<mcmc::Edge>& edges, uint32_t iters, bool device_sampling = false,
                         bool async = false, bool graph = false) {
  mcmc::Config cfg;
  cfg.device_sampling = device_sampling;
  cfg.async_launch = async;
  cfg.graph_launch = graph;
  cfg.N = N;
  cfg.K = 64;
  cfg.mini_batch_size = 256;
  cfg.num_node_sample = 16;
  cfg.heldout_ratio = 0.05;
  cfg.alpha = static_cast<mcmc::Float>(1) / cfg.K;  // main.cc:153
  cfg.phi_wg_size = cfg.beta_wg_size = cfg.ppx_wg_size = 64;
  cfg.beta_seed = {44, 45};
  cfg.neighbor_seed = {56, 57};
  srand(12345);  // GenerateSetsFromEdges draws the fake held-out pairs from rand()
  std::vector<mcmc::Edge> e = edges;
  bool ok = false;
  for (int attempt = 0; attempt < 64 && !ok; ++attempt) {  // the reference's cuckoo build can fail for some sizes
    cfg.training_edges.clear();
    cfg.heldout_edges.clear();
    ok = mcmc::GenerateSetsFromEdges(cfg.N, e, cfg.heldout_ratio, &cfg.training_edges, &cfg.heldout_edges, &cfg.training,
                                     &cfg.heldout);
    if (!ok) e.resize(e.size() - 40);
  }
  EXPECT(ok);
  cfg.trainingGraph.reset(new mcmc::Graph(cfg.N, cfg.training_edges));
  cfg.heldoutGraph.reset(new mcmc::Graph(cfg.N, cfg.heldout_edges));
  cfg.E = e.size();

  clcuda::Platform platform((size_t)0);
  clcuda::Device dev(platform, 0);
  clcuda::Context context(dev);
  clcuda::Queue queue(context, dev);
  mcmc::Learner learner(cfg, queue);
  RunResult r;
  if (getenv("AMMSB_TEST_DEBUG")) {
    double sb = 0, sr = 0;
    for (float v : learner.GetBeta()) sb += v;
    for (float v : learner.GetPiRow(17)) sr += v * v;
    unsigned long long h = 0;
    for (auto x : cfg.heldout_edges) h = h * 1315423911ull + x;
    printf("debug: |E|=%zu heldout=%zu hash=%llx beta_sum=%.9f row17_sq=%.9g bins=%zu/%zu\n", e.size(),
           cfg.heldout_edges.size(), h, sb, sr, cfg.training->BinsPerBucket(), cfg.heldout->BinsPerBucket());
  }
  r.p0 = learner.HeldoutPerplexity();
  learner.Run(iters / 2);
  r.p1 = learner.HeldoutPerplexity();
  learner.Run(iters - iters / 2);
  r.p2 = learner.HeldoutPerplexity();
  r.beta = learner.GetBeta();
  r.row = learner.GetPiRow(17);
  EXPECT(learner.MiniBatchEdges() > 0);
  return r;
}

// ---- checkpoint modes (serialize-test.cc:90-134 + cross-language round trip with the Python learner)
//   learner_test ckpt DIR     build the data set, write DIR/edges.bin; run 40, Serialize -> DIR/cpp.ckpt,
//                             run 40 -> ppx; fresh learner, Parse, run 40 -> same ppx; write DIR/cpp_ppx.txt
//   learner_test resume DIR   read DIR/edges.bin + DIR/py.ckpt (written by the Python learner), run 40,
//                             write DIR/cpp_from_py_ppx.txt

static void FillConfig(mcmc::Config* cfg, uint64_t N) {
  cfg->N = N;
  cfg->K = 64;
  cfg->mini_batch_size = 256;
  cfg->num_node_sample = 16;
  cfg->heldout_ratio = 0.05;
  cfg->alpha = static_cast<mcmc::Float>(1) / cfg->K;
  cfg->phi_wg_size = cfg->beta_wg_size = cfg->ppx_wg_size = 64;
  cfg->beta_seed = {44, 45};
  cfg->neighbor_seed = {56, 57};
}

static bool Prepare(mcmc::Config* cfg, const std::vector<mcmc::Edge>& e) {
  cfg->training_edges.clear();
  cfg->heldout_edges.clear();
  srand(12345);
  if (!mcmc::GenerateSetsFromEdges(cfg->N, e, cfg->heldout_ratio, &cfg->training_edges, &cfg->heldout_edges,
                                   &cfg->training, &cfg->heldout))
    return false;
  cfg->trainingGraph.reset(new mcmc::Graph(cfg->N, cfg->training_edges));
  cfg->heldoutGraph.reset(new mcmc::Graph(cfg->N, cfg->heldout_edges));
  cfg->E = e.size();
  return true;
}

static void WritePpx(const std::string& path, mcmc::Float ppx) {
  uint32_t bits;
  memcpy(&bits, &ppx, 4);
  std::ofstream f(path);
  f << bits << "\n";
}

static int CheckpointMode(const std::string& mode, const std::string& dir) {
  const uint64_t N = 20000;
  const uint32_t iters = 40;
  mcmc::Config cfg;
  FillConfig(&cfg, N);
  cfg.device_sampling = getenv("AMMSB_TEST_DEVICE_SAMPLING") != nullptr;  // C++-only end-to-end variant
  cfg.async_launch = getenv("AMMSB_TEST_ASYNC") != nullptr;
  cfg.graph_launch = getenv("AMMSB_TEST_GRAPH") != nullptr;
  std::vector<mcmc::Edge> e;
  if (mode == "ckpt") {
    e = mcmc::GenerateSyntheticGraph(N, 16, 16, 7);
    bool ok = false;
    for (int attempt = 0; attempt < 64 && !(ok = Prepare(&cfg, e)); ++attempt) e.resize(e.size() - 40);
    EXPECT(ok);
    std::ofstream f(dir + "/edges.bin", std::ios::binary);
    f.write(reinterpret_cast<const char*>(e.data()), e.size() * sizeof(mcmc::Edge));
  } else {
    std::ifstream f(dir + "/edges.bin", std::ios::binary | std::ios::ate);
    e.resize(static_cast<size_t>(f.tellg()) / sizeof(mcmc::Edge));
    f.seekg(0);
    f.read(reinterpret_cast<char*>(e.data()), e.size() * sizeof(mcmc::Edge));
    EXPECT(Prepare(&cfg, e));
  }
  clcuda::Platform platform((size_t)0);
  clcuda::Device dev(platform, 0);
  clcuda::Context context(dev);
  clcuda::Queue queue(context, dev);
  if (mode == "ckpt") {
    std::ostringstream out;
    mcmc::Float ppx;
    {
      mcmc::Learner learner1(cfg, queue);
      learner1.Run(iters);
      EXPECT(learner1.Serialize(&out));
      learner1.Run(iters);
      ppx = learner1.HeldoutPerplexity();
    }
    {
      std::istringstream in(out.str());
      mcmc::Learner learner2(cfg, queue);
      EXPECT(learner2.Parse(&in));
      learner2.Run(iters);
      EXPECT(ppx == learner2.HeldoutPerplexity());
      // a truncated stream and a stream for another shape are refused
      std::istringstream cut(out.str().substr(0, out.str().size() / 2));
      mcmc::Learner learner3(cfg, queue);
      EXPECT(!learner3.Parse(&cut));
    }
    {
      mcmc::Config other;
      FillConfig(&other, N);
      other.K = 32;
      other.alpha = static_cast<mcmc::Float>(1) / other.K;
      EXPECT(Prepare(&other, e));
      std::istringstream in(out.str());
      mcmc::Learner learner4(other, queue);
      EXPECT(!learner4.Parse(&in));
    }
    std::ofstream f(dir + "/cpp.ckpt", std::ios::binary);
    f << out.str();
    WritePpx(dir + "/cpp_ppx.txt", ppx);
    printf("checkpoint: %zu bytes, ppx %.6f\n", out.str().size(), ppx);
  } else {
    std::ifstream in(dir + "/py.ckpt", std::ios::binary);
    mcmc::Learner learner(cfg, queue);
    EXPECT(learner.Parse(&in));
    learner.Run(iters);
    WritePpx(dir + "/cpp_from_py_ppx.txt", learner.HeldoutPerplexity());
  }
  printf(fails ? "FAILED (%d)\n" : "OK\n", fails);
  return fails ? 1 : 0;
}

int main(int argc, char** argv) {
  if (argc == 3) return CheckpointMode(argv[1], argv[2]);
  const uint64_t N = 20000;
  const std::vector<mcmc::Edge> edges = mcmc::GenerateSyntheticGraph(N, 16, 16, 7);
  EXPECT(edges.size() > 100000);
  const RunResult a = RunOnce(N, edges, 300);
  const RunResult b = RunOnce(N, edges, 300);
  printf("ppx: %.6f -> %.6f -> %.6f\n", a.p0, a.p1, a.p2);
  EXPECT(std::isfinite(a.p0) && std::isfinite(a.p2));
  EXPECT(a.p2 < a.p0);                                     // it learns
  EXPECT(a.p0 == b.p0 && a.p1 == b.p1 && a.p2 == b.p2);    // bit-identical reruns
  EXPECT(a.beta == b.beta && a.row == b.row);
  // device-side mini-batch sampling (Config::device_sampling): learns, reruns are bit-identical
  const RunResult c = RunOnce(N, edges, 300, true);
  const RunResult d = RunOnce(N, edges, 300, true);
  printf("ppx (device sampling): %.6f -> %.6f -> %.6f\n", c.p0, c.p1, c.p2);
  EXPECT(c.p0 == a.p0);                                    // same initial state
  EXPECT(std::isfinite(c.p2) && c.p2 < c.p0);
  EXPECT(c.p1 == d.p1 && c.p2 == d.p2 && c.beta == d.beta && c.row == d.row);
  // enqueue-only loop (Config::async_launch): same launches in the same order => the same bits
  const RunResult e = RunOnce(N, edges, 300, true, true);
  EXPECT(e.p1 == c.p1 && e.p2 == c.p2 && e.beta == c.beta && e.row == c.row);
  // whole iterations as captured hipGraphs (Config::graph_launch): the same kernels again => the same bits
  const RunResult g = RunOnce(N, edges, 300, true, true, true);
  printf("ppx (graph launch): %.6f -> %.6f -> %.6f\n", g.p0, g.p1, g.p2);
  EXPECT(g.p1 == c.p1 && g.p2 == c.p2 && g.beta == c.beta && g.row == c.row);
  double s = 0;
  for (float v : a.row) s += v;
  EXPECT(std::fabs(s - 1.0) < 1e-4);
  for (size_t k = 0; k + 1 < a.beta.size(); k += 2) EXPECT(std::fabs(a.beta[k] + a.beta[k + 1] - 1.0f) < 1e-6f);
  // error behaviour: phi.cc:732
  printf(fails ? "FAILED (%d)\n" : "OK\n", fails);
  return fails ? 1 : 0;
}
