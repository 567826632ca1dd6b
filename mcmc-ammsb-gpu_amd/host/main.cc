// Command-line driver with the reference's flags and flow (main.cc:26-172): load a SNAP edge list
// (or a gzip data-set dump), split off the held-out set, build the Learner, alternate Run(ppx_interval)
// with HeldoutPerplexity() until max-iters or SIGINT, print the statistics.
// boost::program_options is replaced by a small table-driven parser accepting the same spellings
// (--name value, --name=value, -x value).  New flags are marked (new).
#include <signal.h>

#include <algorithm>
#include <cstring>
#include <fstream>
#include <functional>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "mcmc/data.h"
#include "mcmc/exchange.h"
#include "mcmc/learner.h"

namespace clcuda = mcmc::clcuda;

namespace {

sig_atomic_t signaled = 0;
void handler(int) { signaled = 1; }

struct Option {
  std::string name;  // long name
  char shorthand;    // 0 if none
  std::string help;  // default value as text
  std::function<bool(const std::string&)> set;
};

template <class T>
Option Opt(const std::string& name, char s, T* target, const std::string& def) {
  return Option{name, s, def, [target](const std::string& v) {
                  try {  // (the enum / pair parsers throw on a bad token: config.cc:118-131)
                    std::istringstream in(v);
                    in >> std::boolalpha >> *target;
                    if (in.fail()) {  // bool also accepts 0/1, as program_options does
                      std::istringstream again(v);
                      again >> *target;
                      return !again.fail();
                    }
                    return true;
                  } catch (const std::exception&) {
                    return false;
                  }
                }};
}

Option OptStr(const std::string& name, char s, std::string* target) {
  return Option{name, s, "", [target](const std::string& v) {
                  *target = v;
                  return true;
                }};
}

bool FileExists(const std::string& f) {
  std::ifstream in(f);
  return in.good();
}

[[noreturn]] void Fatal(const std::string& msg) {
  std::cerr << "F " << msg << std::endl;
  exit(2);
}

}  // namespace

int main(int argc, char** argv) {
  {
    std::ostringstream s;
    for (int i = 0; i < argc; ++i) s << argv[i] << " ";
    std::cerr << "I " << s.str() << std::endl;
  }
  std::string filename, loadFile, dumpFile, ckptIn, ckptOut, exchangeKind;
  int deviceId = -1;
  mcmc::Config cfg;
  uint32_t max_iters = 100;
  bool dumpDataset = false, loadDataset = false;
  cfg.alpha = 0;            // main.cc:50
  cfg.beta_seed = {44, 45};  // main.cc:69-70 (the struct defaults differ)
  cfg.neighbor_seed = {56, 57};
  std::vector<Option> options = {
      OptStr("file", 'f', &filename),
      Opt("train-ppx-ratio", 0, &cfg.training_ppx_ratio, "0.01"),  // main.cc:47-49 (MCMC_CALC_TRAIN_PPX builds)
      Opt("train-ppx", 0, &cfg.calc_train_ppx, "0 (new: the reference's MCMC_CALC_TRAIN_PPX build option at run time)"),
      Opt("heldout-ratio", 'r', &cfg.heldout_ratio, "0.01"),
      Opt("alpha", 0, &cfg.alpha, "0"),
      Opt("a", 'a', &cfg.a, "0.0315"),
      Opt("b", 'b', &cfg.b, "1024"),
      Opt("c", 'c', &cfg.c, "0.5"),
      Opt("epsilon", 'e', &cfg.epsilon, "1e-07"),
      Opt("eta0", 0, &cfg.eta0, "1"),
      Opt("eta1", 0, &cfg.eta1, "1"),
      Opt("k", 'k', &cfg.K, "32"),
      Opt("mini_batch", 'm', &cfg.mini_batch_size, "32"),
      Opt("neighbors", 'n', &cfg.num_node_sample, "32"),
      Opt("ppx-wg", 0, &cfg.ppx_wg_size, "32"),
      Opt("ppx-interval", 'i', &cfg.ppx_interval, "100"),
      Opt("phi-wg", 0, &cfg.phi_wg_size, "32"),
      Opt("beta-wg", 0, &cfg.beta_wg_size, "32"),
      Opt("max-iters", 'x', &max_iters, "100"),
      Opt("sample", 's', &cfg.strategy, "Node"),
      Opt("sampler-wg", 0, &cfg.neighbor_sampler_wg_size, "32"),
      Opt("phi-seed", 0, &cfg.phi_seed, "42,43"),
      Opt("beta-seed", 0, &cfg.beta_seed, "44,45"),
      Opt("neighbor-seed", 0, &cfg.neighbor_seed, "56,57"),
      Opt("phi-mode", 0, &cfg.phi_mode, "PHI_NODE_PER_WORKGROUP_NAIVE"),
      Opt("phi-probs-shared", 0, &cfg.phi_probs_shared, "1"),
      Opt("phi-grads-shared", 0, &cfg.phi_grads_shared, "1"),
      Opt("phi-pi-shared", 0, &cfg.phi_pi_shared, "1"),
      Opt("phi-vwidth", 0, &cfg.phi_vector_width, "1"),
      Opt("beta-sum-grads-vwidth", 0, &cfg.sum_grads_vector_width, "1"),
      Opt("dump-data", 0, &dumpDataset, "0"),
      OptStr("dump-file", 0, &dumpFile),
      Opt("load-data", 0, &loadDataset, "0"),
      OptStr("load-file", 0, &loadFile),
      Opt("phi-disable-noise", 0, &cfg.phi_disable_noise, "0 (new)"),
      Opt("sample-seed0", 0, &cfg.sample_seed[0], "1804289383 (new: rand_r seeds of the two sample buffers)"),
      Opt("sample-seed1", 0, &cfg.sample_seed[1], "846930886 (new)"),
      Opt("device-sampling", 0, &cfg.device_sampling, "0 (new: draw mini-batches on the device)"),
      Opt("async", 0, &cfg.async_launch, "0 (new: enqueue-only loop; needs --device-sampling 1)"),
      Opt("graph", 0, &cfg.graph_launch, "0 (new: iterations as captured hipGraphs; needs --async 1)"),
      Opt("loop-timers", 0, &cfg.loop_timers, "1 (new: per-kernel device times in PrintStats under --async / --graph)"),
      Opt("phi-chunks", 0, &cfg.phi_chunks, "4 (new, with --exchange: blocks per rank whose exchange overlaps the next block's update_phi)"),
      Opt("phi-replicate", 0, &cfg.phi_replicate, "-1 (new, with --exchange: fraction of the virtual groups every rank computes itself; < 0 = measured at start-up)"),
      Opt("pi-candidates", 0, &cfg.pi_placement_candidates, "12 (new: allocations of pi timed under update_phi at start-up, the fastest kept; 0 = off)"),
      Opt("beta-grads", 0, &cfg.beta_grads, "-1 (new, with --exchange: 0 = gradient cut over the ranks + all-gather, 1 = every rank the whole gradient, -1 = 1 where update_pi folds into its launch)"),
      Opt("beta-shard-min-edges", 0, &cfg.beta_shard_min_edges, "4096 (new, with --exchange: mini-batches of at most this many edges keep their whole gradient on every rank)"),
      OptStr("exchange", 0, &exchangeKind),  // (new) rccl | host: one process per GPU, RANK / WORLD_SIZE / MASTER_* from the env
      Opt("device", 0, &deviceId, "-1 (new: HIP device; default LOCAL_RANK with --exchange, else 0)"),
      OptStr("checkpoint-in", 0, &ckptIn),    // (new) Learner::Parse before the first iteration
      OptStr("checkpoint-out", 0, &ckptOut),  // (new) Learner::Serialize after the last one
  };
  for (int i = 1; i < argc; ++i) {
    std::string arg = argv[i], value;
    if (arg == "--help" || arg == "-h") {
      for (const Option& o : options) {
        std::cout << "  ";
        if (o.shorthand) std::cout << "-" << o.shorthand << " [ --" << o.name << " ]";
        else std::cout << "--" << o.name;
        std::cout << " arg";
        if (!o.help.empty()) std::cout << " (=" << o.help << ")";
        std::cout << "\n";
      }
      return 1;  // main.cc:86-89
    }
    const Option* opt = nullptr;
    bool have_value = false;
    if (arg.rfind("--", 0) == 0) {
      const size_t eq = arg.find('=');
      const std::string name = arg.substr(2, eq == std::string::npos ? std::string::npos : eq - 2);
      for (const Option& o : options)
        if (o.name == name) opt = &o;
      if (eq != std::string::npos) {
        value = arg.substr(eq + 1);
        have_value = true;
      }
    } else if (arg.size() >= 2 && arg[0] == '-') {
      for (const Option& o : options)
        if (o.shorthand && o.shorthand == arg[1]) opt = &o;
      if (arg.size() > 2) {
        value = arg.substr(2);
        have_value = true;
      }
    }
    if (!opt) Fatal("unrecognised option '" + arg + "'");
    if (!have_value) {
      if (i + 1 >= argc) Fatal("the required argument for option '" + arg + "' is missing");
      value = argv[++i];
    }
    if (!opt->set(value)) Fatal("the argument ('" + value + "') for option '--" + opt->name + "' is invalid");
  }
  if (!loadDataset && !FileExists(filename)) Fatal("Failed to detect file: " + filename);  // main.cc:91-96
  if (loadDataset && loadFile.empty()) Fatal("load-file is required with load-data");
  if (dumpDataset && dumpFile.empty()) Fatal("dump-file is required with dump-data");

  std::vector<mcmc::Edge> unique_edges;
  if (!loadDataset) {
    if (!mcmc::GetUniqueEdgesFromFile(filename, &cfg.N, &unique_edges)) Fatal("Failed to generate sets from file " + filename);
    if (dumpDataset) {  // main.cc:110-127: dump and stop
      if (!mcmc::DumpDataset(dumpFile, cfg.N, cfg.heldout_ratio, unique_edges)) Fatal("cannot write " + dumpFile);
      return 0;
    }
  } else if (!mcmc::LoadDataset(loadFile, &cfg.N, &cfg.heldout_ratio, &unique_edges)) {
    Fatal("cannot read " + loadFile);
  }
  if (!mcmc::GenerateSetsFromEdges(cfg.N, unique_edges, cfg.heldout_ratio, &cfg.training_edges, &cfg.heldout_edges,
                                   &cfg.training, &cfg.heldout))
    Fatal("Failed to generate training/heldout sets");
  cfg.trainingGraph.reset(new mcmc::Graph(cfg.N, cfg.training_edges));
  cfg.heldoutGraph.reset(new mcmc::Graph(cfg.N, cfg.heldout_edges));
  if (cfg.alpha == 0) cfg.alpha = static_cast<mcmc::Float>(1) / cfg.K;  // main.cc:153
  cfg.E = unique_edges.size();

  if (deviceId < 0) {
    const char* lr = exchangeKind.empty() ? nullptr : getenv("LOCAL_RANK");
    deviceId = lr ? atoi(lr) : 0;
  }
  clcuda::Platform platform((size_t)0);
  clcuda::Device dev(platform, static_cast<size_t>(deviceId));
  clcuda::Context context(dev);
  clcuda::Queue queue(context, dev);
  int rank = 0;
  if (!exchangeKind.empty()) {
    try {
      cfg.exchange = mcmc::Exchange::FromEnvironment(exchangeKind, deviceId);
    } catch (const std::exception& e) {
      Fatal(std::string("exchange: ") + e.what());
    }
    rank = cfg.exchange->rank();
    std::cerr << "I exchange " << cfg.exchange->kind() << ": rank " << rank << " of " << cfg.exchange->world()
              << " on device " << deviceId << std::endl;
  }
  std::cerr << "I HIP:\n  Platform: " << dev.Vendor() << "\n  Device: " << dev.Name()
            << "\n  Device Driver: " << dev.Version() << std::endl;
  std::cerr << "I Loaded file " << (loadDataset ? loadFile : filename)
            << " (training max fan out = " << cfg.trainingGraph->MaxFanOut()
            << ", heldout max fan out = " << cfg.heldoutGraph->MaxFanOut() << ")" << std::endl;
  std::cerr << "I " << cfg << std::endl;
  signal(SIGINT, handler);
  std::unique_ptr<mcmc::Learner> learner_ptr;
  try {
    learner_ptr.reset(new mcmc::Learner(cfg, queue));
  } catch (const std::exception& e) {  // the reference LOG(FATAL)s on an unusable configuration (phi.cc:660, learner.cc:146)
    Fatal(e.what());
  }
  mcmc::Learner& learner = *learner_ptr;
  if (!ckptIn.empty()) {
    std::ifstream in(ckptIn, std::ios::binary);
    if (!in.good() || !learner.Parse(&in)) Fatal("cannot restore checkpoint " + ckptIn);
  }
  // with an exchange every rank computes every value (the calls are collectives); the lines carry the rank
  const std::string tag = cfg.exchange ? "I [" + std::to_string(rank) + "] " : "I ";
  std::cerr << tag << "ppx[0] = " << learner.HeldoutPerplexity() << std::endl;
  for (uint64_t i = 0; i < max_iters && !signaled; i += cfg.ppx_interval) {  // main.cc:162-168
    const uint64_t step = std::min<uint64_t>(max_iters - i, cfg.ppx_interval);
    learner.Run(static_cast<uint32_t>(step), &signaled);
    if (!signaled) std::cerr << tag << "ppx[" << i + step << "] = " << learner.HeldoutPerplexity() << std::endl;
    if (!signaled && cfg.calc_train_ppx)
      std::cerr << tag << "train ppx[" << i + step << "] = " << learner.TrainingPerplexity() << std::endl;
  }
  if (signaled) std::cerr << "I FORCED TERMINATE" << std::endl;
  if (!ckptOut.empty()) {
    // Serialize is a collective with an exchange; the states are identical afterwards and rank 0's file is the checkpoint
    std::ofstream out(rank == 0 ? ckptOut : std::string("/dev/null"), std::ios::binary);
    if (!learner.Serialize(&out)) Fatal("cannot write checkpoint " + ckptOut);
  }
  learner.PrintStats();
  return 0;
}
