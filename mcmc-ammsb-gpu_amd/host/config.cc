// mcmc::Config defaults, printing and the kernel-constant POD (reference: mcmc/config.{h,cc}).
#include "mcmc/config.h"

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <istream>
#include <ostream>
#include <sstream>
#include <stdexcept>

namespace mcmc {

Config::Config() {  // config.h:69-101
  heldout_ratio = 0.01;
  alpha = 0.001;
  a = 0.0315;
  b = 1024;
  c = 0.5;
  epsilon = 1e-7;
  eta0 = 1;
  eta1 = 1;
  K = 32;
  mini_batch_size = 32;
  num_node_sample = 32;
  N = 0;
  E = 0;
  ppx_wg_size = 32;
  ppx_interval = 100;
  neighbor_sampler_wg_size = 32;
  phi_wg_size = 32;
  beta_wg_size = 32;
  phi_disable_noise = false;
  phi_seed = {42, 43};
  beta_seed = {113, 117};
  neighbor_seed = {3337, 54351};
  strategy = Node;
  phi_mode = PHI_NODE_PER_WORKGROUP_NAIVE;
  phi_probs_shared = true;
  phi_grads_shared = true;
  phi_pi_shared = true;
  phi_vector_width = 1;
  sum_grads_vector_width = 1;
  device_sampling = false;
  async_launch = false;
  calc_train_ppx = false;
  training_ppx_ratio = 0.01;  // config.h:72
  training_ppx_seed = 1;
  graph_launch = false;
  loop_timers = true;
  device_sampling_seed = {1234, 5678};
  device_sampling_host_seed = 20260101;
  sample_seed[0] = 1804289383u;
  sample_seed[1] = 846930886u;
  phi_chunks = 4;
  phi_replicate = -1;
  beta_shard_min_edges = 4096;
  beta_grads = -1;
  pi_placement_candidates = 12;
}

std::ostream& operator<<(std::ostream& out, const ulong2& v) { return out << v[0] << "," << v[1]; }

std::istream& operator>>(std::istream& in, ulong2& v) {
  in >> v[0];
  if (in.get() != ',') throw std::invalid_argument("Invalid ulong2");
  in >> v[1];
  return in;
}

namespace {
std::string FloatToString(Float f) {  // config.cc:57-64
  char buf[64];
  snprintf(buf, sizeof buf, "%ef", static_cast<double>(f));
  return buf;
}
}  // namespace

std::vector<std::string> MakeCompileFlags(const Config& cfg) {
  return {"-DFLOAT_TYPE=float",
          "-DVERTEX_TYPE=uint",
          "-DEDGE_TYPE=ulong",
          "-DK=" + std::to_string(cfg.K),
          "-DN=" + std::to_string(cfg.N),
          "-DE=" + std::to_string(cfg.E),
          "-DALPHA=" + FloatToString(cfg.alpha),
          "-DEPS_A=" + FloatToString(cfg.a),
          "-DEPS_B=" + FloatToString(cfg.b),
          "-DEPS_C=" + FloatToString(cfg.c),
          "-DEPSILON=" + FloatToString(cfg.epsilon),
          "-DETA0=" + FloatToString(cfg.eta0),
          "-DETA1=" + FloatToString(cfg.eta1),
          "-DNUM_NEIGHBORS=" + std::to_string(cfg.num_node_sample)};
}

ammsb_params MakeKernelParams(const Config& cfg) {
  ammsb_params p;
  p.N = cfg.N;
  p.K = cfg.K;
  p.E = cfg.E;
  p.num_node_sample = static_cast<uint32_t>(cfg.num_node_sample);
  p.alpha = cfg.alpha;
  p.a = cfg.a;
  p.b = cfg.b;
  p.c = cfg.c;
  p.epsilon = cfg.epsilon;
  p.eta0 = cfg.eta0;
  p.eta1 = cfg.eta1;
  ammsb_params_quantize(&p);
  return p;
}

std::ostream& operator<<(std::ostream& out, const Config& cfg) {  // config.cc:85-116
  out << "Config:\n"
      << "heldout ratio: " << cfg.heldout_ratio << "\n"
      << "alpha: " << cfg.alpha << "\n"
      << "a: " << cfg.a << ", b: " << cfg.b << ", c: " << cfg.c << "\n"
      << "epsilon: " << cfg.epsilon << "\n"
      << "eta: (" << cfg.eta0 << ", " << cfg.eta1 << ")\n"
      << "K: " << cfg.K << "\n"
      << "m: " << cfg.mini_batch_size << "\n"
      << "n: " << cfg.num_node_sample << "\n"
      << "strategy: " << to_string(cfg.strategy) << "\n"
      << "ppx-wg: " << cfg.ppx_wg_size << "\n"
      << "phi-wg: " << cfg.phi_wg_size << "\n"
      << "beta-wg: " << cfg.beta_wg_size << "\n"
      << "phi-seed: " << cfg.phi_seed << "\n"
      << "beta-seed: " << cfg.beta_seed << "\n"
      << "neighbor-seed: " << cfg.neighbor_seed << "\n"
      << "|N|: " << cfg.N << "\n"
      << "|E|: " << cfg.E << "\n"
      << "phi_mode: " << to_string(cfg.phi_mode) << "\n"
      << "phi_vwidth: " << cfg.phi_vector_width << "\n"
      << "device_sampling: " << cfg.device_sampling << "\n";
  if (cfg.training) out << "|Training edges|: " << cfg.training->Size() << "\n";
  if (cfg.heldout) out << "|Heldout edges|: " << cfg.heldout->Size() << "\n";
  return out;
}

std::istream& operator>>(std::istream& in, PhiUpdaterMode& mode) {
  std::string token;
  in >> token;
  std::string t;
  for (char c : token) t += static_cast<char>(std::toupper(static_cast<unsigned char>(c)));
  if (t == "THREAD") mode = PHI_NODE_PER_THREAD;
  else if (t == "WG-NAIVE") mode = PHI_NODE_PER_WORKGROUP_NAIVE;
  else if (t == "WG-SHARED") mode = PHI_NODE_PER_WORKGROUP_SHARED;
  else if (t == "WG-GEN") mode = PHI_NODE_PER_WORKGROUP_CODE_GEN;
  else throw std::invalid_argument("Invalid phi mode: " + token);
  return in;
}

std::string to_string(const PhiUpdaterMode& mode) {
  switch (mode) {
    case PHI_NODE_PER_THREAD: return "THREAD";
    case PHI_NODE_PER_WORKGROUP_NAIVE: return "WG-NAIVE";
    case PHI_NODE_PER_WORKGROUP_SHARED: return "WG-SHARED";
    case PHI_NODE_PER_WORKGROUP_CODE_GEN: return "WG-GEN";
  }
  return "";
}

}  // namespace mcmc
