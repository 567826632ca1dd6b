// Host mini-batch samplers.  Behavioural restatement of the reference's mcmc/sample.cc:177-303 and
// learner.cc:162-173: identical rand_r() call order and identical container types, so a seed yields
// the same mini-batch (edge order included) under the same libc / libstdc++.
#include "mcmc/sample.h"

#include <algorithm>
#include <cctype>
#include <cstdlib>
#include <istream>
#include <queue>
#include <stdexcept>
#include <unordered_set>

#include "mcmc/config.h"

namespace mcmc {

namespace {
inline Edge Canon(Vertex a, Vertex b) { return MakeEdge(std::min(a, b), std::max(a, b)); }

std::string Lower(std::string s) {
  for (char& c : s) c = static_cast<char>(std::tolower(static_cast<unsigned char>(c)));
  return s;
}

// shared frontier walk of the two breadth-first strategies (sample.cc:177-244): pop a vertex, expand
// it once, push what `expand` discovers
template <class Expand>
void BreadthFirstFill(const Config& cfg, unsigned int* seed, std::unordered_set<Edge>* picked, Expand expand) {
  std::unordered_set<Vertex> visited;
  std::queue<Vertex> frontier;
  while (picked->size() < cfg.mini_batch_size) {
    if (frontier.empty()) {
      Vertex start;
      do {
        start = rand_r(seed) % cfg.N;
      } while (visited.count(start));
      frontier.push(start);
    }
    const Vertex u = frontier.front();
    frontier.pop();
    if (visited.insert(u).second) expand(u, &frontier);
  }
}
}  // namespace

std::string to_string(const SampleStrategy& s) {
  switch (s) {
    case NodeLink: return "NodeLink";
    case NodeNonLink: return "NodeNonLink";
    case Node: return "Node";
    case BFLink: return "BFLink";
    case BFNonLink: return "BFNonLink";
    case BF: return "BF";
  }
  throw std::invalid_argument("Invalid strategy");
}

std::istream& operator>>(std::istream& in, SampleStrategy& strategy) {
  std::string token;
  in >> token;
  const std::string t = Lower(token);
  if (t == "nodelink") strategy = NodeLink;
  else if (t == "nodenonlink") strategy = NodeNonLink;
  else if (t == "node") strategy = Node;
  else if (t == "bflink") strategy = BFLink;
  else if (t == "bfnonlink") strategy = BFNonLink;
  else if (t == "bf") strategy = BF;
  else throw std::invalid_argument("Invalid SampleStrategy: " + token);
  return in;
}

Float sampleBreadthFirstNonLink(const Config& cfg, std::vector<Edge>* edges, unsigned int* seed) {
  std::unordered_set<Edge> picked;
  BreadthFirstFill(cfg, seed, &picked, [&](Vertex u, std::queue<Vertex>* frontier) {
    const auto& nbrs = cfg.trainingGraph->NeighborsOf(u);
    for (uint32_t i = 0; i < 32 && picked.size() < cfg.mini_batch_size; ++i) {
      Vertex v;
      do {
        v = rand_r(seed) % cfg.N;
      } while (u == v || std::find(nbrs.begin(), nbrs.end(), v) != nbrs.end());
      frontier->push(v);
      picked.insert(Canon(u, v));
    }
  });
  edges->insert(edges->begin(), picked.begin(), picked.end());
  return static_cast<Float>((cfg.N * (cfg.N - 1) / 2.0 - cfg.E) / cfg.mini_batch_size);
}

Float sampleBreadthFirstLink(const Config& cfg, std::vector<Edge>* edges, unsigned int* seed) {
  std::unordered_set<Edge> picked;
  BreadthFirstFill(cfg, seed, &picked, [&](Vertex u, std::queue<Vertex>* frontier) {
    for (Vertex v : cfg.trainingGraph->NeighborsOf(u)) {
      if (picked.size() >= cfg.mini_batch_size) break;
      frontier->push(v);
      picked.insert(Canon(u, v));
    }
  });
  edges->insert(edges->begin(), picked.begin(), picked.end());
  return static_cast<Float>(cfg.E) / cfg.mini_batch_size;
}

Float sampleBreadthFirst(const Config& cfg, std::vector<Edge>* edges, unsigned int* seed) {
  return (rand_r(seed) % 2) ? sampleBreadthFirstLink(cfg, edges, seed) : sampleBreadthFirstNonLink(cfg, edges, seed);
}

// all training edges of one random vertex that has any (sample.cc:249-267); weight N
Float sampleNodeLink(const Config& cfg, std::vector<Edge>* edges, unsigned int* seed) {
  std::unordered_set<Vertex> tried;
  std::unordered_set<Edge> picked;
  while (picked.empty()) {
    const Vertex u = rand_r(seed) % cfg.N;
    if (tried.insert(u).second)
      for (Vertex v : cfg.trainingGraph->NeighborsOf(u)) picked.insert(Canon(u, v));
  }
  edges->insert(edges->begin(), picked.begin(), picked.end());
  return static_cast<Float>(cfg.N);
}

// m distinct non-links (absent from training and held-out) sharing one random end point
// (sample.cc:273-293); weight 2E/m.  As in the reference v == u is not excluded.
Float sampleNodeNonLink(const Config& cfg, std::vector<Edge>* edges, unsigned int* seed) {
  std::unordered_set<Edge> picked;
  const Vertex u = rand_r(seed) % cfg.N;
  while (picked.size() < cfg.mini_batch_size) {
    Edge e;
    do {
      const Vertex v = rand_r(seed) % cfg.N;
      e = Canon(u, v);
    } while ((cfg.heldout && cfg.heldout->Has(e)) || cfg.training->Has(e));
    picked.insert(e);
  }
  edges->insert(edges->begin(), picked.begin(), picked.end());
  return (2 * cfg.E) / static_cast<Float>(cfg.mini_batch_size);
}

Float sampleNode(const Config& cfg, std::vector<Edge>* edges, unsigned int* seed) {
  return (rand_r(seed) % 2) ? sampleNodeLink(cfg, edges, seed) : sampleNodeNonLink(cfg, edges, seed);
}

SamplerFn GetSampler(SampleStrategy s) {
  switch (s) {
    case NodeLink: return sampleNodeLink;
    case NodeNonLink: return sampleNodeNonLink;
    case Node: return sampleNode;
    case BFLink: return sampleBreadthFirstLink;
    case BFNonLink: return sampleBreadthFirstNonLink;
    case BF: return sampleBreadthFirst;
  }
  return nullptr;
}

void ExtractNodesFromMiniBatch(const std::vector<Edge>& edges, std::vector<Vertex>* nodes_vec) {
  std::unordered_set<Vertex> nodes;
  for (Edge e : edges) {
    Vertex u, v;
    std::tie(u, v) = Vertices(e);
    nodes.insert(u);
    nodes.insert(v);
  }
  nodes_vec->clear();
  nodes_vec->insert(nodes_vec->begin(), nodes.begin(), nodes.end());
}

}  // namespace mcmc
