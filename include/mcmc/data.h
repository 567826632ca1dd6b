// Graph container and data-set preparation: drop-in for the reference's mcmc/data.h:16-52.
#ifndef MCMC_AMD_DATA_H_
#define MCMC_AMD_DATA_H_

#include <memory>
#include <string>
#include <vector>

#include "mcmc/cuckoo.h"

namespace mcmc {

using namespace mcmc::cuckoo;

// Undirected adjacency in edge-insertion order (data.cc:12-34).  ExportCSR() flattens it for the
// device-side mini-batch sampler.
class Graph {
 public:
  Graph(uint64_t num_nodes, const std::vector<Edge>& unique_edges);

  Edge GetRandomEdge() const;

  // neighbours of u, in the order the edges were inserted
  const std::vector<Vertex>& NeighborsOf(Vertex u) const { return adjacency_[u]; }

  const std::vector<Edge>& UniqueEdges() const { return unique_edges_; }
  uint64_t MaxFanOut() const { return max_fan_out_; }
  uint64_t NumNodes() const { return num_nodes_; }

  // offsets[N+1], targets[2E]: neighbours of u are targets[offsets[u] .. offsets[u+1])
  void ExportCSR(std::vector<uint64_t>* offsets, std::vector<Vertex>* targets) const;

 private:
  uint64_t num_nodes_;
  std::vector<Edge> unique_edges_;
  std::vector<std::vector<Vertex>> adjacency_;
  uint64_t max_fan_out_;
};

// SNAP-style text edge list: 4 header lines, then "a b" pairs (data.cc:36-78).  Vertices are
// renumbered to [0, N), edges canonicalised (u < v), sorted, de-duplicated and shuffled.
bool GetUniqueEdgesFromFile(const std::string& filename, uint64_t* count_vertices, std::vector<Edge>* vals);

// Training / held-out split with as many fake (non-link) held-out pairs as real ones (data.cc:80-128).
bool GenerateSetsFromEdges(uint64_t N, const std::vector<Edge>& vals, double heldout_ratio,
                           std::vector<Edge>* training_edges, std::vector<Edge>* heldout_edges,
                           std::unique_ptr<Set>* training, std::unique_ptr<Set>* heldout);

bool GenerateSetsFromFile(const std::string& filename, double heldout_ratio, uint64_t* count_vertices,
                          std::vector<Edge>* training_edges, std::vector<Edge>* heldout_edges,
                          std::unique_ptr<Set>* training, std::unique_ptr<Set>* heldout);

// gzip'd binary data-set dump of the reference CLI (main.cc:109-143):
// u64 N, f32 heldout_ratio, u64 num_edges, u64 edges[num_edges]
bool DumpDataset(const std::string& filename, uint64_t N, Float heldout_ratio, const std::vector<Edge>& edges);
bool LoadDataset(const std::string& filename, uint64_t* N, Float* heldout_ratio, std::vector<Edge>* edges);

// Synthetic a-MMSB graph (new: the reference has no generator).  K_true planted overlapping
// communities, 1-3 memberships per node, Erdos-Renyi inside each community with p_k chosen for the
// requested average degree; canonical, unique, no self loops; SplitMix64 stream from `seed`.
std::vector<Edge> GenerateSyntheticGraph(uint64_t N, uint32_t K_true, double avg_degree, uint64_t seed);

}  // namespace mcmc

#endif  // MCMC_AMD_DATA_H_
