// Graph, data-set preparation, data-set files and the synthetic a-MMSB generator.
#include "mcmc/data.h"

#include <zlib.h>

#include <algorithm>
#include <thread>
#include <sched.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <unordered_map>
#include <unordered_set>

namespace mcmc {

// ------------------------------------------------------------------------------------ Graph

Graph::Graph(uint64_t num_nodes, const std::vector<Edge>& unique_edges)
    : num_nodes_(num_nodes), unique_edges_(unique_edges), adjacency_(num_nodes), max_fan_out_(0) {
  for (Edge e : unique_edges_) {
    Vertex u, v;
    std::tie(u, v) = Vertices(e);
    adjacency_[u].push_back(v);
    adjacency_[v].push_back(u);
    max_fan_out_ = std::max<uint64_t>(max_fan_out_, adjacency_[u].size());
    max_fan_out_ = std::max<uint64_t>(max_fan_out_, adjacency_[v].size());
  }
}

Edge Graph::GetRandomEdge() const {  // data.cc:26-34 (global rand(), as there)
  Vertex u;
  do {
    u = rand() % num_nodes_;
  } while (adjacency_[u].empty());
  const Vertex v = adjacency_[u][rand() % adjacency_[u].size()];
  return MakeEdge(u, v);
}

void Graph::ExportCSR(std::vector<uint64_t>* offsets, std::vector<Vertex>* targets) const {
  offsets->assign(num_nodes_ + 1, 0);
  for (uint64_t u = 0; u < num_nodes_; ++u) (*offsets)[u + 1] = (*offsets)[u] + adjacency_[u].size();
  targets->resize((*offsets)[num_nodes_]);
  for (uint64_t u = 0; u < num_nodes_; ++u)
    std::copy(adjacency_[u].begin(), adjacency_[u].end(), targets->begin() + (*offsets)[u]);
}

// -------------------------------------------------------------------------------- text loader

bool GetUniqueEdgesFromFile(const std::string& filename, uint64_t* count_vertices, std::vector<Edge>* vals) {
  std::ifstream in(filename);
  if (!in.is_open()) return false;
  std::string line;
  for (int i = 0; i < 4; ++i) std::getline(in, line);  // header
  std::unordered_set<Vertex> seen;
  std::vector<Edge> raw;
  uint64_t a, b;
  while (in >> a >> b) {
    const uint64_t x = std::min(a, b), y = std::max(a, b);
    raw.push_back(MakeEdge(static_cast<Vertex>(x), static_cast<Vertex>(y)));
    seen.insert(static_cast<Vertex>(x));
    seen.insert(static_cast<Vertex>(y));
  }
  if (in.bad()) return false;
  // dense ids in the set's iteration order (data.cc:63-67)
  std::unordered_map<Vertex, Vertex> dense;
  Vertex next = 0;
  for (Vertex v : seen) dense[v] = next++;
  *count_vertices = dense.size();
  for (Edge e : raw) {
    Vertex u, v;
    std::tie(u, v) = Vertices(e);
    vals->push_back(MakeEdge(dense[u], dense[v]));  // NB: not re-canonicalised, as in data.cc:72
  }
  std::sort(vals->begin(), vals->end());
  vals->erase(std::unique(vals->begin(), vals->end()), vals->end());
  // std::random_shuffle (data.cc:76) no longer exists in C++17; its libstdc++ form is this loop
  // over the global rand():
  for (size_t i = 1; i < vals->size(); ++i) std::swap((*vals)[i], (*vals)[rand() % (i + 1)]);
  return true;
}

// ------------------------------------------------------------------------ train / held-out

bool GenerateSetsFromEdges(uint64_t N, const std::vector<Edge>& vals, double heldout_ratio,
                           std::vector<Edge>* training_edges, std::vector<Edge>* heldout_edges,
                           std::unique_ptr<Set>* training, std::unique_ptr<Set>* heldout) {
  const size_t training_len = static_cast<size_t>(std::ceil((1 - heldout_ratio / 2) * vals.size()));
  const size_t heldout_len = vals.size() - training_len;
  if (heldout_len > 0) {
    heldout->reset(new Set(heldout_len));
    if (!(*heldout)->SetContents(vals.begin(), vals.begin() + heldout_len)) {
      heldout->reset();
      return false;
    }
    heldout_edges->insert(heldout_edges->end(), vals.begin(), vals.begin() + heldout_len);
  }
  training->reset(new Set(training_len));
  if (!(*training)->SetContents(vals.begin() + heldout_len, vals.end())) {
    training->reset();
    if (heldout_len > 0) heldout->reset();
    return false;
  }
  training_edges->insert(training_edges->end(), vals.begin() + heldout_len, vals.end());
  // as many fake (non-link) pairs as real held-out links (data.cc:108-126); global rand() as there
  std::unordered_set<Edge> fake;
  for (size_t i = 0; i < heldout_len; ++i) {
    Edge e;
    do {
      const Vertex u = rand() % N;
      Vertex v;
      do {
        v = rand() % N;
      } while (u == v);
      e = MakeEdge(std::min(u, v), std::max(u, v));
    } while (fake.count(e) || (*heldout)->Has(e) || (*training)->Has(e));
    fake.insert(e);
    heldout_edges->push_back(e);
  }
  return true;
}

bool GenerateSetsFromFile(const std::string& filename, double heldout_ratio, uint64_t* count_vertices,
                          std::vector<Edge>* training_edges, std::vector<Edge>* heldout_edges,
                          std::unique_ptr<Set>* training, std::unique_ptr<Set>* heldout) {
  std::vector<Edge> vals;
  return GetUniqueEdgesFromFile(filename, count_vertices, &vals) &&
         GenerateSetsFromEdges(*count_vertices, vals, heldout_ratio, training_edges, heldout_edges, training,
                               heldout);
}

// ---------------------------------------------------------------------------- data-set files

bool DumpDataset(const std::string& filename, uint64_t N, Float heldout_ratio, const std::vector<Edge>& edges) {
  gzFile f = gzopen(filename.c_str(), "wb");
  if (!f) return false;
  const uint64_t n = edges.size();
  bool ok = gzwrite(f, &N, sizeof N) == (int)sizeof N &&
            gzwrite(f, &heldout_ratio, sizeof heldout_ratio) == (int)sizeof heldout_ratio &&
            gzwrite(f, &n, sizeof n) == (int)sizeof n;
  const char* p = reinterpret_cast<const char*>(edges.data());
  size_t left = n * sizeof(Edge);
  while (ok && left) {
    const unsigned chunk = left > (1u << 30) ? (1u << 30) : static_cast<unsigned>(left);
    ok = gzwrite(f, p, chunk) == (int)chunk;
    p += chunk;
    left -= chunk;
  }
  return (gzclose(f) == Z_OK) && ok;
}

bool LoadDataset(const std::string& filename, uint64_t* N, Float* heldout_ratio, std::vector<Edge>* edges) {
  gzFile f = gzopen(filename.c_str(), "rb");
  if (!f) return false;
  uint64_t n = 0;
  bool ok = gzread(f, N, sizeof *N) == (int)sizeof *N &&
            gzread(f, heldout_ratio, sizeof *heldout_ratio) == (int)sizeof *heldout_ratio &&
            gzread(f, &n, sizeof n) == (int)sizeof n;
  if (ok) {
    edges->resize(n);
    char* p = reinterpret_cast<char*>(edges->data());
    size_t left = n * sizeof(Edge);
    while (ok && left) {
      const unsigned chunk = left > (1u << 30) ? (1u << 30) : static_cast<unsigned>(left);
      ok = gzread(f, p, chunk) == (int)chunk;
      p += chunk;
      left -= chunk;
    }
  }
  gzclose(f);
  return ok;
}

// ---------------------------------------------------------------------- synthetic generator

namespace {
struct SplitMix64 {
  uint64_t s;
  uint64_t next() {
    uint64_t z = (s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
  }
  double unit() { return (next() >> 11) * (1.0 / 9007199254740992.0); }
  uint64_t below(uint64_t n) { return next() % n; }
};
}  // namespace

// std::sort's result with the threads this process may use: chunks sorted side by side, then merged pairwise level by
// level (a sorted array is a sorted array: the output does not depend on the thread count).
static void ParallelSort(std::vector<Edge>* v) {
  const size_t n = v->size();
  unsigned T = std::thread::hardware_concurrency();
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0) T = std::min<unsigned>(T ? T : 1u, static_cast<unsigned>(CPU_COUNT(&set)));
  T = std::max(1u, std::min(T, 16u));
  while (T > 1 && n / T < (1u << 20)) T /= 2;  // small inputs: not worth the threads
  unsigned chunks = 1;
  while (chunks * 2 <= T) chunks *= 2;
  if (chunks == 1) {
    std::sort(v->begin(), v->end());
    return;
  }
  std::vector<size_t> cut(chunks + 1);
  for (unsigned c = 0; c <= chunks; ++c) cut[c] = n * c / chunks;
  {
    std::vector<std::thread> th;
    for (unsigned c = 0; c < chunks; ++c)
      th.emplace_back([&, c] { std::sort(v->begin() + cut[c], v->begin() + cut[c + 1]); });
    for (std::thread& t : th) t.join();
  }
  for (unsigned width = 1; width < chunks; width *= 2) {
    std::vector<std::thread> th;
    for (unsigned c = 0; c + width < chunks; c += 2 * width)
      th.emplace_back([&, c, width] {
        std::inplace_merge(v->begin() + cut[c], v->begin() + cut[c + width], v->begin() + cut[std::min(c + 2 * width, chunks)]);
      });
    for (std::thread& t : th) t.join();
  }
}

std::vector<Edge> GenerateSyntheticGraph(uint64_t N, uint32_t K_true, double avg_degree, uint64_t seed) {
  SplitMix64 rng{seed};
  // memberships: 1-3 communities per node (uniform count, uniform choice)
  std::vector<std::vector<Vertex>> members(K_true);
  for (uint64_t v = 0; v < N; ++v) {
    const unsigned cnt = 1 + static_cast<unsigned>(rng.below(3));
    unsigned got[3];
    for (unsigned c = 0; c < cnt; ++c) {
      unsigned k;
      bool dup;
      do {
        k = static_cast<unsigned>(rng.below(K_true));
        dup = false;
        for (unsigned d = 0; d < c; ++d) dup |= (got[d] == k);
      } while (dup);
      got[c] = k;
      members[k].push_back(static_cast<Vertex>(v));
    }
  }
  // community strengths beta_k ~ U(0.3, 0.7); the number of intra-community edges is proportional
  // to beta_k * |C_k|^2 and scaled so that the total is N * avg_degree / 2
  std::vector<double> weight(K_true);
  double total_w = 0;
  for (uint32_t k = 0; k < K_true; ++k) {
    const double beta_k = 0.3 + 0.4 * rng.unit();
    const double sz = static_cast<double>(members[k].size());
    weight[k] = beta_k * sz * (sz - 1) / 2;
    total_w += weight[k];
  }
  const double target = N * avg_degree / 2 * 1.02;  // small surplus for the duplicates removed below
  std::vector<Edge> edges;
  edges.reserve(static_cast<size_t>(target * 1.01) + 16);
  for (uint32_t k = 0; k < K_true; ++k) {
    const auto& m = members[k];
    if (m.size() < 2) continue;
    const uint64_t cnt = static_cast<uint64_t>(target * weight[k] / total_w);
    for (uint64_t i = 0; i < cnt; ++i) {
      const Vertex a = m[rng.below(m.size())], b = m[rng.below(m.size())];
      if (a == b) continue;
      edges.push_back(MakeEdge(std::min(a, b), std::max(a, b)));
    }
  }
  ParallelSort(&edges);  // (the same sorted array as std::sort: 3.3e8 keys at C5 take 25 s on one thread)
  edges.erase(std::unique(edges.begin(), edges.end()), edges.end());
  for (size_t i = edges.size(); i > 1; --i) std::swap(edges[i - 1], edges[rng.below(i)]);  // Fisher-Yates
  return edges;
}

}  // namespace mcmc
