/*
 * ammsb_oracle_samplers.c -- CPU restatement of the reference's HOST mini-batch samplers.
 * TEST INFRASTRUCTURE ONLY (see ammsb_oracle.h): the checker for mcmc-ammsb-gpu_amd/host/sample.cc.
 *
 * Restates, in plain C:
 *   mcmc/sample.cc:177-303   sampleBreadthFirstNonLink / sampleBreadthFirstLink / sampleBreadthFirst /
 *                            sampleNodeLink / sampleNodeNonLink / sampleNode
 *   mcmc/learner.cc:162-173  Learner::ExtractNodesFromMiniBatch
 *   mcmc/data.cc:12-26       Graph adjacency in edge-insertion order
 *
 * The reference's outputs depend on two platform libraries, and both are part of what is restated:
 *   - glibc rand_r (the caller's `unsigned int* seed` stream) -- called directly, as the reference does;
 *   - the ITERATION ORDER of libstdc++'s std::unordered_set<Edge> / <Vertex> (`edges->insert(begin,
 *     set.begin(), set.end())`, sample.cc:205,236,266,290; `nodes.begin(), nodes.end()`, learner.cc:172).
 *     That order is a deterministic function of the insertion sequence: std::hash of an integer is the
 *     identity, buckets are chosen by `hash % bucket_count`, a node entering an empty bucket goes to the FRONT of
 *     the global singly linked list, a node entering a non-empty bucket goes right after that bucket's
 *     "before" node, and the bucket count follows _Prime_rehash_policy (growth factor 2, max load 1.0) over
 *     the library's prime table.  `uset_*` below restates that algorithm (bits/hashtable.h _M_insert_bucket_begin,
 *     _M_rehash_aux(unique keys); bits/hashtable_policy.h + src/c++11/hashtable_c++0x.cc _M_next_bkt,
 *     _M_need_rehash, as shipped with GCC 11) instead of calling the C++ container.  The prime table itself is
 *     data of the platform's libstdc++ (std::__detail::__prime_list, an exported object): it is read from the
 *     library, exactly the table the reference would use on this platform.
 *
 * Pin status: the reference holds no fixture for sampler outputs (its tests only run them); the restatement is
 * pinned by construction against the container it emulates (tests/test_oracle_samplers.py compares uset_* with
 * the real std::unordered_set through libammsb_host.so on random insertion sequences) and is then the independent
 * checker of host/sample.cc's rand_r stream, edge order, weight and node order for seeds 1..4.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "ammsb_oracle.h"

/* std::__detail::__prime_list (libstdc++.so.6): 256 + 48 primes + a terminator on LP64 */
extern const unsigned long _ZNSt8__detail12__prime_listE[];
#define PRIME_LIST _ZNSt8__detail12__prime_listE
#define N_PRIMES (256 + 48) /* sizeof(__prime_list) / sizeof(unsigned long) - 1 */

/* ------------------------------------------------------------------ std::unordered_set<integer> */

typedef struct {
  uint64_t key;
  int64_t next; /* index of the next node in the global list, -1 = end */
} uset_node;

typedef struct {
  uset_node* nodes; /* arena, index = insertion number */
  int64_t n_nodes, cap_nodes;
  int64_t* buckets; /* per bucket: index of the node BEFORE its first node; -2 = before_begin; -1 = empty */
  uint64_t bucket_count;
  int64_t head;            /* _M_before_begin._M_nxt */
  uint64_t next_resize;    /* _Prime_rehash_policy::_M_next_resize */
} uset;

#define BEFORE_BEGIN (-2)
#define EMPTY_BKT (-1)

static void uset_init(uset* s) {
  memset(s, 0, sizeof *s);
  s->bucket_count = 1; /* the single-bucket initial state */
  s->buckets = (int64_t*)malloc(sizeof(int64_t));
  s->buckets[0] = EMPTY_BKT;
  s->head = -1;
}

static void uset_free(uset* s) {
  free(s->nodes);
  free(s->buckets);
  memset(s, 0, sizeof *s);
}

/* _Prime_rehash_policy::_M_next_bkt (hashtable_c++0x.cc) */
static uint64_t next_bkt(uset* s, uint64_t n) {
  static const unsigned char fast_bkt[] = {2, 2, 2, 3, 5, 5, 7, 7, 11, 11, 11, 11, 13, 13};
  if (n < sizeof fast_bkt) {
    if (n == 0) return 1;
    s->next_resize = (uint64_t)floor(fast_bkt[n] * 1.0);
    return fast_bkt[n];
  }
  const unsigned long* lo = PRIME_LIST + 6;
  const unsigned long* last = PRIME_LIST + N_PRIMES - 1;
  /* std::lower_bound(lo, last, n) */
  size_t len = (size_t)(last - lo);
  while (len > 0) {
    size_t half = len >> 1;
    if (lo[half] < n) {
      lo += half + 1;
      len -= half + 1;
    } else {
      len = half;
    }
  }
  if (lo == last)
    s->next_resize = (uint64_t)-1;
  else
    s->next_resize = (uint64_t)floor((double)*lo * 1.0);
  return *lo;
}

/* _M_rehash_aux(n, true_type): relink every node, in list order, into n buckets */
static void uset_rehash(uset* s, uint64_t n) {
  int64_t* nb = (int64_t*)malloc(sizeof(int64_t) * n);
  for (uint64_t i = 0; i < n; ++i) nb[i] = EMPTY_BKT;
  int64_t p = s->head;
  s->head = -1;
  uint64_t bbegin_bkt = 0;
  while (p >= 0) {
    const int64_t next = s->nodes[p].next;
    const uint64_t bkt = s->nodes[p].key % n;
    if (nb[bkt] == EMPTY_BKT) {
      s->nodes[p].next = s->head;
      s->head = p;
      nb[bkt] = BEFORE_BEGIN;
      if (s->nodes[p].next >= 0) nb[bbegin_bkt] = p;
      bbegin_bkt = bkt;
    } else {
      const int64_t before = nb[bkt];
      int64_t* slot = before == BEFORE_BEGIN ? &s->head : &s->nodes[before].next;
      s->nodes[p].next = *slot;
      *slot = p;
    }
    p = next;
  }
  free(s->buckets);
  s->buckets = nb;
  s->bucket_count = n;
}

static int uset_contains(const uset* s, uint64_t key) {
  const uint64_t bkt = key % s->bucket_count;
  const int64_t before = s->buckets[bkt];
  if (before == EMPTY_BKT) return 0;
  int64_t p = before == BEFORE_BEGIN ? s->head : s->nodes[before].next;
  while (p >= 0) {
    if (s->nodes[p].key == key) return 1;
    p = s->nodes[p].next;
    if (p >= 0 && s->nodes[p].key % s->bucket_count != bkt) break;
  }
  return 0;
}

/* insert(key): returns 1 if inserted (std::pair<iterator,bool>::second) */
static int uset_insert(uset* s, uint64_t key) {
  if (uset_contains(s, key)) return 0;
  /* _M_need_rehash(bucket_count, element_count, 1) */
  const uint64_t n_elt = (uint64_t)s->n_nodes, n_ins = 1;
  if (n_elt + n_ins > s->next_resize) {
    double min_bkts = (double)((n_elt + n_ins > (s->next_resize ? 0u : 11u)) ? n_elt + n_ins
                                                                              : (s->next_resize ? 0u : 11u)) / 1.0;
    if (min_bkts >= (double)s->bucket_count) {
      const uint64_t a = (uint64_t)floor(min_bkts) + 1, b = s->bucket_count * 2;
      uset_rehash(s, next_bkt(s, a > b ? a : b));
    } else {
      s->next_resize = (uint64_t)floor((double)s->bucket_count * 1.0);
    }
  }
  if (s->n_nodes == s->cap_nodes) {
    s->cap_nodes = s->cap_nodes ? 2 * s->cap_nodes : 64;
    s->nodes = (uset_node*)realloc(s->nodes, sizeof(uset_node) * (size_t)s->cap_nodes);
  }
  const int64_t node = s->n_nodes++;
  s->nodes[node].key = key;
  /* _M_insert_bucket_begin */
  const uint64_t bkt = key % s->bucket_count;
  if (s->buckets[bkt] != EMPTY_BKT) {
    const int64_t before = s->buckets[bkt];
    int64_t* slot = before == BEFORE_BEGIN ? &s->head : &s->nodes[before].next;
    s->nodes[node].next = *slot;
    *slot = node;
  } else {
    s->nodes[node].next = s->head;
    s->head = node;
    if (s->nodes[node].next >= 0) s->buckets[s->nodes[s->nodes[node].next].key % s->bucket_count] = node;
    s->buckets[bkt] = BEFORE_BEGIN;
  }
  return 1;
}

static uint64_t uset_size(const uset* s) { return (uint64_t)s->n_nodes; }

/* copy in iteration order (begin() .. end()); returns the count */
static uint64_t uset_dump(const uset* s, uint64_t* out) {
  uint64_t n = 0;
  for (int64_t p = s->head; p >= 0; p = s->nodes[p].next) out[n++] = s->nodes[p].key;
  return n;
}

/* test entry: the iteration order after inserting keys[0..n) one by one */
uint64_t orc_uset_order(const uint64_t* keys, uint64_t n, uint64_t* out) {
  uset s;
  uset_init(&s);
  for (uint64_t i = 0; i < n; ++i) uset_insert(&s, keys[i]);
  const uint64_t cnt = uset_dump(&s, out);
  uset_free(&s);
  return cnt;
}

/* ------------------------------------------------------------------ graph (data.cc:12-26) */

typedef struct {
  uint64_t N;
  uint64_t* off;  /* [N + 1] */
  uint32_t* adj;  /* neighbours of u in edge-insertion order */
} orc_graph;

static void graph_build(orc_graph* g, uint64_t N, const uint64_t* edges, uint64_t n_edges) {
  g->N = N;
  g->off = (uint64_t*)calloc(N + 2, sizeof(uint64_t));
  for (uint64_t i = 0; i < n_edges; ++i) {
    g->off[(edges[i] >> 32) + 1]++;
    g->off[(edges[i] & 0xffffffffu) + 1]++;
  }
  for (uint64_t u = 0; u < N; ++u) g->off[u + 1] += g->off[u];
  g->adj = (uint32_t*)malloc(sizeof(uint32_t) * (g->off[N] ? g->off[N] : 1));
  uint64_t* fill = (uint64_t*)malloc(sizeof(uint64_t) * (N + 1));
  memcpy(fill, g->off, sizeof(uint64_t) * (N + 1));
  for (uint64_t i = 0; i < n_edges; ++i) { /* adjacency_[u].push_back(v); adjacency_[v].push_back(u) */
    const uint32_t u = (uint32_t)(edges[i] >> 32), v = (uint32_t)(edges[i] & 0xffffffffu);
    g->adj[fill[u]++] = v;
    g->adj[fill[v]++] = u;
  }
  free(fill);
}

static void graph_free(orc_graph* g) {
  free(g->off);
  free(g->adj);
}

static int graph_has_neighbor(const orc_graph* g, uint32_t u, uint32_t v) { /* std::find over NeighborsOf(u) */
  for (uint64_t i = g->off[u]; i < g->off[u + 1]; ++i)
    if (g->adj[i] == v) return 1;
  return 0;
}

static uint64_t canon(uint32_t u, uint32_t v) { /* MakeEdge(min, max), types.h:72-74 */
  const uint32_t lo = u < v ? u : v, hi = u < v ? v : u;
  return ((uint64_t)lo << 32) | hi;
}

/* ------------------------------------------------------------------ FIFO of vertices (std::queue<Vertex>) */

typedef struct {
  uint32_t* v;
  uint64_t head, tail, cap;
} fifo;
static void fifo_push(fifo* q, uint32_t x) {
  if (q->tail == q->cap) {
    q->cap = q->cap ? 2 * q->cap : 256;
    q->v = (uint32_t*)realloc(q->v, sizeof(uint32_t) * q->cap);
  }
  q->v[q->tail++] = x;
}

/* ------------------------------------------------------------------ the samplers (sample.cc:177-303) */

typedef struct {
  uint64_t N, E, m;
  const orc_graph* g;
  const uint64_t *tr_slots, *ho_slots;
  uint64_t tr_bins, ho_bins;
  uint32_t tr_prime, ho_prime;
  int has_heldout;
} sampler_cfg;

/* sample.cc:177-207 */
static float bf_nonlink(const sampler_cfg* c, uset* set, unsigned int* seed) {
  uset Us;
  fifo q = {0, 0, 0, 0};
  uset_init(&Us);
  while (uset_size(set) < c->m) {
    if (q.head == q.tail) {
      uint32_t u;
      do {
        u = (uint32_t)((uint64_t)rand_r(seed) % c->N);
      } while (uset_contains(&Us, u));
      fifo_push(&q, u);
    }
    const uint32_t u = q.v[q.head++];
    if (uset_insert(&Us, u)) {
      for (uint32_t i = 0; i < 32 && uset_size(set) < c->m; ++i) {
        uint32_t v;
        do {
          v = (uint32_t)((uint64_t)rand_r(seed) % c->N);
        } while (u == v || graph_has_neighbor(c->g, u, v));
        fifo_push(&q, v);
        uset_insert(set, canon(u, v));
      }
    }
  }
  uset_free(&Us);
  free(q.v);
  return (float)(((double)c->N * (double)(c->N - 1) / 2.0 - (double)c->E) / (double)c->m);
}

/* sample.cc:209-238 */
static float bf_link(const sampler_cfg* c, uset* set, unsigned int* seed) {
  uset Us;
  fifo q = {0, 0, 0, 0};
  uset_init(&Us);
  while (uset_size(set) < c->m) {
    if (q.head == q.tail) {
      uint32_t u;
      do {
        u = (uint32_t)((uint64_t)rand_r(seed) % c->N);
      } while (uset_contains(&Us, u));
      fifo_push(&q, u);
    }
    const uint32_t u = q.v[q.head++];
    if (uset_insert(&Us, u)) {
      for (uint64_t i = c->g->off[u]; i < c->g->off[u + 1]; ++i) {
        if (uset_size(set) < c->m) {
          const uint32_t v = c->g->adj[i];
          fifo_push(&q, v);
          uset_insert(set, canon(u, v));
        } else {
          break;
        }
      }
    }
  }
  uset_free(&Us);
  free(q.v);
  return (float)c->E / (float)c->m; /* static_cast<Float>(cfg.E) / cfg.mini_batch_size */
}

/* sample.cc:249-267 */
static float node_link(const sampler_cfg* c, uset* set, unsigned int* seed) {
  uset Us;
  uset_init(&Us);
  while (uset_size(set) == 0) {
    const uint32_t u = (uint32_t)((uint64_t)rand_r(seed) % c->N);
    if (uset_insert(&Us, u))
      for (uint64_t i = c->g->off[u]; i < c->g->off[u + 1]; ++i) uset_insert(set, canon(u, c->g->adj[i]));
  }
  uset_free(&Us);
  return (float)c->N;
}

/* sample.cc:273-293 (Vs is never filled in the reference, so its find() never hits) */
static float node_nonlink(const sampler_cfg* c, uset* set, unsigned int* seed) {
  const uint32_t u = (uint32_t)((uint64_t)rand_r(seed) % c->N);
  while (uset_size(set) < c->m) {
    uint64_t e;
    do {
      const uint32_t v = (uint32_t)((uint64_t)rand_r(seed) % c->N);
      e = canon(u, v);
    } while ((c->has_heldout && orc_set_has(c->ho_slots, c->ho_bins, c->ho_prime, e)) ||
             orc_set_has(c->tr_slots, c->tr_bins, c->tr_prime, e));
    uset_insert(set, e);
  }
  return (float)(2 * c->E) / (float)c->m; /* (2 * cfg.E) / static_cast<Float>(cfg.mini_batch_size) */
}

/*
 * One DoSample host half (learner.cc:175-178): sampler + ExtractNodesFromMiniBatch.
 * strategy: 0 Node, 1 NodeLink, 2 NodeNonLink, 3 BFLink, 4 BFNonLink, 5 BF (sample.h enum order of the build).
 * training_edges: the list the training Graph was built from (adjacency order).  Returns 0, or -1 on bad
 * arguments / too-small outputs (n_edges / n_nodes then hold the required sizes).
 */
int orc_host_sample(uint64_t N, uint64_t E, uint64_t mini_batch, int strategy, unsigned int* seed,
                    const uint64_t* training_edges, uint64_t n_training, const uint64_t* tr_slots, uint64_t tr_bins,
                    uint32_t tr_prime, const uint64_t* ho_slots, uint64_t ho_bins, uint32_t ho_prime,
                    uint64_t* edges_out, uint64_t edges_cap, uint64_t* n_edges, uint32_t* nodes_out,
                    uint64_t nodes_cap, uint64_t* n_nodes, float* weight) {
  if (!seed || !training_edges || !tr_slots || !edges_out || !nodes_out || !n_edges || !n_nodes || !weight) return -1;
  if (strategy < 0 || strategy > 5 || N == 0) return -1;
  orc_graph g;
  graph_build(&g, N, training_edges, n_training);
  sampler_cfg c = {N, E, mini_batch, &g, tr_slots, ho_slots, tr_bins, ho_bins, tr_prime, ho_prime, ho_slots != NULL};
  uset set;
  uset_init(&set);
  float w;
  switch (strategy) {
    case 0: w = (rand_r(seed) % 2) ? node_link(&c, &set, seed) : node_nonlink(&c, &set, seed); break; /* sample.cc:295-303 */
    case 1: w = node_link(&c, &set, seed); break;
    case 2: w = node_nonlink(&c, &set, seed); break;
    case 3: w = bf_link(&c, &set, seed); break;
    case 4: w = bf_nonlink(&c, &set, seed); break;
    default: w = (rand_r(seed) % 2) ? bf_link(&c, &set, seed) : bf_nonlink(&c, &set, seed); break; /* :240-247 */
  }
  *weight = w;
  int rc = 0;
  *n_edges = uset_size(&set);
  if (*n_edges > edges_cap) {
    rc = -1;
  } else {
    uset_dump(&set, edges_out); /* edges->insert(edges->begin(), set.begin(), set.end()) */
    /* learner.cc:162-173 */
    uset nodes;
    uset_init(&nodes);
    for (uint64_t i = 0; i < *n_edges; ++i) {
      uset_insert(&nodes, edges_out[i] >> 32);
      uset_insert(&nodes, edges_out[i] & 0xffffffffu);
    }
    *n_nodes = uset_size(&nodes);
    if (*n_nodes > nodes_cap) {
      rc = -1;
    } else {
      uint64_t* tmp = (uint64_t*)malloc(sizeof(uint64_t) * (*n_nodes ? *n_nodes : 1));
      uset_dump(&nodes, tmp);
      for (uint64_t i = 0; i < *n_nodes; ++i) nodes_out[i] = (uint32_t)tmp[i];
      free(tmp);
    }
    uset_free(&nodes);
  }
  uset_free(&set);
  graph_free(&g);
  return rc;
}

/* =====================================================================================================
 * The DEVICE mini-batch sampler (mcmc-ammsb-gpu_amd/csrc/ammsb_minibatch.hip), restated serially.
 *
 * It replaces sampleNodeLink / sampleNodeNonLink + ExtractNodesFromMiniBatch (sample.cc:249-293,
 * learner.cc:162-173) with its OWN random streams, so it has no reference output to equal; what it must equal is
 * this statement of its algorithm, bit for bit (tests/test_gpu_minibatch_oracle.py):
 *
 *   streams     candidate j owns xorshift128+ stream j, states {mix(sx + 2j), mix(sy + 2j + 1)}, mix = the SplitMix64
 *               finaliser (ammsb_rng_init_mixed), all-zero state replaced by {1, 0};
 *   draw        v_j = rand(stream j) mod N                     -- one draw per candidate per call, stream advanced;
 *   validity    v_j != u, (u, v_j) in neither the training nor the held-out set   (sample.cc:283-285; the reference's
 *               loop also lets v == u through -- its `Vs` is never filled -- which this sampler does not);
 *   de-dup      a valid candidate is kept iff no valid candidate with a smaller index drew the same v (the reference's
 *               std::unordered_set<Edge> keeps one copy of an edge; which copy is immaterial there);
 *   order       kept candidates in candidate order; the first m are the mini-batch: edges[r] = MakeEdge(min, max),
 *               nodes = {u, v_0, v_1, ...}  (learner.cc:162-173 builds the node list from the edges; here the edge list
 *               is node-stratified by construction: edge r joins nodes[0] and nodes[r + 1]);
 *   count       count_out[0] = kept candidates (may exceed m); fewer than m: the tail repeats earlier entries
 *               (edges[r] = edges[r mod count]) so that the buffers stay valid, and the caller is told (count < m);
 *   weight      link: N (sample.cc:268); non-link: 2 E / m (sample.cc:292), both as float.
 * ===================================================================================================== */

static uint64_t dev_splitmix64(uint64_t z) {
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

void orc_rng_init_mixed(orc_seed_t* seeds, uint64_t n, uint64_t sx, uint64_t sy) {
  for (uint64_t i = 0; i < n; ++i) {
    uint64_t x = dev_splitmix64(sx + 2 * i), y = dev_splitmix64(sy + 2 * i + 1);
    if ((x | y) == 0) x = 1;
    seeds[i].x = x;
    seeds[i].y = y;
  }
}

int orc_device_minibatch_nonlink(orc_seed_t* seeds, uint32_t n_candidates, uint32_t u, uint32_t m, uint64_t N,
                                 const uint64_t* tr_slots, uint64_t tr_bins, uint32_t tr_prime,
                                 const uint64_t* ho_slots, uint64_t ho_bins, uint32_t ho_prime, uint64_t* edges_out,
                                 uint32_t* nodes_out, uint32_t* count_out) {
  if (!seeds || !tr_slots || !edges_out || !nodes_out || !count_out || m == 0 || N == 0 || u >= N) return -1;
  uint32_t* first = (uint32_t*)malloc(sizeof(uint32_t) * N); /* first[v] = 1 + smallest valid candidate that drew v */
  if (!first) return -1;
  memset(first, 0, sizeof(uint32_t) * N);
  uint32_t kept = 0;
  nodes_out[0] = u;
  for (uint32_t j = 0; j < n_candidates; ++j) {
    const uint32_t v = (uint32_t)(orc_rand(&seeds[j]) % N);
    if (v == u) continue;
    const uint64_t a = u < v ? u : v, b = u < v ? v : u;
    const uint64_t e = (a << 32) | b;
    if (orc_set_has(tr_slots, tr_bins, tr_prime, e)) continue;
    if (ho_slots && orc_set_has(ho_slots, ho_bins, ho_prime, e)) continue;
    if (first[v]) continue; /* an earlier valid candidate has this partner */
    first[v] = j + 1;
    if (kept < m) {
      edges_out[kept] = e;
      nodes_out[1 + kept] = v;
    }
    ++kept;
  }
  free(first);
  count_out[0] = kept;
  if (kept < m && kept > 0) {
    for (uint32_t r = kept; r < m; ++r) {
      edges_out[r] = edges_out[r % kept];
      nodes_out[1 + r] = nodes_out[1 + r % kept];
    }
  }
  return 0;
}

/* all training edges (u, v) of u, v in CSR (adjacency) order: sampleNodeLink's edge set for one u (sample.cc:252-266) */
int orc_device_minibatch_link(const uint64_t* csr_offsets, const uint32_t* csr_targets, uint32_t u, uint64_t* edges_out,
                              uint32_t* nodes_out, uint32_t* n_out) {
  if (!csr_offsets || !csr_targets || !edges_out || !nodes_out || !n_out) return -1;
  const uint64_t lo = csr_offsets[u], hi = csr_offsets[u + 1];
  nodes_out[0] = u;
  for (uint64_t t = lo; t < hi; ++t) {
    const uint32_t v = csr_targets[t];
    const uint64_t a = u < v ? u : v, b = u < v ? v : u;
    edges_out[t - lo] = (a << 32) | b;
    nodes_out[1 + (t - lo)] = v;
  }
  *n_out = (uint32_t)(hi - lo);
  return 0;
}

float orc_device_minibatch_weight(int link, uint64_t N, uint64_t E, uint32_t m) {
  return link ? (float)N : (float)(2 * E) / (float)m;
}
