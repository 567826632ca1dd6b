/*
 * ammsb_host.h -- C view of the host-side data structures (libammsb_host.so) for callers that cannot
 * include the C++ headers in include/mcmc/ (the Python harness, other FFIs).  Host memory only; no
 * device work happens behind these calls.  Handles are opaque; every function returns 0 / a count
 * on success and a negative value on failure unless stated otherwise.
 */
#ifndef AMMSB_HOST_H
#define AMMSB_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ammsb_host_set ammsb_host_set;       /* mcmc::cuckoo::Set, mcmc/cuckoo.h:16-67 */
typedef struct ammsb_host_graph ammsb_host_graph;   /* mcmc::Graph, mcmc/data.h:16-35 */
typedef struct ammsb_host_dataset ammsb_host_dataset; /* training/held-out split + both graphs (a Config's data half) */

/* cuckoo set: Set(n) + SetContents(); NULL if all four prime pairs fail */
ammsb_host_set* ammsb_host_set_create(const uint64_t* keys, uint64_t n);
void ammsb_host_set_destroy(ammsb_host_set* s);
uint64_t ammsb_host_set_bins(const ammsb_host_set* s);
uint32_t ammsb_host_set_prime_idx(const ammsb_host_set* s);
uint64_t ammsb_host_set_size(const ammsb_host_set* s);
const uint64_t* ammsb_host_set_data(const ammsb_host_set* s); /* Serialize() image, 2*bins*4 keys */
int ammsb_host_set_has(const ammsb_host_set* s, const uint64_t* keys, uint64_t n, uint8_t* out);

/* synthetic a-MMSB graph; returns the number of unique edges written to *edges (malloc'd; free with
 * ammsb_host_free) */
int64_t ammsb_host_generate_graph(uint64_t N, uint32_t K_true, double avg_degree, uint64_t seed, uint64_t** edges);
void ammsb_host_free(void* p);

/* SNAP text loader / gzip data-set files (data.cc:36-78, main.cc:109-143) */
int64_t ammsb_host_load_snap(const char* path, uint64_t* N, uint64_t** edges);
int ammsb_host_dump_dataset(const char* path, uint64_t N, float heldout_ratio, const uint64_t* edges, uint64_t n);
int64_t ammsb_host_load_dataset(const char* path, uint64_t* N, float* heldout_ratio, uint64_t** edges);

/* GenerateSetsFromEdges + Graph x2 (main.cc:102-151); srand(rand_seed) is called first so that the
 * fake held-out pairs are reproducible */
ammsb_host_dataset* ammsb_host_dataset_create(uint64_t N, const uint64_t* edges, uint64_t n, double heldout_ratio,
                                              unsigned rand_seed);
void ammsb_host_dataset_destroy(ammsb_host_dataset* d);
uint64_t ammsb_host_dataset_num_training(const ammsb_host_dataset* d);
uint64_t ammsb_host_dataset_num_heldout(const ammsb_host_dataset* d);
const uint64_t* ammsb_host_dataset_training_edges(const ammsb_host_dataset* d);
const uint64_t* ammsb_host_dataset_heldout_edges(const ammsb_host_dataset* d);
const ammsb_host_set* ammsb_host_dataset_training_set(const ammsb_host_dataset* d);
const ammsb_host_set* ammsb_host_dataset_heldout_set(const ammsb_host_dataset* d);
uint64_t ammsb_host_dataset_max_fan_out(const ammsb_host_dataset* d);
/* CSR of the training graph: offsets[N+1], targets[2*num_training] (caller-allocated) */
int ammsb_host_dataset_training_csr(const ammsb_host_dataset* d, uint64_t* offsets, uint32_t* targets);

/* theta_0 of Learner::Learner (learner.cc:150-153): 2K draws of std::gamma_distribution<float>(eta0,
 * eta1) from std::mt19937(6342455113) */
int ammsb_host_theta_init(uint64_t K, float eta0, float eta1, float* theta_out);

/* MakeEdgesForTrainingPerplexity (learner.cc:47-75): the first ratio * |training| training edges followed by
 * links * (N(N-1)/2) / E sampled non-links (rand_r stream from `seed`).  *out is malloc'd (ammsb_host_free).
 * Returns the number of edges or -1 (too many for one launch / bad arguments). */
int64_t ammsb_host_train_ppx_edges(const ammsb_host_dataset* d, uint64_t N, uint64_t E, float ratio, unsigned seed,
                                   uint64_t** out);

/* host mini-batch sampling (sample.cc:177-303 + learner.cc:162-173).  strategy: 0 Node, 1 NodeLink,
 * 2 NodeNonLink, 3 BFLink, 4 BFNonLink, 5 BF.  *seed is the rand_r state (Sample::seed).  edges_out
 * must hold max(mini_batch, max_fan_out) keys, nodes_out max(2*mini_batch, 1+max_fan_out) ids.
 * edges_cap / nodes_cap are the capacities of the two arrays: a mini-batch that does not fit is NOT copied; the call
 * returns -2 with the needed sizes in n_edges / n_nodes (the reference aborts with "N | cap", learner.cc:184-189).
 * Returns 0 on success; writes counts and the mini-batch weight. */
int ammsb_host_sample(const ammsb_host_dataset* d, uint64_t N, uint64_t E, uint64_t mini_batch, int strategy,
                      unsigned* seed, uint64_t* edges_out, uint64_t edges_cap, uint64_t* n_edges, uint32_t* nodes_out,
                      uint64_t nodes_cap, uint64_t* n_nodes, float* weight);

#ifdef __cplusplus
}
#endif
#endif
