// Host mini-batch edge samplers: drop-in for the reference's mcmc/sample.h:94-123 (free functions,
// same names, same argument meaning, same return value = the mini-batch weight `scale` that
// BetaUpdater receives).  They consume glibc rand_r() and iterate std::unordered_set exactly as the
// reference does, so that with the same libc / libstdc++ a given seed yields the same mini-batch.
#ifndef MCMC_AMD_SAMPLE_H_
#define MCMC_AMD_SAMPLE_H_

#include <iosfwd>
#include <string>
#include <vector>

#include "mcmc/types.h"

namespace mcmc {

struct Config;

enum SampleStrategy { Node, NodeLink, NodeNonLink, BFLink, BFNonLink, BF };

std::string to_string(const SampleStrategy& s);
std::istream& operator>>(std::istream& in, SampleStrategy& strategy);

Float sampleBreadthFirstLink(const Config& cfg, std::vector<Edge>* edges, unsigned int* seed);
Float sampleBreadthFirstNonLink(const Config& cfg, std::vector<Edge>* edges, unsigned int* seed);
Float sampleBreadthFirst(const Config& cfg, std::vector<Edge>* edges, unsigned int* seed);
Float sampleNodeLink(const Config& cfg, std::vector<Edge>* edges, unsigned int* seed);
Float sampleNodeNonLink(const Config& cfg, std::vector<Edge>* edges, unsigned int* seed);
Float sampleNode(const Config& cfg, std::vector<Edge>* edges, unsigned int* seed);

typedef Float (*SamplerFn)(const Config& cfg, std::vector<Edge>* edges, unsigned int* seed);
SamplerFn GetSampler(SampleStrategy s);  // learner.cc:126-147

// unique end points of the mini-batch in std::unordered_set iteration order (learner.cc:162-173)
void ExtractNodesFromMiniBatch(const std::vector<Edge>& edges, std::vector<Vertex>* nodes);

}  // namespace mcmc

#endif  // MCMC_AMD_SAMPLE_H_
