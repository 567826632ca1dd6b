// Multi-GPU exchange for the C++ mcmc::Learner (new: the reference is single-device).
//
// One process per GPU.  The learner shards the mini-batch (pi, phi_sum, theta / beta and both sets are
// replicated; rank r computes a block of virtual groups of update_phi, a slice of the gradient's edges, a slice of
// the held-out edges) and needs three things from the fabric, all on device memory and ordered on a HIP stream:
//   - AllGatherInPlace: every rank's block of phi_vec rows to every rank (the bandwidth-bound step);
//   - Broadcast: one owner's rows to the others (link batches live in rank 0's block; tail rows);
//   - AllGather of a few words per rank (the [2K] gradient partials, the 4 perplexity sums).
// Two implementations behind one interface:
//   "rccl"  RCCL over xGMI (ncclAllGather / ncclBroadcast on the caller's stream) -- the production path;
//   "host"  the same calls staged through host memory and TCP sockets: a rehearsal transport for boxes with fewer
//           GPUs than ranks (several ranks may share one device, which RCCL refuses: "Duplicate GPU detected"),
//           used by the tests.  Never for numbers.
// Rendezvous (both): rank 0 listens on MASTER_ADDR:MASTER_PORT (default 127.0.0.1:29531), the others connect;
// RANK / WORLD_SIZE as torchrun exports them.
#ifndef MCMC_AMD_EXCHANGE_H_
#define MCMC_AMD_EXCHANGE_H_

#include <cstddef>
#include <memory>
#include <string>

namespace mcmc {

class Exchange {
 public:
  virtual ~Exchange() {}
  virtual int rank() const = 0;
  virtual int world() const = 0;
  virtual const char* kind() const = 0;
  // region = world() consecutive chunks of chunk_bytes; on return chunk r holds rank r's chunk everywhere
  virtual void AllGatherInPlace(void* dev_region, size_t chunk_bytes, void* stream) = 0;
  virtual void Broadcast(void* dev, size_t bytes, int root, void* stream) = 0;
  // dev_all = world() * bytes; slot r <- rank r's dev_local
  virtual void AllGather(const void* dev_local, void* dev_all, size_t bytes, void* stream) = 0;
  virtual void Barrier() = 0;
  // the same two shapes on host memory, over the rendezvous sockets (control plane: agreeing on a value, tests)
  virtual void AllGatherHost(const void* mine, void* all, size_t bytes) = 0;
  virtual void BroadcastHost(void* buf, size_t bytes, int root) = 0;

  // kind: "rccl" | "host".  Reads RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT; device = the HIP device this rank uses.
  // Throws std::runtime_error on rendezvous / RCCL failures.
  static std::shared_ptr<Exchange> FromEnvironment(const std::string& kind, int device);
};

}  // namespace mcmc

#endif  // MCMC_AMD_EXCHANGE_H_
