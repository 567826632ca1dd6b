// Host cuckoo edge set: same public interface and same table image as the reference's
// mcmc::cuckoo::Set (mcmc/cuckoo.h:16-67, cuckoo.cc:92-220): 2 buckets x N_ bins x 4 slots,
// N_ = 1 + ceil(1.15 n / 8), four (multiplier, xor) prime pairs tried in turn, random-walk eviction
// driven by rand_r(seed_ = 42).  Serialize() is the image ammsb_set.slots expects.
#ifndef MCMC_AMD_CUCKOO_H_
#define MCMC_AMD_CUCKOO_H_

#include <array>
#include <memory>
#include <vector>

#include "mcmc/types.h"

namespace mcmc {
namespace cuckoo {

class Set {
 public:
  static const size_t NUM_BUCKETS = 2;
  static const size_t NUM_SLOTS = 4;
  static const Edge KEY_INVALID;

  explicit Set(size_t n);

  bool SetContents(std::vector<Edge>::const_iterator start, std::vector<Edge>::const_iterator end);
  bool SetContents(const Edge* start, const Edge* end);

  bool Has(Edge k) const;

  size_t BinsPerBucket() const { return bins_; }
  size_t Capacity() const { return bins_ * NUM_SLOTS * NUM_BUCKETS; }
  size_t Size() const { return count_; }
  uint32_t PrimeIdx() const { return prime_idx_; }

  // [bucket][bin][slot], empty slots = KEY_INVALID
  std::vector<Edge> Serialize() const { return table_; }
  const Edge* Data() const { return table_.data(); }

 private:
  Edge* Bin(size_t bucket, size_t h) { return table_.data() + (bucket * bins_ + h) * NUM_SLOTS; }
  const Edge* Bin(size_t bucket, size_t h) const { return table_.data() + (bucket * bins_ + h) * NUM_SLOTS; }
  size_t Hash(Edge k, size_t bucket) const;
  bool Insert(Edge k);
  Edge Place(Edge k, Edge* bin);

  size_t count_;
  const size_t bins_;
  std::vector<Edge> table_;
  unsigned int seed_;
  const size_t displacements_max_;
  uint32_t prime_idx_;
};

}  // namespace cuckoo
}  // namespace mcmc

#endif  // MCMC_AMD_CUCKOO_H_
