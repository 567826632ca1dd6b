// Host cuckoo set.  Behavioural restatement of the reference's mcmc/cuckoo.cc:92-220; the table is
// kept flat in the serialised layout so that uploading it is one copy.
#include "mcmc/cuckoo.h"

#include <cmath>
#include <cstdlib>
#include <limits>

namespace mcmc {
namespace cuckoo {

const Edge Set::KEY_INVALID = std::numeric_limits<Edge>::max();

namespace {
// (multiplier for bucket 0, xor mask for bucket 1), tried in this order (cuckoo.cc:92-96)
const uint64_t kPrimes[4][2] = {{15485807ull, 920429591ull},
                                {379906717ull, 740320571ull},
                                {256204747ull, 379927517ull},
                                {13ull, 17ull}};
}  // namespace

Set::Set(size_t n)
    : count_(0),
      bins_(static_cast<size_t>(1 + std::ceil((1.15 * n) / (NUM_BUCKETS * NUM_SLOTS)))),
      seed_(42),
      displacements_max_(n / 2 + 1),
      prime_idx_(0) {}

size_t Set::Hash(Edge k, size_t bucket) const {
  return bucket == 0 ? (kPrimes[prime_idx_][0] * k) % bins_ : (k ^ kPrimes[prime_idx_][1]) % bins_;
}

// Put k into the first free slot of `bin`; if the bin is full evict a rand_r-chosen resident and
// return it (KEY_INVALID when nothing was evicted).
Edge Set::Place(Edge k, Edge* bin) {
  for (size_t s = 0; s < NUM_SLOTS; ++s) {
    if (bin[s] == KEY_INVALID) {
      bin[s] = k;
      return KEY_INVALID;
    }
  }
  const size_t victim = rand_r(&seed_) % NUM_SLOTS;
  const Edge old = bin[victim];
  bin[victim] = k;
  return old;
}

bool Set::Insert(Edge k) {
  size_t moves = 0;
  do {
    for (size_t b = 0; b < NUM_BUCKETS; ++b) {
      Edge* bin = Bin(b, Hash(k, b));
      bool has_room = false, present = false;
      for (size_t s = 0; s < NUM_SLOTS; ++s) {
        has_room |= (bin[s] == KEY_INVALID);
        present |= (bin[s] == k);
      }
      if (has_room && !present) {
        Place(k, bin);
        ++count_;
        return true;
      }
    }
    const size_t b = rand_r(&seed_) % NUM_BUCKETS;
    k = Place(k, Bin(b, Hash(k, b)));
  } while (++moves < displacements_max_);
  return false;
}

bool Set::SetContents(const Edge* start, const Edge* end) {
  // seed_ and count_ carry over between attempts, as in the reference (cuckoo.cc:117-129)
  for (prime_idx_ = 0; prime_idx_ < 4; ++prime_idx_) {
    table_.assign(Capacity(), KEY_INVALID);
    bool ok = true;
    for (const Edge* it = start; ok && it != end; ++it) ok = Insert(*it);
    if (ok) return true;
  }
  return false;
}

bool Set::SetContents(std::vector<Edge>::const_iterator start, std::vector<Edge>::const_iterator end) {
  const Edge* b = start == end ? nullptr : &*start;
  return SetContents(b, b + (end - start));
}

bool Set::Has(Edge k) const {
  if (table_.empty()) return false;
  for (size_t b = 0; b < NUM_BUCKETS; ++b) {
    const Edge* bin = Bin(b, Hash(k, b));
    for (size_t s = 0; s < NUM_SLOTS; ++s)
      if (bin[s] == k) return true;
  }
  return false;
}

}  // namespace cuckoo
}  // namespace mcmc
