// Parallel construction of the cuckoo edge set on the device (opt-in).
//
// The reference builds the table on one host thread: random-walk insertion with rand_r (mcmc/cuckoo.cc:117-161),
// 51 s for the 3.3e8 training keys of the N = 10M configuration.  The image that build produces is what parity
// runs use and it stays the default (host/cuckoo.cc).  This is the other way to get a VALID table for the same
// layout and the same hash pair -- [2 buckets][num_bins][4 slots] u64, empty = UINT64_MAX, bucket 0 indexed by
// (P1 k) mod bins, bucket 1 by (k xor P2) mod bins (cuckoo.cc:92-104) -- so that Set_HasEdge / set_has() and every
// kernel that probes the set work unchanged: one thread per key claims a free slot of its bin with atomicCAS; when
// both bins of a key are full it swaps itself into a slot (atomicExch) and carries the evicted key to that key's
// other bin, up to a bound.  Membership is exact; the slot a key ends up in depends on the interleaving.
#include "ammsb_ctx.h"
#include "ammsb_dev.h"

using namespace ammsb;

namespace {

constexpr unsigned long long kEmpty = ~0ull;
constexpr int kMaxWalk = 2000;  // displacements one thread makes before it reports failure

__device__ __forceinline__ uint64_t bin_of(uint64_t k, int bucket, uint32_t prime_idx, uint64_t bins) {
  return bucket == 0 ? (kSetPrimes[2 * prime_idx] * k) % bins : (k ^ kSetPrimes[2 * prime_idx + 1]) % bins;
}

__global__ __launch_bounds__(256) void set_build_kernel(const uint64_t* keys, uint64_t n, unsigned long long* slots,
                                                         uint64_t bins, uint32_t prime_idx, uint32_t* failed) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned long long cur = keys[i];
  if (cur == kEmpty) {  // the empty marker cannot be a member (cuckoo.cc:91)
    atomicAdd(failed, 1u);
    return;
  }
  uint32_t rnd = (uint32_t)(i * 2654435761u) ^ (uint32_t)(cur >> 17);
  int bucket = 0;
  for (int walk = 0; walk < kMaxWalk; ++walk) {
    // a free (or already matching) slot in either bin of the key in hand
    for (int b = 0; b < 2; ++b) {
      const int bb = bucket ^ b;
      unsigned long long* bin = slots + (bb * bins + bin_of(cur, bb, prime_idx, bins)) * 4;
      for (int s = 0; s < 4; ++s) {
        // a plain load may be stale (this CU's L1 is not refreshed by anybody's atomics, its own included): fine as a
        // hint for "is this slot worth a CAS" -- slots only ever go from empty to taken -- but NOT as proof that the
        // key in hand is already a member: a displaced key WAS in the slot it was just swapped out of, and the L1 line
        // this thread loaded before the swap still says so.  Only the thread's own, never yet placed key (walk == 0)
        // may meet a copy of itself (a duplicate in the input).
        const unsigned long long seen = bin[s];
        if (walk == 0 && seen == cur) return;
        if (seen == kEmpty) {
          const unsigned long long old = atomicCAS(&bin[s], kEmpty, cur);
          if (old == kEmpty || (walk == 0 && old == cur)) return;
        }
      }
    }
    // both bins full: take a slot of the current bucket's bin, carry its occupant to the occupant's other bin
    rnd = rnd * 1664525u + 1013904223u;
    unsigned long long* bin = slots + (bucket * bins + bin_of(cur, bucket, prime_idx, bins)) * 4;
    cur = atomicExch(&bin[(rnd >> 16) & 3], cur);
    if (cur == kEmpty) return;  // the slot had just been vacated... by nobody: slots never empty out; kept for safety
    bucket ^= 1;
  }
  atomicAdd(failed, 1u);  // `cur` is homeless: the caller retries with the next hash pair
}

__global__ void set_clear_kernel(unsigned long long* slots, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    slots[i] = kEmpty;
}

}  // namespace

// bins per bucket for n keys, cuckoo.cc:98-104
extern "C" uint64_t ammsb_set_num_bins(uint64_t n) { return 1 + (uint64_t)ceil((1.15 * (double)n) / (2 * 4)); }

extern "C" int ammsb_set_build(ammsb_ctx* ctx, const uint64_t* keys, uint64_t n, uint64_t* slots, uint64_t num_bins,
                               uint32_t* prime_idx_out, uint32_t* scratch, void* stream) {
  AMMSB_CHECK_ARG(ctx, ctx && keys && slots && prime_idx_out && scratch, "null argument");
  AMMSB_CHECK_ARG(ctx, n > 0 && num_bins >= ammsb_set_num_bins(n), "table too small for the key count (cuckoo.cc:98-104)");
  AMMSB_CHECK_ARG(ctx, (n + 255) / 256 <= 0x7fffffffull, "too many keys");
  hipStream_t s = as_stream(stream);
  for (uint32_t p = 0; p < 4; ++p) {  // SetContents tries the four hash pairs in turn, cuckoo.cc:117-129
    set_clear_kernel<<<4096, 256, 0, s>>>(reinterpret_cast<unsigned long long*>(slots), 2 * num_bins * 4);
    AMMSB_HIP(ctx, hipMemsetAsync(scratch, 0, sizeof(uint32_t), s));
    set_build_kernel<<<(uint32_t)((n + 255) / 256), 256, 0, s>>>(keys, n, reinterpret_cast<unsigned long long*>(slots),
                                                                 num_bins, p, scratch);
    AMMSB_LAUNCH_CHECK(ctx);
    uint32_t failed = 0;
    AMMSB_HIP(ctx, hipMemcpyAsync(&failed, scratch, sizeof failed, hipMemcpyDeviceToHost, s));
    AMMSB_HIP(ctx, hipStreamSynchronize(s));  // the answer decides whether another attempt is needed
    if (failed == 0) {
      *prime_idx_out = p;
      return AMMSB_OK;
    }
  }
  snprintf(ctx->err, sizeof ctx->err, "ammsb_set_build: all four hash pairs failed (as Set::SetContents can, cuckoo.cc:117-129)");
  return AMMSB_ERANGE;
}
