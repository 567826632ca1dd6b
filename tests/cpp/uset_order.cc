// Test helper (built on the fly by tests/test_oracle_samplers.py): the REAL libstdc++ container the reference's
// samplers iterate over, to pin the plain-C restatement in oracle/ammsb_oracle_samplers.c.
#include <cstdint>
#include <unordered_set>
extern "C" uint64_t real_uset_order_u64(const uint64_t* keys, uint64_t n, uint64_t* out) {
  std::unordered_set<uint64_t> s;
  for (uint64_t i = 0; i < n; ++i) s.insert(keys[i]);
  uint64_t c = 0;
  for (uint64_t k : s) out[c++] = k;
  return c;
}
extern "C" uint64_t real_uset_order_u32(const uint32_t* keys, uint64_t n, uint32_t* out) {
  std::unordered_set<uint32_t> s;
  for (uint64_t i = 0; i < n; ++i) s.insert(keys[i]);
  uint64_t c = 0;
  for (uint32_t k : s) out[c++] = k;
  return c;
}
