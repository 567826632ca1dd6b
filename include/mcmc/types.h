// Basic value types of the mcmc:: API (drop-in for the reference's mcmc/types.h:31-74).
// The kernel-language shim that makes up the rest of the reference's types.{h,cc} (macro
// generators for OpenCL/NVRTC text) has no counterpart: the gfx950 kernels are compiled ahead of
// time from C++.
#ifndef MCMC_AMD_TYPES_H_
#define MCMC_AMD_TYPES_H_

#include <cstddef>
#include <cstdint>
#include <initializer_list>
#include <iosfwd>
#include <string>
#include <tuple>
#include <vector>

namespace mcmc {

typedef uint64_t Edge;    // (u << 32) | v with u = min, v = max
typedef uint32_t Vertex;
typedef float Float;

// seed of one xorshift128+ stream family; same shape as the reference's ulong2 (types.h:35-54)
struct ulong2 {
  uint64_t values[2];
  uint64_t& operator[](size_t i) { return values[i]; }
  uint64_t operator[](size_t i) const { return values[i]; }
  ulong2() : values{0, 0} {}
  ulong2(std::initializer_list<uint64_t> l) : values{*l.begin(), *(l.begin() + 1)} {}
  ulong2& operator=(std::initializer_list<uint64_t> l) {
    values[0] = *l.begin();
    values[1] = *(l.begin() + 1);
    return *this;
  }
} __attribute__((aligned(16)));

std::ostream& operator<<(std::ostream& out, const ulong2& v);
// parses "a,b" (the reference reads into v[2] here, types.cc:551 -- a defect that is not replicated)
std::istream& operator>>(std::istream& in, ulong2& v);

inline std::tuple<Vertex, Vertex> Vertices(Edge e) {
  return std::make_tuple(static_cast<Vertex>(e >> 32), static_cast<Vertex>(e & 0xffffffffull));
}

inline Edge MakeEdge(Vertex u, Vertex v) { return (static_cast<Edge>(u) << 32) | static_cast<Edge>(v); }

inline uint32_t GetMaxGroups() { return 65535; }  // types.cc:537

}  // namespace mcmc

#endif  // MCMC_AMD_TYPES_H_
