// Unit test of mcmc::Exchange (include/mcmc/exchange.h), one process per rank (RANK / WORLD_SIZE / MASTER_* from
// the environment).  argv[1] = kind (host | rccl), argv[2] = "hostmem" (sockets only: runs without a GPU) or
// "device" (the three device collectives on hipMalloc'ed buffers).  Prints "OK <rank>" and exits 0 on success.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "mcmc/exchange.h"

namespace {
uint8_t Pattern(int rank, size_t i, int salt) { return static_cast<uint8_t>((rank * 131 + i * 7 + salt * 29 + (i >> 8)) & 0xFF); }

#define REQUIRE(c)                                                                   \
  do {                                                                               \
    if (!(c)) {                                                                      \
      std::fprintf(stderr, "FAILED %s (%s:%d)\n", #c, __FILE__, __LINE__);           \
      std::exit(1);                                                                  \
    }                                                                                \
  } while (0)

void HostMem(mcmc::Exchange& x) {
  const int R = x.world(), r = x.rank();
  int salt = 0;
  for (size_t bytes : {size_t(1), size_t(24), size_t(1000), size_t(1) << 20, (size_t(3) << 20) + 5}) {
    ++salt;
    std::vector<uint8_t> mine(bytes), all(bytes * R, 0xEE);
    for (size_t i = 0; i < bytes; ++i) mine[i] = Pattern(r, i, salt);
    x.AllGatherHost(mine.data(), all.data(), bytes);
    for (int q = 0; q < R; ++q)
      for (size_t i = 0; i < bytes; ++i) REQUIRE(all[q * bytes + i] == Pattern(q, i, salt));
    for (int root = 0; root < R; ++root) {
      std::vector<uint8_t> buf(bytes, 0xEE);
      if (r == root)
        for (size_t i = 0; i < bytes; ++i) buf[i] = Pattern(root, i, salt + 100);
      x.BroadcastHost(buf.data(), bytes, root);
      for (size_t i = 0; i < bytes; ++i) REQUIRE(buf[i] == Pattern(root, i, salt + 100));
    }
    x.Barrier();
  }
}

void Hip(hipError_t e) {
  if (e != hipSuccess) {
    std::fprintf(stderr, "HIP error: %s\n", hipGetErrorString(e));
    std::exit(1);
  }
}

void Device(mcmc::Exchange& x) {
  const int R = x.world(), r = x.rank();
  hipStream_t stream;
  Hip(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  int salt = 0;
  for (size_t bytes : {size_t(16), size_t(4096), (size_t(5) << 20) + 64}) {
    ++salt;
    // all-gather in place: world chunks, only this rank's chunk is valid before the call
    std::vector<uint8_t> h(bytes * R, 0xEE);
    for (size_t i = 0; i < bytes; ++i) h[r * bytes + i] = Pattern(r, i, salt);
    uint8_t *d_region, *d_local, *d_all;
    Hip(hipMalloc(&d_region, bytes * R));
    Hip(hipMalloc(&d_local, bytes));
    Hip(hipMalloc(&d_all, bytes * R));
    Hip(hipMemcpyAsync(d_region, h.data(), bytes * R, hipMemcpyHostToDevice, stream));
    x.AllGatherInPlace(d_region, bytes, stream);
    Hip(hipMemcpyAsync(h.data(), d_region, bytes * R, hipMemcpyDeviceToHost, stream));
    Hip(hipStreamSynchronize(stream));
    for (int q = 0; q < R; ++q)
      for (size_t i = 0; i < bytes; ++i) REQUIRE(h[q * bytes + i] == Pattern(q, i, salt));
    // out-of-place all-gather
    Hip(hipMemcpyAsync(d_local, h.data() + r * bytes, bytes, hipMemcpyHostToDevice, stream));
    Hip(hipMemsetAsync(d_all, 0xEE, bytes * R, stream));
    x.AllGather(d_local, d_all, bytes, stream);
    std::vector<uint8_t> g(bytes * R);
    Hip(hipMemcpyAsync(g.data(), d_all, bytes * R, hipMemcpyDeviceToHost, stream));
    Hip(hipStreamSynchronize(stream));
    REQUIRE(g == h);
    // broadcast from every root
    for (int root = 0; root < R; ++root) {
      std::vector<uint8_t> b(bytes, 0xEE);
      if (r == root)
        for (size_t i = 0; i < bytes; ++i) b[i] = Pattern(root, i, salt + 50);
      Hip(hipMemcpyAsync(d_local, b.data(), bytes, hipMemcpyHostToDevice, stream));
      x.Broadcast(d_local, bytes, root, stream);
      Hip(hipMemcpyAsync(b.data(), d_local, bytes, hipMemcpyDeviceToHost, stream));
      Hip(hipStreamSynchronize(stream));
      for (size_t i = 0; i < bytes; ++i) REQUIRE(b[i] == Pattern(root, i, salt + 50));
    }
    Hip(hipFree(d_region));
    Hip(hipFree(d_local));
    Hip(hipFree(d_all));
    x.Barrier();
  }
  Hip(hipStreamDestroy(stream));
}
}  // namespace

int main(int argc, char** argv) {
  if (argc < 3) {
    std::cerr << "usage: exchange_test host|rccl hostmem|device" << std::endl;
    return 2;
  }
  const std::string kind = argv[1], mode = argv[2];
  try {
    if (mode == "device") Hip(hipSetDevice(0));
    std::shared_ptr<mcmc::Exchange> x = mcmc::Exchange::FromEnvironment(kind, 0);
    if (mode == "hostmem")
      HostMem(*x);
    else
      Device(*x);
    std::cout << "OK " << x->rank() << " of " << x->world() << " " << x->kind() << std::endl;
  } catch (const std::exception& e) {
    std::cerr << "exception: " << e.what() << std::endl;
    return 1;
  }
  return 0;
}
