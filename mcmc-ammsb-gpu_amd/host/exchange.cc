// mcmc::Exchange (include/mcmc/exchange.h): TCP rendezvous + the host-staged and RCCL transports.
#include "mcmc/exchange.h"

#include <arpa/inet.h>
#include <hip/hip_runtime.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <rccl/rccl.h>
#include <errno.h>
#include <poll.h>
#include <sys/socket.h>
#include <unistd.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace mcmc {

namespace {

[[noreturn]] void Fail(const std::string& what) { throw std::runtime_error("mcmc::Exchange: " + what); }

void HipCheck(hipError_t e, const char* what) {
  if (e != hipSuccess) Fail(std::string(what) + ": " + hipGetErrorString(e));
}

void SendAll(int fd, const void* p, size_t n) {
  const char* c = static_cast<const char*>(p);
  while (n) {
    const ssize_t k = ::send(fd, c, n, MSG_NOSIGNAL);
    if (k <= 0) Fail("socket send failed");
    c += k;
    n -= static_cast<size_t>(k);
  }
}

void RecvAll(int fd, void* p, size_t n) {
  char* c = static_cast<char*>(p);
  while (n) {
    const ssize_t k = ::recv(fd, c, n, 0);
    if (k <= 0) Fail("socket recv failed (peer gone?)");
    c += k;
    n -= static_cast<size_t>(k);
  }
}

int EnvInt(const char* name, int def) {
  const char* v = getenv(name);
  return v && *v ? atoi(v) : def;
}

// Star through rank 0: rank 0 holds one socket per peer (index = peer rank), a peer holds one socket to rank 0.
struct Star {
  int rank = 0, world = 1;
  std::vector<int> fds;  // rank 0: [world] (fds[0] unused); others: [1]

  void Connect() {
    rank = EnvInt("RANK", 0);
    world = EnvInt("WORLD_SIZE", 1);
    if (world < 1 || rank < 0 || rank >= world) Fail("bad RANK / WORLD_SIZE");
    if (world == 1) return;
    const char* addr_env = getenv("MASTER_ADDR");
    const std::string addr = addr_env && *addr_env ? addr_env : "127.0.0.1";
    const int port = EnvInt("AMMSB_EXCHANGE_PORT", EnvInt("MASTER_PORT", 29531) + 1);  // next to torchrun's own port
    sockaddr_in sa;
    memset(&sa, 0, sizeof sa);
    sa.sin_family = AF_INET;
    sa.sin_port = htons(static_cast<uint16_t>(port));
    if (inet_pton(AF_INET, addr == "localhost" ? "127.0.0.1" : addr.c_str(), &sa.sin_addr) != 1) Fail("MASTER_ADDR must be an IPv4 address");
    const int one = 1;
    if (rank == 0) {
      const int ls = ::socket(AF_INET, SOCK_STREAM, 0);
      if (ls < 0) Fail("socket()");
      setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
      // listen on MASTER_ADDR only (loopback for a single node): nothing outside the job's own network can connect.
      // AMMSB_EXCHANGE_BIND names another local address to listen on ("0.0.0.0" = every interface); and when
      // MASTER_ADDR is reachable by the peers without being an address of THIS host (a NAT / service address, a
      // forwarded container port: EADDRNOTAVAIL) the listener falls back to every interface, saying so.
      sockaddr_in la = sa;
      if (const char* b = getenv("AMMSB_EXCHANGE_BIND")) {
        if (*b && inet_pton(AF_INET, b, &la.sin_addr) != 1) Fail("AMMSB_EXCHANGE_BIND must be an IPv4 address");
      }
      int brc = ::bind(ls, reinterpret_cast<sockaddr*>(&la), sizeof la);
      if (brc != 0 && errno == EADDRNOTAVAIL) {
        fprintf(stderr, "W mcmc::Exchange: %s is not an address of this host; listening on every interface (port %d)\n",
                addr.c_str(), port);
        la.sin_addr.s_addr = htonl(INADDR_ANY);
        brc = ::bind(ls, reinterpret_cast<sockaddr*>(&la), sizeof la);
      }
      if (brc != 0) {
        const int err = errno;
        Fail("bind() to " + addr + ":" + std::to_string(port) + " failed: " + strerror(err) +
             (err == EADDRINUSE ? " (MASTER_PORT + 1 taken? pass a free AMMSB_EXCHANGE_PORT to every rank)" : ""));
      }
      if (::listen(ls, world) != 0) Fail("listen()");
      fds.assign(world, -1);
      // the peers retry for a minute; so does this side: a rank that died before connecting must not hang rank 0
      const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(EnvInt("AMMSB_EXCHANGE_TIMEOUT_S", 60));
      for (int got = 1; got < world;) {
        const auto left = std::chrono::duration_cast<std::chrono::milliseconds>(deadline - std::chrono::steady_clock::now()).count();
        if (left <= 0) Fail("rendezvous timed out: " + std::to_string(world - got) + " rank(s) never connected");
        pollfd pf = {ls, POLLIN, 0};
        const int pr = ::poll(&pf, 1, static_cast<int>(left < 1000 ? left : 1000));
        if (pr < 0 && errno != EINTR) Fail("poll()");
        if (pr <= 0) continue;
        const int fd = ::accept(ls, nullptr, nullptr);
        if (fd < 0) continue;
        setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
        timeval tv = {5, 0};  // a connection that says nothing for 5 s is not one of ours
        setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
        int peer = -1;
        const ssize_t k = ::recv(fd, &peer, sizeof peer, MSG_WAITALL);
        if (k != static_cast<ssize_t>(sizeof peer) || peer <= 0 || peer >= world || fds[peer] != -1) {
          ::close(fd);  // a stray connection (port scanner, a rank of another job): ignore it, keep waiting
          continue;
        }
        timeval none = {0, 0};
        setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &none, sizeof none);
        fds[peer] = fd;
        ++got;
      }
      ::close(ls);
    } else {
      int fd = -1;
      for (int attempt = 0; attempt < 600; ++attempt) {  // rank 0 may still be starting: retry for a minute
        fd = ::socket(AF_INET, SOCK_STREAM, 0);
        if (fd < 0) Fail("socket()");
        if (::connect(fd, reinterpret_cast<sockaddr*>(&sa), sizeof sa) == 0) break;
        ::close(fd);
        fd = -1;
        std::this_thread::sleep_for(std::chrono::milliseconds(100));
      }
      if (fd < 0) Fail("cannot reach rank 0 at " + addr + ":" + std::to_string(port));
      setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
      SendAll(fd, &rank, sizeof rank);
      fds.assign(1, fd);
    }
  }

  ~Star() {
    for (int fd : fds)
      if (fd >= 0) ::close(fd);
  }

  // every rank contributes `bytes`; on return all[r * bytes ..] holds rank r's contribution everywhere
  void AllGatherHost(const void* mine, void* all, size_t bytes) {
    char* a = static_cast<char*>(all);
    if (world == 1) {
      memcpy(a, mine, bytes);
      return;
    }
    if (rank == 0) {
      memcpy(a, mine, bytes);
      for (int r = 1; r < world; ++r) RecvAll(fds[r], a + r * bytes, bytes);
      for (int r = 1; r < world; ++r) SendAll(fds[r], a, bytes * world);
    } else {
      SendAll(fds[0], mine, bytes);
      RecvAll(fds[0], a, bytes * world);
    }
  }

  void BroadcastHost(void* buf, size_t bytes, int root) {
    if (world == 1) return;
    if (rank == 0) {
      if (root != 0) RecvAll(fds[root], buf, bytes);
      for (int r = 1; r < world; ++r)
        if (r != root) SendAll(fds[r], buf, bytes);
    } else if (rank == root) {
      SendAll(fds[0], buf, bytes);
    } else {
      RecvAll(fds[0], buf, bytes);
    }
  }
};

class HostExchange : public Exchange {
 public:
  HostExchange() { star_.Connect(); }
  int rank() const override { return star_.rank; }
  int world() const override { return star_.world; }
  const char* kind() const override { return "host"; }

  void AllGatherInPlace(void* dev_region, size_t chunk_bytes, void* stream) override {
    if (world() == 1) return;
    hipStream_t s = static_cast<hipStream_t>(stream);
    HipCheck(hipStreamSynchronize(s), "hipStreamSynchronize");
    std::vector<char> mine(chunk_bytes), all(chunk_bytes * world());
    char* region = static_cast<char*>(dev_region);
    HipCheck(hipMemcpy(mine.data(), region + rank() * chunk_bytes, chunk_bytes, hipMemcpyDeviceToHost), "hipMemcpy D2H");
    star_.AllGatherHost(mine.data(), all.data(), chunk_bytes);
    for (int r = 0; r < world(); ++r)
      if (r != rank())
        HipCheck(hipMemcpy(region + r * chunk_bytes, all.data() + r * chunk_bytes, chunk_bytes, hipMemcpyHostToDevice), "hipMemcpy H2D");
  }

  void Broadcast(void* dev, size_t bytes, int root, void* stream) override {
    if (world() == 1) return;
    HipCheck(hipStreamSynchronize(static_cast<hipStream_t>(stream)), "hipStreamSynchronize");
    std::vector<char> buf(bytes);
    if (rank() == root) HipCheck(hipMemcpy(buf.data(), dev, bytes, hipMemcpyDeviceToHost), "hipMemcpy D2H");
    star_.BroadcastHost(buf.data(), bytes, root);
    if (rank() != root) HipCheck(hipMemcpy(dev, buf.data(), bytes, hipMemcpyHostToDevice), "hipMemcpy H2D");
  }

  void AllGather(const void* dev_local, void* dev_all, size_t bytes, void* stream) override {
    HipCheck(hipStreamSynchronize(static_cast<hipStream_t>(stream)), "hipStreamSynchronize");
    std::vector<char> mine(bytes), all(bytes * world());
    HipCheck(hipMemcpy(mine.data(), dev_local, bytes, hipMemcpyDeviceToHost), "hipMemcpy D2H");
    star_.AllGatherHost(mine.data(), all.data(), bytes);
    HipCheck(hipMemcpy(dev_all, all.data(), bytes * world(), hipMemcpyHostToDevice), "hipMemcpy H2D");
  }

  void Barrier() override {
    char c = 0;
    std::vector<char> all(world());
    star_.AllGatherHost(&c, all.data(), 1);
  }
  void AllGatherHost(const void* mine, void* all, size_t bytes) override { star_.AllGatherHost(mine, all, bytes); }
  void BroadcastHost(void* buf, size_t bytes, int root) override { star_.BroadcastHost(buf, bytes, root); }

 private:
  Star star_;
};

class RcclExchange : public Exchange {
 public:
  explicit RcclExchange(int device) {
    star_.Connect();
    HipCheck(hipSetDevice(device), "hipSetDevice");
    ncclUniqueId id;
    memset(&id, 0, sizeof id);
    if (star_.rank == 0) Nccl(ncclGetUniqueId(&id), "ncclGetUniqueId");
    star_.BroadcastHost(&id, sizeof id, 0);
    Nccl(ncclCommInitRank(&comm_, star_.world, id, star_.rank), "ncclCommInitRank");
    const char* form = getenv("AMMSB_EXCHANGE_FORM");
    direct_ = form && std::string(form) == "p2p";
  }
  ~RcclExchange() override {
    if (comm_) ncclCommDestroy(comm_);
  }
  int rank() const override { return star_.rank; }
  int world() const override { return star_.world; }
  const char* kind() const override { return "rccl"; }

  void AllGatherInPlace(void* dev_region, size_t chunk_bytes, void* stream) override {
    if (direct_) {
      // the direct form: this rank's chunk to every peer and theirs back as one group of point-to-point
      // operations -- RCCL runs them concurrently, one per xGMI link of a fully connected node (a ring is bound by
      // one link's rate).  AMMSB_EXCHANGE_FORM=p2p; the Python learner times both forms at start-up.
      char* region = static_cast<char*>(dev_region);
      hipStream_t s = static_cast<hipStream_t>(stream);
      Nccl(ncclGroupStart(), "ncclGroupStart");
      for (int d = 1; d < star_.world; ++d) {
        const int to = (star_.rank + d) % star_.world, from = (star_.rank - d + star_.world) % star_.world;
        Nccl(ncclSend(region + star_.rank * chunk_bytes, chunk_bytes, ncclChar, to, comm_, s), "ncclSend");
        Nccl(ncclRecv(region + from * chunk_bytes, chunk_bytes, ncclChar, from, comm_, s), "ncclRecv");
      }
      Nccl(ncclGroupEnd(), "ncclGroupEnd");
      return;
    }
    // in place: the send buffer is this rank's slot of the receive buffer (each chunk moves once)
    Nccl(ncclAllGather(static_cast<char*>(dev_region) + star_.rank * chunk_bytes, dev_region, chunk_bytes, ncclChar, comm_,
                       static_cast<hipStream_t>(stream)),
         "ncclAllGather");
  }
  void Broadcast(void* dev, size_t bytes, int root, void* stream) override {
    Nccl(ncclBroadcast(dev, dev, bytes, ncclChar, root, comm_, static_cast<hipStream_t>(stream)), "ncclBroadcast");
  }
  void AllGather(const void* dev_local, void* dev_all, size_t bytes, void* stream) override {
    Nccl(ncclAllGather(dev_local, dev_all, bytes, ncclChar, comm_, static_cast<hipStream_t>(stream)), "ncclAllGather");
  }
  void Barrier() override {
    char c = 0;
    std::vector<char> all(world());
    star_.AllGatherHost(&c, all.data(), 1);
  }
  void AllGatherHost(const void* mine, void* all, size_t bytes) override { star_.AllGatherHost(mine, all, bytes); }
  void BroadcastHost(void* buf, size_t bytes, int root) override { star_.BroadcastHost(buf, bytes, root); }

 private:
  static void Nccl(ncclResult_t r, const char* what) {
    if (r != ncclSuccess) Fail(std::string(what) + ": " + ncclGetErrorString(r));
  }
  Star star_;
  ncclComm_t comm_ = nullptr;
  bool direct_ = false;
};

}  // namespace

std::shared_ptr<Exchange> Exchange::FromEnvironment(const std::string& kind, int device) {
  if (kind == "host") return std::shared_ptr<Exchange>(new HostExchange());
  if (kind == "rccl") return std::shared_ptr<Exchange>(new RcclExchange(device));
  Fail("unknown exchange kind '" + kind + "' (rccl | host)");
}

}  // namespace mcmc
