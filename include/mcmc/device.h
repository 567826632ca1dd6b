// Minimal CLCudaAPI-shaped facade over the HIP runtime (the reference programs against
// CLCudaAPI's clcuda::{Platform, Device, Context, Queue, Event, Buffer<T>}, mcmc/types.h:11-29), so
// that callers written like the reference's main.cc / tests compile against this library unchanged.
// Only what the hot path needs exists: there is no Program/Kernel (kernels are ahead-of-time gfx950).
#ifndef MCMC_AMD_DEVICE_H_
#define MCMC_AMD_DEVICE_H_

#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace mcmc {
namespace clcuda {

void Check(int hip_error, const char* what);  // throws std::runtime_error (the reference aborts)

class Platform {
 public:
  explicit Platform(size_t platform_id) : id_(platform_id) {}
  size_t id() const { return id_; }

 private:
  size_t id_;
};

class Device {
 public:
  Device(const Platform& platform, size_t device_id);
  int id() const { return id_; }
  std::string Vendor() const { return "Advanced Micro Devices, Inc."; }
  std::string Name() const { return name_; }
  std::string Version() const { return version_; }
  std::string Type() const { return "GPU"; }  // learner.cc:105-114 keys the work-group kernels off this
  uint64_t MaxAllocSize() const { return max_alloc_; }

 private:
  int id_;
  std::string name_, version_;
  uint64_t max_alloc_;
};

class Context {
 public:
  explicit Context(const Device& device) : device_(device.id()) {}
  int device() const { return device_; }

 private:
  int device_;
};

// A HIP stream.  Copyable handle (the reference copies Queues freely); the stream lives as long as
// any copy does.
class Queue {
 public:
  Queue(const Context& context, const Device& device);
  void Finish() const;
  void* stream() const { return impl_ ? impl_->stream : nullptr; }
  int device() const { return device_; }
  Context GetContext() const;
  Device GetDevice() const;

 private:
  struct Impl {
    void* stream;
    int device;
    ~Impl();
  };
  std::shared_ptr<Impl> impl_;
  int device_;
};

// hipMalloc'd array of T with the Read / Write / CopyTo / GetSize vocabulary of clcuda::Buffer.
// Copyable handle (shared ownership), like the reference's.
template <class T>
class Buffer {
 public:
  Buffer(const Context& context, size_t count);
  template <class It>
  Buffer(const Context& context, const Queue& queue, It begin, It end) : Buffer(context, static_cast<size_t>(end - begin)) {
    std::vector<T> host(begin, end);
    Write(queue, host.size(), host.data());
  }
  size_t GetSize() const { return count_ * sizeof(T); }  // bytes, as in CLCudaAPI
  size_t Count() const { return count_; }
  T* data() const { return impl_ ? static_cast<T*>(impl_->ptr) : nullptr; }
  T* operator()() const { return data(); }

  void Read(const Queue& queue, size_t n, T* host, size_t offset = 0) const;
  void Read(const Queue& queue, size_t n, std::vector<T>& host, size_t offset = 0) const {
    if (host.size() < n) host.resize(n);
    Read(queue, n, host.data(), offset);
  }
  void Write(const Queue& queue, size_t n, const T* host, size_t offset = 0);
  void Write(const Queue& queue, size_t n, const std::vector<T>& host, size_t offset = 0) { Write(queue, n, host.data(), offset); }
  void CopyTo(const Queue& queue, size_t n, const Buffer<T>& destination) const;

 private:
  struct Impl {
    void* ptr;
    int device;
    ~Impl();
  };
  std::shared_ptr<Impl> impl_;
  size_t count_;
};

// raw helpers shared by the template instantiations (device.cc)
void* DeviceAlloc(int device, size_t bytes);
void DeviceFree(int device, void* p);
void CopyH2D(const Queue& q, void* dst, const void* src, size_t bytes);
void CopyD2H(const Queue& q, void* dst, const void* src, size_t bytes);
void CopyD2D(const Queue& q, void* dst, const void* src, size_t bytes);

template <class T>
Buffer<T>::Impl::~Impl() { DeviceFree(device, ptr); }

template <class T>
Buffer<T>::Buffer(const Context& context, size_t count) : impl_(new Impl{nullptr, context.device()}), count_(count) {
  impl_->ptr = DeviceAlloc(context.device(), (count ? count : 1) * sizeof(T));
}

template <class T>
void Buffer<T>::Read(const Queue& queue, size_t n, T* host, size_t offset) const {
  CopyD2H(queue, host, data() + offset, n * sizeof(T));
}

template <class T>
void Buffer<T>::Write(const Queue& queue, size_t n, const T* host, size_t offset) {
  CopyH2D(queue, data() + offset, host, n * sizeof(T));
}

template <class T>
void Buffer<T>::CopyTo(const Queue& queue, size_t n, const Buffer<T>& destination) const {
  CopyD2D(queue, destination.data(), data(), n * sizeof(T));
}

}  // namespace clcuda
}  // namespace mcmc

#endif  // MCMC_AMD_DEVICE_H_
