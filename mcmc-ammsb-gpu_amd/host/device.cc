// HIP-backed implementation of the CLCudaAPI-shaped facade (include/mcmc/device.h).
#include "mcmc/device.h"

#include <hip/hip_runtime.h>

#include <stdexcept>
#include <string>

namespace mcmc {
namespace clcuda {

void Check(int e, const char* what) {
  if (e != hipSuccess)
    throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(static_cast<hipError_t>(e)));
}

Device::Device(const Platform&, size_t device_id) : id_(static_cast<int>(device_id)) {
  int count = 0;
  Check(hipGetDeviceCount(&count), "hipGetDeviceCount");
  if (id_ >= count) throw std::runtime_error("no such HIP device");
  hipDeviceProp_t prop;
  Check(hipGetDeviceProperties(&prop, id_), "hipGetDeviceProperties");
  name_ = prop.name;
  version_ = prop.gcnArchName;
  max_alloc_ = prop.totalGlobalMem;  // one allocation may span the whole HBM
}

Queue::Impl::~Impl() {
  if (stream) {
    (void)hipSetDevice(device);
    (void)hipStreamDestroy(static_cast<hipStream_t>(stream));
  }
}

Queue::Queue(const Context& context, const Device&) : impl_(new Impl{nullptr, context.device()}), device_(context.device()) {
  Check(hipSetDevice(device_), "hipSetDevice");
  hipStream_t s;
  Check(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreate");
  impl_->stream = s;
}

void Queue::Finish() const { Check(hipStreamSynchronize(static_cast<hipStream_t>(stream())), "hipStreamSynchronize"); }
Context Queue::GetContext() const { return Context(Device(Platform(0), device_)); }
Device Queue::GetDevice() const { return Device(Platform(0), device_); }

void* DeviceAlloc(int device, size_t bytes) {
  Check(hipSetDevice(device), "hipSetDevice");
  void* p = nullptr;
  Check(hipMalloc(&p, bytes), "hipMalloc");
  return p;
}

void DeviceFree(int device, void* p) {
  if (!p) return;
  (void)hipSetDevice(device);
  (void)hipFree(p);
}

// Read / Write are synchronous in CLCudaAPI; keep that.
void CopyH2D(const Queue& q, void* dst, const void* src, size_t bytes) {
  if (!bytes) return;
  Check(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, static_cast<hipStream_t>(q.stream())), "H2D");
  q.Finish();
}
void CopyD2H(const Queue& q, void* dst, const void* src, size_t bytes) {
  if (!bytes) return;
  Check(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, static_cast<hipStream_t>(q.stream())), "D2H");
  q.Finish();
}
void CopyD2D(const Queue& q, void* dst, const void* src, size_t bytes) {
  if (!bytes) return;
  Check(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(q.stream())), "D2D");
}

}  // namespace clcuda
}  // namespace mcmc
