// Per-device launch helpers: the CU count and the "max dynamic LDS" function attribute are properties of the
// CURRENT device (the one the caller's stream belongs to; the host wrappers make it current before a launch), so
// both are cached per device id -- a process that drives several GPUs must not reuse device 0's answers.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>

namespace fc {

constexpr int kMaxDevices = 64;

inline int current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
  return dev;
}

inline int device_cu_count() {
  static std::atomic<int> cus[kMaxDevices];
  const int dev = current_device();
  int c = cus[dev].load(std::memory_order_relaxed);
  if (c == 0) {
    hipDeviceProp_t prop;
    c = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 0;
    if (c <= 0) c = 256;
    cus[dev].store(c, std::memory_order_relaxed);
  }
  return c;
}

// One of these per kernel instantiation (a function-local static): remembers on which devices the attribute is set.
struct PerDeviceOnce {
  std::atomic<uint64_t> mask{0};
};

// hipFuncAttributeMaxDynamicSharedMemorySize applies to the current device only.  Thread-safe: two threads racing
// here both set the (idempotent) attribute.
inline hipError_t ensure_max_dynamic_lds(PerDeviceOnce& once, const void* func, int bytes) {
  const uint64_t bit = 1ull << current_device();
  if (once.mask.load(std::memory_order_acquire) & bit) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return e;
  once.mask.fetch_or(bit, std::memory_order_release);
  return hipSuccess;
}

}  // namespace fc
