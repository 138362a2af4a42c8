// pstl_common.hpp -- shared by the HIP translation units of libpstl_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/pstl_hip.h"

namespace pstl {

inline int check_cfg(const pstl_cfg* c) {
  if (!c) return PSTL_ERR_ARG;
  if (c->bs <= 0 || c->rows_per_scene <= 0 || c->K < 1 || c->steps < 2) return PSTL_ERR_ARG;  // the reference needs K >= 1 too
  return PSTL_OK;
}

// Row buffers of 40 controls / (T,4) states are read and written 16 bytes at a time: they must be 16-byte aligned (every
// torch allocation and every row-wise slice of one is).
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline long n_rows(const pstl_cfg* c) { return (long)c->bs * c->rows_per_scene; }

inline int launch_status() { return hipGetLastError() == hipSuccess ? PSTL_OK : PSTL_ERR_LAUNCH; }

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Per-device facts, for a host that drives several GPUs from several threads (single-slot function statics raced
// there and could hand one device's CU count to another): one lock-free table indexed by the device id, shared by every
// translation unit (the entries are idempotent: two threads that fill one at the same time write the same value).
constexpr int kMaxDevices = 64;
inline int current_device() {
  int dev = 0;
  return hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < kMaxDevices ? dev : -1;
}
inline int device_cus(int dev) {   // compute units of `dev` (256 when it cannot be asked)
  static std::atomic<int> table[kMaxDevices];
  if (dev < 0 || dev >= kMaxDevices) return 256;
  int n = table[dev].load(std::memory_order_relaxed);
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    table[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}
inline int device_cus() { return device_cus(current_device()); }
// "this was done once for device `dev`" flags of one call site (a 64-bit mask: declare `static DeviceOnce once;` there)
struct DeviceOnce {
  std::atomic<unsigned long long> mask{0ull};
  bool done(int dev) const { return dev >= 0 && dev < kMaxDevices && ((mask.load(std::memory_order_acquire) >> dev) & 1ull); }
  void set(int dev) {
    if (dev >= 0 && dev < kMaxDevices) mask.fetch_or(1ull << dev, std::memory_order_release);
  }
};

}  // namespace pstl
