// Device-selection and once-per-device bookkeeping used by the C-ABI entry points.  Plain C++ (no HIP types), so the
// logic is unit-tested on the CPU (tests/test_device_logic.py compiles this header with g++).
#pragma once
#include <atomic>

namespace kp2d {

// Memory kinds as hipPointerGetAttributes reports them, reduced to what the guard needs.
enum PtrKind : int { PTR_UNKNOWN = 0, PTR_HOST = 1, PTR_DEVICE = 2, PTR_MANAGED = 3 };

// Which device a handle-less entry point should make current.
//   caller_dev   the device current on entry
//   kind, owner  what the pointer query said about the argument (owner = device that allocated it)
//   query_ok     false: the query failed (an unregistered host pointer, a pointer HIP does not know)
// Only DEVICE (or managed) memory names its device: pinned HOST memory also reports the device that was current when
// it was allocated (normally 0), which is not where the caller wants the kernels to run.
inline int pick_device(int caller_dev, bool query_ok, int kind, int owner, int ndev) {
  if (!query_ok) return caller_dev;
  if (kind != PTR_DEVICE && kind != PTR_MANAGED) return caller_dev;
  if (owner < 0 || owner >= ndev) return caller_dev;
  return owner;
}

// "Has this been done on device d yet?" for per-device function attributes (hipFuncSetAttribute is per device, a
// process-wide `static bool` was wrong with two devices in one process).  Thread-safe: two threads may both run
// `fn` the first time (the attribute call is idempotent), neither skips it.  Devices past 63 are never cached.
struct PerDeviceOnce {
  std::atomic<unsigned long long> done{0};
  // returns fn()'s error code (0 = ok) or 0 when the device was already served
  template <typename Fn>
  int ensure(int dev, Fn&& fn) {
    const bool cached = dev >= 0 && dev < 64;
    if (cached && ((done.load(std::memory_order_acquire) >> dev) & 1ull)) return 0;
    const int e = fn();
    if (e == 0 && cached) done.fetch_or(1ull << dev, std::memory_order_release);
    return e;
  }
};

}  // namespace kp2d
