// RAII "current device" guard for the C-ABI entry points: every entry point that launches kernels or touches HIP
// objects of a handle runs with the handle's (or the tensors') device current and restores the caller's device on
// the way out, so a multi-GPU host thread never finds its device changed behind its back.
#pragma once
#include <hip/hip_runtime.h>

#include "device_logic.h"

namespace kp2d {

// Opt one kernel function in to the full 160 KB of dynamic LDS on the CURRENT device (the attribute is per device).
inline int lds_opt_in(PerDeviceOnce& once, const void* fn) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = -1; }
  return once.ensure(dev, [&]() -> int {
    return (int)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
}

// Compute units of the CURRENT device, queried once per device (the persistent kernels size their grids by it at every
// launch; all visible devices of a node are the same part, but the answer is kept per device anyway).
inline int device_cu_count() {
  static int cached[16] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 256; }
  if (dev >= 0 && dev < 16 && cached[dev] > 0) return cached[dev];
  int n = 256;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); n = 256; }
  if (n <= 0) n = 256;
  if (dev >= 0 && dev < 16) cached[dev] = n;
  return n;
}

struct DeviceGuard {
  int prev = -1;
  bool changed = false;
  explicit DeviceGuard(int dev) { enter(dev); }
  // handle-less entry points: the device that owns `devptr`, which must be DEVICE memory to count (pinned host memory
  // reports the device it was allocated under, not the one the caller runs on: device_logic.h::pick_device).  Skipped
  // when only one device is visible, and while the caller's stream is being captured into a graph: a capturing caller
  // already has the tensors' device current, and the pointer query is not a call to make inside a capture.
  explicit DeviceGuard(const void* devptr, hipStream_t stream) {
    static const int ndev = [] { int n = 0; return hipGetDeviceCount(&n) == hipSuccess ? n : 0; }();
    if (ndev < 2 || !devptr) return;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess) { (void)hipGetLastError(); return; }
    if (cs != hipStreamCaptureStatusNone) return;
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); return; }
    hipPointerAttribute_t at;
    const bool ok = hipPointerGetAttributes(&at, devptr) == hipSuccess;
    if (!ok) (void)hipGetLastError();   // not a pointer HIP knows: the launch itself will report it
    const int kind = !ok ? PTR_UNKNOWN : at.type == hipMemoryTypeDevice ? PTR_DEVICE
                   : at.type == hipMemoryTypeManaged ? PTR_MANAGED : at.type == hipMemoryTypeHost ? PTR_HOST : PTR_UNKNOWN;
    const int dev = pick_device(cur, ok, kind, ok ? at.device : -1, ndev);
    if (dev != cur) enter(dev);
  }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
  ~DeviceGuard() { if (changed) (void)hipSetDevice(prev); }

 private:
  void enter(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) { prev = -1; return; }
    if (prev != dev) changed = hipSetDevice(dev) == hipSuccess;
  }
};

}  // namespace kp2d
