// RAII "current device" guard for the C-ABI entry points: every entry point that launches kernels or touches HIP
// objects of a handle runs with the handle's (or the tensors') device current and restores the caller's device on
// the way out, so a multi-GPU host thread never finds its device changed behind its back.
#pragma once
#include <hip/hip_runtime.h>

namespace kp2d {

struct DeviceGuard {
  int prev = -1;
  bool changed = false;
  explicit DeviceGuard(int dev) { enter(dev); }
  // handle-less entry points: the device that owns `devptr` (skipped when only one device is visible, and while the
  // caller's stream is being captured into a graph: a capturing caller already has the tensors' device current, and
  // the pointer query is not a call to make inside a capture)
  explicit DeviceGuard(const void* devptr, hipStream_t stream) {
    static const int ndev = [] { int n = 0; return hipGetDeviceCount(&n) == hipSuccess ? n : 0; }();
    if (ndev < 2 || !devptr) return;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess) { (void)hipGetLastError(); return; }
    if (cs != hipStreamCaptureStatusNone) return;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, devptr) == hipSuccess) enter(at.device);
    else (void)hipGetLastError();   // not a device pointer: the launch itself will report it
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
