"""CPU unit test of nano-vs-slam_amd/csrc/device_logic.h: which device a handle-less C-ABI entry point makes current,
and the once-per-device bookkeeping behind hipFuncSetAttribute.  The header is plain C++ and is compiled here with g++."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r"""
#include <cstdio>
#include <thread>
#include <vector>
#include "device_logic.h"
using namespace kp2d;
static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAIL line %d: %s\n", __LINE__, #c); ++fails; } } while (0)
int main() {
  // pick_device: only device / managed memory names its device
  CHECK(pick_device(1, true, PTR_DEVICE, 0, 2) == 0);
  CHECK(pick_device(0, true, PTR_DEVICE, 1, 2) == 1);
  CHECK(pick_device(1, true, PTR_MANAGED, 0, 2) == 0);
  CHECK(pick_device(1, true, PTR_HOST, 0, 2) == 1);       // pinned host buffer allocated under device 0: stay on 1
  CHECK(pick_device(1, true, PTR_UNKNOWN, 0, 2) == 1);
  CHECK(pick_device(1, false, PTR_DEVICE, 0, 2) == 1);    // query failed (unregistered host pointer)
  CHECK(pick_device(1, true, PTR_DEVICE, 7, 2) == 1);     // nonsense owner
  CHECK(pick_device(1, true, PTR_DEVICE, -1, 2) == 1);
  // PerDeviceOnce: once per device, not once per process
  PerDeviceOnce once;
  int calls[4] = {0, 0, 0, 0};
  for (int rep = 0; rep < 3; ++rep)
    for (int d = 0; d < 4; ++d) CHECK(once.ensure(d, [&] { ++calls[d]; return 0; }) == 0);
  for (int d = 0; d < 4; ++d) CHECK(calls[d] == 1);
  // a failing call is reported and retried
  PerDeviceOnce bad;
  int n = 0;
  CHECK(bad.ensure(2, [&] { ++n; return 719; }) == 719);
  CHECK(bad.ensure(2, [&] { ++n; return 0; }) == 0);
  CHECK(bad.ensure(2, [&] { ++n; return 0; }) == 0);
  CHECK(n == 2);
  // devices outside the cached range always run fn
  PerDeviceOnce far;
  n = 0;
  far.ensure(64, [&] { ++n; return 0; });
  far.ensure(64, [&] { ++n; return 0; });
  far.ensure(-1, [&] { ++n; return 0; });
  CHECK(n == 3);
  // racing threads: every device is served at least once, none is skipped
  PerDeviceOnce race;
  std::atomic<int> served[8];
  for (auto& s : served) s = 0;
  std::vector<std::thread> th;
  for (int t = 0; t < 8; ++t)
    th.emplace_back([&, t] { for (int i = 0; i < 1000; ++i) race.ensure((t + i) & 7, [&] { ++served[(t + i) & 7]; return 0; }); });
  for (auto& t : th) t.join();
  for (auto& s : served) CHECK(s.load() >= 1 && s.load() <= 8);
  std::printf(fails ? "FAILED\n" : "OK\n");
  return fails;
}
"""


def test_device_logic_header(tmp_path):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    src = tmp_path / "t.cpp"
    src.write_text(SRC)
    exe = str(tmp_path / "t")
    subprocess.run([gxx, "-std=c++17", "-O1", "-pthread", "-I" + os.path.join(ROOT, "nano-vs-slam_amd", "csrc"), str(src), "-o", exe],
                   check=True, timeout=120)
    res = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert res.returncode == 0 and "OK" in res.stdout, res.stdout + res.stderr
