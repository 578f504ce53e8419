#!/usr/bin/env python3
"""Timeline of the streams of a bench.py run from a rocprofv3 --kernel-trace CSV: which kernels of which HIP stream (queue)
overlap in time.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -o t -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-precision-modes
    python3 tools/trace_lanes.py gpurun_out/trace/t_kernel_trace.csv [AT] > profiles/rN_trace_lanes.txt

AT (default 0.5): where in the run's sequence of forwards the window starts, as a fraction (bench.py times the
one-step-at-a-time loop first and the steps-in-flight loop second: 0.3 lands in the first, 0.7 in the second).

Prints, for a window of the timed steps: every kernel with its queue, start offset and duration, then the share of the
window in which 0 / 1 / 2+ queues had a kernel running.
"""
import csv
import re
import sys


def short(n):
    n = re.sub(r"\(.*", "", n).replace("kp2d::", "").replace("void ", "")
    return n[:46]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    seen = set()
    rows = [r for r in rows if "kp2d::" in r["Kernel_Name"] and not (r["Dispatch_Id"] in seen or seen.add(r["Dispatch_Id"]))]
    rows.sort(key=lambda r: r["s"])
    # a forward starts with conv1a, or — big grids, float frames — with conv1b's launch that computes conv1a itself (STEM)
    firsts = [i for i, r in enumerate(rows) if "conv1a" in r["Kernel_Name"] or "_ws_kernel" in r["Kernel_Name"]]
    if any("conv1a" in rows[i]["Kernel_Name"] for i in firsts):      # (unfused: conv1b's own launch is not a start)
        firsts = [i for i in firsts if "conv1a" in rows[i]["Kernel_Name"]]
    # a window of two consecutive forwards in the middle of the run (the timed steps, away from warm-up and the profiling forward)
    at = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    mid = int(len(firsts) * at)
    lo, hi = rows[firsts[mid]]["s"], rows[firsts[min(mid + 2, len(firsts) - 1)]]["s"]
    win = [r for r in rows if r["s"] >= lo and r["s"] < hi]
    queues = sorted({r["Queue_Id"] for r in win})
    print(f"# {sys.argv[1]}: {len(rows)} kp2d kernels, window of two forwards = {(hi - lo) / 1e6:.3f} ms, queues {queues}")
    print("# start_us   dur_us  queue  workgroups  kernel")
    for r in win:
        wg = (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))) * int(r["Grid_Size_Y"])
        col = queues.index(r["Queue_Id"])
        print(f"{(r['s'] - lo) / 1e3:9.1f} {(r['e'] - r['s']) / 1e3:8.1f}  {'    ' * col}q{r['Queue_Id']:<3s}{'    ' * (len(queues) - 1 - col)} {wg:6d}  {short(r['Kernel_Name'])}")
    ev = []
    for r in win:
        ev.append((r["s"], 1)); ev.append((min(r["e"], hi), -1))
    ev.sort()
    busy = {0: 0, 1: 0, 2: 0}
    n, t = 0, lo
    for ts, d in ev:
        busy[min(n, 2)] += ts - t
        t, n = ts, n + d
    busy[min(n, 2)] += hi - t
    tot = hi - lo
    print(f"# share of the window with 0 / 1 / 2+ kernels running: {busy[0] / tot:.3f} / {busy[1] / tot:.3f} / {busy[2] / tot:.3f}")


if __name__ == "__main__":
    main()
