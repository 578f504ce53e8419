#!/usr/bin/env python3
"""Headline benchmark: frames/sec of the KP2DTiny-S 240x320 multi-task inference path.

    python bench.py --gpus N --steps K --warmup W

N > 1 runs one rank per GPU either way it is started:
  * under a launcher (python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...): the
    process IS a rank (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment; WORLD_SIZE != --gpus is refused);
  * started plainly (no WORLD_SIZE in the environment): this process only launches — before anything touches the GPU it
    starts N fresh child processes of this same file with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 /
    MASTER_PORT set, relays rank 0's JSON line and exits with the worst child code (self_launch()).

One "step" = one pass of the hot path over one batch of synthetic frames already resident in HBM:
``model(x)`` (backbone + score/loc/descriptor/segmentation/NetVLAD heads) + ``post_processing`` +
threshold/top-k keypoint selection — all on the device, through the C ABI.  Weak scaling: every rank
owns ``--batch`` frames (frame-batch sharding); the only collective is the start-up
weight broadcast.  Rank 0 prints ONE JSON line.

Extra objects on that line:
  roofline      the dominant kernel (3x3-conv implicit GEMM): algorithmic FLOPs / HIP-event time per launch.  Peak: in
                the default f16x3 arithmetic one fp32-grade product costs three fp16 MFMAs (v_mfma_f32_16x16x32_f16),
                so the algorithmic peak is the dense fp16 matrix peak / 3 = 838.9 TFLOP/s; with --precision fp32 it is
                the fp32 matrix peak 157.3 TFLOP/s (v_mfma_f32_32x32x2_f32)
  collective    (N > 1) the backend and rank count torch.distributed reports, and the one collective of the job: the
                start-up weight broadcast (bytes, ms)
  cpu_baseline  the same workload on the host cores (oracle/torch_port.py, plain torch CPU ops), bounded sample
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz
PEAK_F16_MFMA_TFLOPS = 2516.6  # dense fp16 MFMA: 256 CU x 4 SIMD x 1024 FLOP/clk x 2.4 GHz (MI355X_MICROARCH.md "~2.5 PF")
PEAK_HBM_GBS = 8000.0
LAYER_BOUNDARY_MB = {(240, 320): 123.86, (120, 160): 30.96, (480, 640): 495.44}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 50 timed steps (0.13 s at 64 frames of 240x320) — with two steps in flight the first and the last step of the
    # timed region run alone (pipeline fill and drain between the two synchronisations), which at 20 steps read 1.2 % below
    # the rate sustained over 1000 (profiles/r5_box_variance.txt); 10 warm-up steps
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64, help="frames per GPU per step")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="frames per step over ALL ranks, split into contiguous shards (BASELINE cfg 3: 256 over 8 GPUs "
                         "= 32 per GPU); overrides --batch")
    ap.add_argument("--no-precision-modes", action="store_true",
                    help="skip timing the other arithmetic mode (precision_modes in the JSON line)")
    ap.add_argument("--height", type=int, default=240)
    ap.add_argument("--width", type=int, default=320)
    ap.add_argument("--config", default="S")
    ap.add_argument("--v3", action="store_true")
    ap.add_argument("--n-classes", type=int, default=28)
    ap.add_argument("--top-k", type=int, default=1000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--profile-steps", type=int, default=3)
    ap.add_argument("--precision", default=os.environ.get("KP2D_PRECISION", "f16x3"), choices=["f16x3", "fp32"])
    ap.add_argument("--in-flight", type=int, default=2,
                    help="batches in flight per GPU (pipeline.BatchStream: steps alternate over this many HIP streams, each "
                         "forward as one engine lane); 1 = one step after the other on one stream, the engine's two lanes inside")
    ap.add_argument("--dry-run", action="store_true",
                    help="rendezvous only: every rank joins the process group, one all_gather, rank 0 prints a line marked "
                         "dry_run with no throughput in it (checks the launch path on a box without a GPU)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only for "
                                                      "rehearsing the multi-rank path on a single-GPU box)")
    return ap.parse_args()


def seeded_state_dict(model):
    from nano_vs_slam_amd.synthetic import spread_state_dict
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    return spread_state_dict(shapes)


def kernel_profile(model, x, H, W, steps):
    """Per-launch HIP-event timing inside the engine (events recorded on the launch stream)."""
    eng = model._engine
    lib = eng.lib
    lib.kp2d_set_profiling(eng.handle, 1)
    agg = {}
    for _ in range(steps):
        model(x)
        n = lib.kp2d_profile_count(eng.handle)
        layer, kern = C.c_char_p(), C.c_char_p()
        ms, fl, by = C.c_float(), C.c_double(), C.c_double()
        for i in range(n):
            lib.kp2d_profile_get(eng.handle, i, C.byref(layer), C.byref(kern), C.byref(ms), C.byref(fl), C.byref(by))
            fam = kern.value.decode().split("<")[0]
            a = agg.setdefault(fam, dict(ms=0.0, flops=0.0, bytes=0.0, launches=0))
            a["ms"] += ms.value
            a["flops"] += fl.value
            a["bytes"] += by.value
            a["launches"] += 1
    lib.kp2d_set_profiling(eng.handle, 0)
    return agg


def host_cores():
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota (the GPU box
    exposes all host threads in /proc but grants a share of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(args, sd_np):
    """Host-core timing of the same step (forward + post + selection) on a bounded sample."""
    from oracle import kp2d_oracle as orc
    from oracle import torch_port as tp
    from oracle.weights import synthetic_frames
    cfg = orc.get_config(args.config, args.v3)
    p = tp.to_torch(sd_np)
    cores = host_cores()
    torch.set_num_threads(cores)
    best, best_b, log = 0.0, 1, []
    budget = args.cpu_seconds / 2
    for b in (1, 8):
        x = torch.from_numpy(synthetic_frames(b, args.height, args.width, seed=7))
        with torch.no_grad():
            def step():
                out = tp.forward(x, p, cfg)
                post = tp.post_processing(out, args.height, args.width, cfg)
                return tp.select(post, 0.7, args.top_k)
            step()
            t0 = time.perf_counter()
            n = 0
            while True:
                step()
                n += 1
                dt = time.perf_counter() - t0
                if dt > budget or n >= 50:
                    break
        fps = n * b / dt
        log.append(f"B={b}: {n} iters, {fps:.1f} frames/s")
        if fps > best:
            best, best_b = fps, b
    model_name = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model_name = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": round(best, 2), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"torch-CPU port of the path (oracle/torch_port.py), {args.config} {args.height}x{args.width}, "
                      f"forward+post+select, best of B in (1,8) = B{best_b}; " + "; ".join(log) + f"; cpu: {model_name}"}


def rocprof_conv_frac(flops_per_forward, peak_tflops):
    """The conv family's roofline fraction recomputed from the newest committed rocprofv3 --stats summary of this workload
    (profiles/r<round>_kernel_stats_single_lane.csv: `KP2D_LANES=1 ... bench.py --in-flight 1` under the profiler, one lane,
    nothing overlapped): sum of the conv3x3_f16x3* kernels' durations / forwards in that run (NetVLAD's finish pass runs once per forward).
    Printed beside the HIP-event figure so that the line and the CSV cannot drift apart unnoticed."""
    import csv
    import glob
    import re
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_kernel_stats_single_lane.csv")):
        m = re.match(r"r(\d+)_kernel_stats_single_lane\.csv$", os.path.basename(f))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), f)
    if best is None:
        return None
    conv_ns, fwd, launches = 0.0, 0, 0
    for row in csv.DictReader(open(best[1])):
        name = row["Name"]
        if "conv3x3_f16x3" in name:
            conv_ns += float(row["TotalDurationNs"])
            launches += int(row["Calls"])
        if "netvlad_finish" in name:      # once per forward and lane (conv1a is not a launch of its own on big grids: STEM)
            fwd += int(row["Calls"])
    if not fwd or not conv_ns:
        return None
    ms = conv_ns / fwd * 1e-6
    rel = os.path.relpath(best[1], ROOT)
    return {"file": rel, "git": file_commit(rel), "forwards": fwd, "conv_launches_per_forward": launches // fwd,
            "conv_ms_per_forward": round(ms, 4), "achieved": round(flops_per_forward / (ms * 1e-3) / 1e12, 2),
            "frac": round(flops_per_forward / (ms * 1e-3) / 1e12 / peak_tflops, 4)}


def newest_traffic_file():
    """profiles/r<round>_traffic.json of the highest round (None if there is none)."""
    import glob
    import re
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")):
        m = re.match(r"r(\d+)_traffic\.json$", os.path.basename(f))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), f)
    return best[1] if best else None


def file_commit(rel):
    """Short hash of the commit that last changed `rel` (the GPU box has no .git: 'unknown' there)."""
    import subprocess
    try:
        out = subprocess.run(["git", "-C", ROOT, "log", "-1", "--format=%h", "--", rel], capture_output=True, text=True, timeout=10)
        return out.stdout.strip() or "unknown"
    except (OSError, subprocess.SubprocessError):
        return "unknown"


def self_launch(args):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks ourselves.

    The parent never initialises the GPU (it only parses arguments and waits), the children are fresh interpreters
    (subprocess, no fork of a HIP context, no exec from a process that has touched the device).  Rank 0's stdout is
    relayed line by line to stdout, the other ranks' stdout goes to stderr; stderr passes through.  Exit
    code = the worst child code; a rank that dies takes the others down after a grace period instead of leaving
    them waiting in a collective."""
    import socket
    import subprocess
    import threading
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    base = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(args.gpus),
                LOCAL_WORLD_SIZE=str(args.gpus), KP2D_BENCH_SELF_LAUNCHED="1")
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL needs it)
    # the N ranks share this host's cores (a GPU box grants ~2 per GPU: 16 for eight ranks that each enqueue ~40 launches per
    # 2.7 ms step): one rank must not spin up a full-size OpenMP / intra-op pool per process
    per_rank = max(1, host_cores() // max(1, args.gpus))
    base.setdefault("OMP_NUM_THREADS", str(per_rank))
    base.setdefault("MKL_NUM_THREADS", str(per_rank))
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    procs = []
    for r in range(args.gpus):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1))

    def relay(p, r):
        for ln in p.stdout:
            if r == 0:
                sys.stdout.write(ln)
                sys.stdout.flush()
            elif not ln.startswith("{"):
                sys.stderr.write(f"[rank {r}] {ln}")

    threads = [threading.Thread(target=relay, args=(p, r), daemon=True) for r, p in enumerate(procs)]
    for t in threads:
        t.start()
    worst, deadline = 0, None
    alive = set(range(args.gpus))
    while alive:
        for r in sorted(alive):
            rc = procs[r].poll()
            if rc is not None:
                alive.discard(r)
                if rc != 0:
                    worst = worst or rc
                    if deadline is None:
                        deadline = time.monotonic() + 30.0      # the others are probably stuck in a collective
        if deadline is not None and time.monotonic() > deadline:
            for r in alive:
                procs[r].kill()                                  # exact PIDs we started, nothing else
            deadline = None
        time.sleep(0.05)
    for t in threads:
        t.join(timeout=5)
    return worst


def dry_run(args, rank, world):
    """--dry-run: the launch + rendezvous path only (no device, no kernels, no throughput)."""
    import torch.distributed as dist
    from nano_vs_slam_amd.sharding import shard_range
    gb = args.global_batch if args.global_batch > 0 else world * (args.batch if args.batch > 0 else 64)
    if gb < world:
        raise SystemExit("--global-batch must be at least the number of ranks")
    lo, hi = shard_range(gb, rank, world)               # the frame shard this rank would run
    mine = torch.tensor([rank, lo, hi, torch.get_num_threads()], dtype=torch.int64)
    every = [mine]
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
    rows = sorted([int(v) for v in e.tolist()] for e in every)
    seen = [r[0] for r in rows]
    if rank == 0:
        print(json.dumps({"metric": "dry-run (rendezvous only, nothing measured)", "dry_run": True, "value": None,
                          "n_gpus": world, "collective": {"backend": "gloo", "ranks_seen": len(seen), "ranks": seen},
                          "global_batch": gb, "frame_shards": [[r[1], r[2]] for r in rows],
                          "host_threads_per_rank": [r[3] for r in rows], "host_cores": host_cores(),
                          "launcher": "self (bench.py started its own ranks)" if os.environ.get("KP2D_BENCH_SELF_LAUNCHED")
                          else "external (torchrun or equivalent)"}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        # the line's n_gpus must be what was asked for AND what ran: refuse anything else
        raise SystemExit(f"WORLD_SIZE={world} but --gpus {args.gpus}: launch with python -m torch.distributed.run "
                         f"--nproc-per-node {args.gpus} ... bench.py --gpus {args.gpus}")
    if world > 1:      # ranks of one node share its cores (self_launch sets OMP_NUM_THREADS too; a launcher may not have)
        torch.set_num_threads(max(1, min(torch.get_num_threads(), host_cores() // world)))
    if args.dry_run:
        return dry_run(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU path to measure)")
    dev_index = local_rank % torch.cuda.device_count()     # one rank per GPU; wraps only in single-GPU rehearsals
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    from nano_vs_slam_amd.kp2dtiny.models.kp2dtiny import tiny_factory
    from nano_vs_slam_amd.selectors import select_and_gather
    from nano_vs_slam_amd.sharding import broadcast_model_weights

    model = tiny_factory(args.config, args.n_classes, v3=args.v3)
    sd_np = None
    if rank == 0:
        sd_np = seeded_state_dict(model)
        model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd_np.items()})
    model = model.to(dev).eval()
    model.training = False
    model.set_precision(args.precision)
    collective = None
    if world > 1:
        torch.cuda.synchronize(dev)
        dist.barrier()
        t0 = time.perf_counter()
        nbytes = broadcast_model_weights(model, dev, src=0)   # the one RCCL collective of the job
        torch.cuda.synchronize(dev)
        collective = {"backend": dist.get_backend(), "ranks_seen": dist.get_world_size(),
                      "broadcast_bytes": int(nbytes), "broadcast_ms": round((time.perf_counter() - t0) * 1e3, 3),
                      "steady_state_collectives": 0}

    from nano_vs_slam_amd.sharding import shard_range
    if args.global_batch > 0:
        if args.global_batch < world:
            raise SystemExit("--global-batch must be at least the number of ranks")
        lo, hi = shard_range(args.global_batch, rank, world)      # contiguous frame shard of this rank
        B = hi - lo
        global_batch = args.global_batch
    else:
        B = args.batch
        global_batch = world * B
    H, W = args.height, args.width
    g = torch.Generator(device=dev).manual_seed(7 + rank)
    x = torch.rand(B, 3, H, W, device=dev, generator=g) * 2.0 - 1.0   # synthetic frames generated in HBM

    def step():
        out = model(x)
        out = model.post_processing(out, H, W)
        idx, val, cnt, pts, desc = select_and_gather(out["score"], out["coord"], out["feat"], args.top_k, 0.7)
        return out, pts, desc, cnt

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    rank_fps = {}

    from nano_vs_slam_amd.pipeline import BatchStream

    def timed(precision, in_flight=None, key=None):
        """W untimed + K timed steps in one arithmetic mode; seconds for the K steps, MAX over ranks.
        in_flight > 1: the same K steps (same kernels, same batch, every step complete before the closing fence), enqueued
        alternately on that many streams so that consecutive steps overlap (pipeline.BatchStream)."""
        in_flight = args.in_flight if in_flight is None else in_flight
        key = key or precision
        model.set_precision(precision)
        bs = BatchStream(model, slots=in_flight, top_k=args.top_k, nn_thresh=0.7, device=dev) if in_flight > 1 else None
        run = (lambda: bs.submit(x)) if bs is not None else step
        with torch.no_grad():
            for _ in range(args.warmup):
                run()
            fence()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                run()
            fence()
            dt_ = time.perf_counter() - t0
        if bs is not None:
            bs.close()
        t = torch.tensor([dt_], dtype=torch.float64, device=dev)
        if world > 1:
            mine = torch.tensor([B * args.steps / dt_], dtype=torch.float64, device=dev)
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)
            rank_fps[key] = [float(v.item()) for v in every]
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # both arithmetic modes are driver-visible: the headline one (--precision) is `value`; the other is timed the same
    # way (same K, W, barriers) and reported beside it in `precision_modes`
    other = "fp32" if args.precision == "f16x3" else "f16x3"
    modes = {}
    if not args.no_precision_modes:
        modes[other] = timed(other)
    # one step after the other on one stream (the engine's own two lanes inside each forward), reported beside the headline
    dt_serial = timed(args.precision, 1, key="serial") if args.in_flight > 1 else None
    dt = timed(args.precision)
    modes[args.precision] = dt

    if rank == 0:
        frames = global_batch * args.steps
        fps = frames / dt
        with torch.no_grad():
            agg = kernel_profile(model, x, H, W, args.profile_steps)
        dom = max(agg, key=lambda k: agg[k]["ms"])
        a = agg[dom]
        achieved = a["flops"] / (a["ms"] * 1e-3) / 1e12
        total_ms = sum(v["ms"] for v in agg.values())
        split = "f16x3" in dom
        # f16x3: one fp32-grade product = three fp16 MFMAs, so the algorithmic peak is the fp16 peak / 3 — for every
        # split kernel, the attention kernel included (its padded P.V tiles are the kernel's own inefficiency, not a
        # property of the peak).
        mfma_per_flop = 1.0 / 3.0
        peak = PEAK_F16_MFMA_TFLOPS * mfma_per_flop if split else PEAK_F32_MFMA_TFLOPS
        note = ("dense fp16 MFMA 2516.6 TFLOP/s / 3 MFMAs per fp32-grade product (xh*wh + xh*wl + xl*wh)"
                if split else "fp32 MFMA v_mfma_f32_32x32x2_f32")
        roof = {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 2), "peak": round(peak, 1),
                "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": None,
                "peak_note": note,
                "launches_per_step": a["launches"] // args.profile_steps,
                "avg_launch_ms": round(a["ms"] / a["launches"], 4),
                "share_of_kernel_time": round(a["ms"] / total_ms, 4),
                "algorithmic_gb_per_s": round(a["bytes"] / (a["ms"] * 1e-3) / 1e9, 1),
                "kernel_ms_per_step": {k: round(v["ms"] / args.profile_steps, 3) for k, v in agg.items()}}
        # How the pieces fit: the timed steps run the batch as `lanes` sub-batches on that many HIP streams (the tail of
        # one lane's launch overlaps the next launch of the other); the per-kernel figures above come from a separate
        # profiling forward that runs ONE lane with HIP events around every launch — the launch shape rocprofv3 sees
        # under KP2D_LANES=1.  So kernel_ms_sum (one lane, nothing overlapped, forward only) may exceed ms_per_step
        # (lanes overlapped, plus post_processing / top-k / gather, which are not in the sum).
        lanes = max(1, min(8, int(os.environ.get("KP2D_LANES", "2"))))
        roof["lanes"] = 1 if args.in_flight > 1 else min(lanes, B)
        roof["kernel_ms_sum"] = round(total_ms / args.profile_steps, 3)
        if args.in_flight > 1:
            roof["timing_note"] = (f"ms_per_step = time of the K steps / K with {args.in_flight} steps in flight (alternating HIP streams, one engine "
                                   "lane each; forward + post_processing + selection); kernel_ms_per_step / kernel_ms_sum / achieved: "
                                   "single-lane HIP events of the forward's launches, nothing overlapped")
        else:
            roof["timing_note"] = (f"ms_per_step: {roof['lanes']} stream lane(s) overlapped, forward + post_processing + selection; "
                                   "kernel_ms_per_step / kernel_ms_sum / achieved: single-lane HIP events of the forward's launches")
        if split:
            # a bare fp16 MFMA loop sustains 1571 TFLOP/s on this chip (clock drops to ~1.5 GHz under matrix load:
            # tools/probes/mfma_f16_probe.hip, profiles/r1_probe_f16.log) -> 523.7 TFLOP/s of fp32-grade products
            roof["frac_of_sustained_mfma"] = round(achieved / (1571.0 * mfma_per_flop), 4)
        # HBM-side traffic of the dominant kernel: NOT measured in this run (bench.py cannot run rocprofv3 around
        # itself).  It is read back from the newest committed PMC pass on this workload (tools/pmc_collect.sh ->
        # tools/pmc_traffic.py -> profiles/r<round>_traffic.json); the file and the commit that last touched it are named.
        try:
            tfile = newest_traffic_file()
            tr = json.load(open(tfile))
            if B == 64 and (H, W) == (240, 320) and dom in tr["kernels"]:
                roof["traffic"] = round(tr["kernels"][dom]["bytes_per_launch"])
                rel = os.path.relpath(tfile, ROOT)
                roof["traffic_unit"] = (f"bytes per launch; not measured in this run: PMC FETCH_SIZE x2 + WRITE_SIZE from {rel} "
                                        f"(git {tr.get('source_commit') or file_commit(rel)}), collected by tools/pmc_collect.sh on this workload")
                roof["algorithmic_bytes_per_launch"] = round(a["bytes"] / a["launches"])
            if B == 64 and (H, W) == (240, 320) and split and args.config == "S" and not args.v3:
                rp = rocprof_conv_frac(a["flops"] / args.profile_steps, peak)
                if rp:
                    roof["rocprof"] = rp      # (not measured in this run: the committed profiler summary of the same command)
        except (OSError, KeyError, ValueError, TypeError):
            pass
        lb = LAYER_BOUNDARY_MB.get((H, W))
        if lb and args.config == "S" and not args.v3:
            roof["hbm_layer_boundary_frac"] = round(fps / world * lb * 1e6 / (PEAK_HBM_GBS * 1e9), 4)
        line = {
            "metric": "frames/sec KP2DTiny-S 240x320 multitask infer",
            "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16x3 (split-fp16 operands, fp32 accumulate, fp32-grade)" if args.precision == "f16x3" else "f32",
            "data": "synthetic",
            "config": {"workload": f"KP2DTiny-{args.config}{'-V3' if args.v3 else ''} {H}x{W}, batch {B}/GPU"
                                   f"{' (rank 0 shard of ' + str(global_batch) + ')' if args.global_batch > 0 else ''}, "
                                   f"all heads (score/loc/desc/seg/NetVLAD) + post_processing + top-{args.top_k} selection",
                       "global_batch": global_batch, "frame_shards": world, "n_classes": args.n_classes,
                       "steps_in_flight": args.in_flight},
            "roofline": roof,
            "precision_modes": {k: {"value": round(frames / v, 1), "unit": "frames/s",
                                    "ms_per_step": round(v / args.steps * 1e3, 3),
                                    "arithmetic": ("split-fp16 operands on v_mfma_f32_16x16x32_f16, fp32 accumulate"
                                                   if k == "f16x3" else "exact fp32 on v_mfma_f32_32x32x2_f32")}
                                for k, v in sorted(modes.items())},
        }
        if dt_serial is not None:
            line["one_step_at_a_time"] = {"value": round(frames / dt_serial, 1), "unit": "frames/s",
                                          "ms_per_step": round(dt_serial / args.steps * 1e3, 3),
                                          "note": "same K steps on ONE stream, each complete before the next starts (two engine lanes inside a forward)"}
        if collective is not None:
            line["collective"] = collective
            line["launcher"] = "self (bench.py started its own ranks)" if os.environ.get("KP2D_BENCH_SELF_LAUNCHED") else "external (torchrun or equivalent)"
            fr = rank_fps.get(args.precision, [])
            if fr:
                line["per_rank_frames_per_s"] = {"min": round(min(fr), 1), "max": round(max(fr), 1)}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args, sd_np)
            line["speedup_vs_cpu"] = round(fps / line["cpu_baseline"]["value"], 1)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
