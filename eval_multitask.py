#!/usr/bin/env python3
"""eval_multitask.py equivalent: the reference's evaluation CLI (eval_multitask.py:35-344) with its
model-facing contract intact — same flag names, same construction / load_state_dict / eval sequence
(:142-198) — running on the MI355X engine.

The dataset-driven metric bodies (HPatches repeatability, COCO/Cityscapes mIoU, Pittsburgh recall@N, KITTI
VO: src/evaluation/*) need datasets, OpenCV, faiss and segmentation-models-pytorch, none of which exist
in this environment, and are OUT OF SCOPE of this build.  Each task flag therefore
runs the task's inference path on SYNTHETIC frames and reports shape/throughput/self-consistency figures,
so the CLI stays a usable smoke/throughput tool; point it at real data by swapping `synthetic_batches`.
"""
import argparse
import json
import os
import time
from datetime import datetime
from pathlib import Path

import numpy as np
import torch

from src.kp2dtiny.models.kp2dtiny import KP2DTinyV2, KP2DTinyV3, get_config
from nano_vs_slam_amd.selectors import select_keypoints


def parse_args():
    p = argparse.ArgumentParser(description="Evaluate multitask model (MI355X build, synthetic data)")
    p.add_argument("--device", type=str, default="cuda")
    p.add_argument("--model_path", type=str, default=None, help="checkpoint (.ckpt with a 'state_dict' entry)")
    p.add_argument("--dataset_config", type=str, default="datasets.json", help="accepted for CLI parity; unused")
    p.add_argument("--debug", action="store_true")
    p.add_argument("--num_workers", type=int, default=0)
    p.add_argument("--seed", type=int, default=42069)
    p.add_argument("--n_classes", type=int, default=28)
    p.add_argument("--model_type", type=str, default="KeypointNet")
    p.add_argument("--dataset_name", type=str, default="synthetic")
    p.add_argument("--config", type=str, default="S")
    p.add_argument("--batch_size", type=int, default=4)
    p.add_argument("--keypoints", action="store_true")
    p.add_argument("--visloc", action="store_true")
    p.add_argument("--segmentation", action="store_true")
    p.add_argument("--depth", action="store_true")
    p.add_argument("--vo", action="store_true")
    p.add_argument("--load_depth", action="store_true", help="build the model with depth=True (eval_multitask.py:150-153)")
    p.add_argument("--v3", action="store_true")
    p.add_argument("--result_dir", type=str, default="results")
    # flags of the reference CLI that select subsystems outside this build: accepted so existing command lines keep
    # working, ignored with a warning (quantisation: quantize.py / torch.ao; logging: wandb)
    p.add_argument("--quantized", action="store_true", help="accepted for CLI parity; PTQ is not built (warning)")
    p.add_argument("--backend", type=str, default="x86", help="accepted for CLI parity; quantisation backend, unused")
    p.add_argument("--wandb", action="store_true", help="accepted for CLI parity; no wandb logging (warning)")
    p.add_argument("--wandb_project", type=str, default="MT-Evaluation-Seg", help="accepted for CLI parity; unused")
    p.add_argument("--n_batches", type=int, default=8, help="synthetic batches per task")
    return p.parse_args()


def load_checkpoint(filename, optimizer_key=None):
    """The reference's checkpoint wire format (utils/utils.py:9-30), same contract: returns
    ``(state_dict, optimizer, info)``.  A ``.ckpt`` is a ``torch.save``d dict; with a ``"state_dict"`` entry the rest of
    the dict (epoch, config, ...) comes back as ``info``, without one the dict itself IS the state dict and ``info`` is
    None; ``optimizer_key`` names an entry to split off (missing: a warning, as the reference prints)."""
    filename = str(filename)
    assert filename.endswith(".ckpt"), "Error: filename is not a pth file"
    assert os.path.isfile(filename), "Error: checkpoint file not found"
    # weights_only=True: the no-code loader (tensors, numbers, strings, dicts / lists of them — what train_multitask.py
    # :553-562 saves: epoch, state_dict, optimizer state, config, results).  A checkpoint that needs arbitrary unpickling
    # is refused with the loader's own message; there is no fallback to the unsafe mode.
    try:
        checkpoint = torch.load(filename, map_location="cpu", weights_only=True)
    except Exception as e:  # pickle.UnpicklingError and friends
        raise RuntimeError(f"{filename}: refused by the safe checkpoint loader (weights_only=True): {e}") from e
    optimizer = None
    if optimizer_key is not None:
        if optimizer_key in checkpoint.keys():
            optimizer = checkpoint[optimizer_key]
            del checkpoint[optimizer_key]
        else:
            print("Warning: optimizer not found in checkpoint")
    if "state_dict" not in checkpoint:
        return checkpoint, optimizer, None
    state_dict = checkpoint["state_dict"]
    del checkpoint["state_dict"]
    return state_dict, optimizer, checkpoint


def synthetic_batches(n, batch, size, seed, device):
    g = torch.Generator(device=device).manual_seed(seed)
    for _ in range(n):
        yield torch.rand(batch, 3, size[0], size[1], device=device, generator=g) * 2 - 1


@torch.no_grad()
def main(args):
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    if args.quantized:
        print(f"Warning: --quantized (backend {args.backend!r}) ignored: post-training quantisation is outside this build; "
              "running the fp32-grade HIP path")
    if args.wandb:
        print(f"Warning: --wandb (project {args.wandb_project!r}) ignored: no wandb logging in this build")
    conf = get_config(args.config, v3=args.v3)
    model = (KP2DTinyV3 if args.v3 else KP2DTinyV2)(**conf, nClasses=args.n_classes, depth=args.load_depth)
    info = {}
    if args.model_path:
        sd, _optimizer, info = load_checkpoint(Path(args.model_path), optimizer_key="optimizer")
        info = info or {}
        try:
            model.load_state_dict(sd, strict=True)
        except Exception as e:                       # eval_multitask.py:161-167
            print("Error loading model state dict")
            print(e)
            print("Trying to load model state dict with strict=False")
            model.load_state_dict(sd, strict=False)
    else:
        from nano_vs_slam_amd.synthetic import spread_state_dict   # seeded stand-in weights (test infrastructure)
        sd = spread_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
        model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
        print("no --model_path: using seeded synthetic weights")
    model.eval()
    model.training = False
    model.to(args.device)
    model.device = args.device
    results = {"model": model.gather_info(), "checkpoint_info": {k: str(v) for k, v in info.items()}}
    # a fixed probe frame through forward(): lets a caller check that a checkpoint arrived intact (tests/test_ckpt_cli.py)
    probe = torch.from_numpy(np.random.default_rng(args.seed).random((1, 3, 64, 96), np.float32) * 2 - 1).to(args.device)
    po = model(probe)
    results["probe"] = {k: [float(v.double().sum()), float(v.double().abs().max())] for k, v in po.items()}
    if args.depth or args.vo:
        print("--depth / --vo need datasets + OpenCV pose estimation: out of scope of this build")
    for size in [(240, 320)]:
        key = f"{size[0]}x{size[1]}"
        res = results.setdefault(key, {})
        if args.keypoints:
            counts, t0 = [], time.perf_counter()
            for x in synthetic_batches(args.n_batches, args.batch_size, size, args.seed, args.device):
                out = model.post_processing(model(x), *size)
                for k in (300, 1000):
                    counts.append([len(p) for p, _, _ in select_keypoints(out, 0.7, k)])
            torch.cuda.synchronize()
            res["keypoints"] = {"frames": args.n_batches * args.batch_size, "mean_selected": float(np.mean(counts)),
                                "frames_per_s": args.n_batches * args.batch_size / (time.perf_counter() - t0)}
        if args.visloc:
            vl = torch.cat([model(x)["vlad"] for x in synthetic_batches(args.n_batches, args.batch_size, size, args.seed, args.device)])
            d = torch.cdist(vl, vl)
            res["visloc"] = {"db": int(vl.shape[0]), "dim": int(vl.shape[1]),
                             "self_recall@1": float((d.argmin(1) == torch.arange(len(vl), device=vl.device)).float().mean())}
        if args.segmentation:
            hist = torch.zeros(args.n_classes, dtype=torch.long, device=args.device)
            for x in synthetic_batches(args.n_batches, args.batch_size, size, args.seed, args.device):
                ids = model.post_processing(model(x), *size)["seg"]
                hist += torch.bincount(ids.reshape(-1), minlength=args.n_classes)
            res["segmentation"] = {"class_histogram": hist.tolist()}
    os.makedirs(args.result_dir, exist_ok=True)
    out_path = os.path.join(args.result_dir, datetime.now().strftime("%Y-%m-%d_%H-%M-%S") + ".json")
    with open(out_path, "w") as f:
        json.dump(results, f, indent=1, default=str)
    print(json.dumps({k: v for k, v in results.items() if k != "model"}, indent=1, default=str))
    print("saved", out_path)


if __name__ == "__main__":
    main(parse_args())
