"""Parity of the HIP path (through the C ABI) with the reference's golden outputs and with the CPU oracle.

Tolerances (BASELINE.json north_star): float tensors within 1e-3 abs of the reference CPU path;
keypoint index sets identical.  Two faithful fp32 implementations differ by ~2e-5 (fixtures' meta
``reference_vs_fp64``), so the float asserts below use 2e-4 — 5x tighter than the contract — and index
sets are compared exactly except for cells closer than 2e-5 to a decision boundary.
"""
import os
import numpy as np
import pytest
import torch

from conftest import assert_topk_equivalent, golden_inputs, golden_names, load_golden, note_boundary_exempt, product_model
from oracle import kp2d_oracle as orc
from oracle.weights import synthetic_frames

pytestmark = pytest.mark.gpu
TOL = 2e-4
DEV = "cuda:0"


def _run(model, x, H, W):
    with torch.no_grad():
        out = model(torch.from_numpy(x).to(DEV))
        fwd = {k: v.cpu().numpy() for k, v in out.items()}
        post = model.post_processing(out, H, W)
        post_np = {k: v.cpu().numpy() for k, v in post.items()}
    return fwd, post, post_np


def _same_set(a, b, scores, boundary, tol=2e-5, label="set"):
    diff = np.setxor1d(a, b)
    assert np.all(np.abs(scores[diff] - boundary) <= tol), (diff, scores[diff])
    note_boundary_exempt(label, len(diff), len(b))


@pytest.mark.parametrize("precision", ["f16x3", "fp32"])
@pytest.mark.parametrize("name", golden_names())
def test_against_reference_golden(name, precision):
    """Both arithmetic modes of the conv kernels (exact fp32 MFMA, split-fp16 3xMFMA) meet the same bar."""
    from nano_vs_slam_amd.selectors import select_keypoints, select_topk
    meta, z = load_golden(name)
    cfg, sd, x = golden_inputs(meta)
    model, _ = product_model(meta["config"], meta["v3"], meta["n_classes"], recipe=meta.get("weights", "spread"))
    model.set_precision(precision)
    H, W, st = meta["H"], meta["W"], meta["dense_stride"]
    fwd, post, post_np = _run(model, x, H, W)
    assert set(fwd) == {"score", "coord", "feat", "vlad", "seg"} | ({"depth"} if "fwd_depth" in z else set())
    if "fwd_depth" in z:
        assert np.max(np.abs(fwd["depth"] - z["fwd_depth"])) < TOL
    assert np.max(np.abs(fwd["score"] - z["fwd_score"])) < TOL
    assert np.max(np.abs(fwd["coord"] - z["fwd_shift"])) < TOL
    assert np.max(np.abs(fwd["vlad"] - z["fwd_vlad"])) < 1e-5
    assert np.max(np.abs(fwd["feat"][:, :, ::st, ::st] - z["fwd_feat"])) < TOL
    assert np.max(np.abs(fwd["seg"][:, :, ::st, ::st] - z["fwd_seg"])) < TOL
    assert np.max(np.abs(post_np["score"] - z["post_score"])) < TOL
    assert np.max(np.abs(post_np["coord"] - z["post_coord"])) < 5e-4
    assert np.max(np.abs(post_np["feat"] - z["post_feat"])) < TOL
    assert post_np["seg"].dtype == np.int64 and post_np["seg"].shape == (meta["B"], 1, 2 * (H // model.cell), 2 * (W // model.cell))
    clear = z["seg_margin_f16"].astype(np.float32) > 1e-3
    assert np.array_equal(post_np["seg"][:, 0][clear], z["post_seg_u8"][:, 0][clear].astype(np.int64))
    assert (post_np["seg"][:, 0] != z["post_seg_u8"][:, 0]).mean() < 1e-3
    # keypoint selection: same sets as the reference-side selectors
    ref_scores = z["post_score"].reshape(meta["B"], -1)
    for k in (300, 1000, 4000):
        sel = select_keypoints(post, 0.7, k)
        for b in range(meta["B"]):
            pts, desc, idx = sel[b]
            ref = z[f"k1_top{k}_idx_{b}"]
            got = np.sort(idx.cpu().numpy())
            kth = ref_scores[b][ref].min() if len(ref) else 0.7
            bound = 0.7 if len(z[f"keep_idx_{b}"]) <= k else kth
            _same_set(got, ref, ref_scores[b], bound, label=f"{name}[{precision}] K1 top-{k} frame {b}")
            assert pts.shape == (len(got), 2) and desc.shape == (len(got), model.nfeatures)
            i = idx.long()
            assert torch.equal(pts[:, 0], post["coord"][b, 0].reshape(-1)[i])
            assert torch.equal(desc[:, 5], post["feat"][b, 5].reshape(-1)[i])
    kk = z["k3_idx"].shape[1]
    idx, val, cnt = select_topk(post["score"], kk)
    for b in range(meta["B"]):
        assert_topk_equivalent(idx[b].cpu().numpy(), ref_scores[b], z["k3_idx"][b], label=f"{name}[{precision}] K3 frame {b}")


@pytest.mark.parametrize("config,v3,ncls,B,H,W", [
    ("S", False, 28, 3, 72, 104),     # ragged: tiles hang over every edge
    ("S", False, 5, 1, 16, 16),       # smallest legal frame, few classes
    ("S", False, 28, 5, 16, 24),      # 384-pixel frames: every 256-pixel conv1a block straddles a frame boundary
    ("N", False, 28, 2, 40, 56),      # K=32 C=48, KC=8 path, padded channel groups
    ("S", True, 19, 2, 48, 80),       # V3 fused heads + Softmax2d
    ("N", True, 28, 1, 64, 64),
    ("S_A", False, 28, 2, 48, 64),    # attention seg head (V2)
    ("S_A", True, 19, 1, 72, 104),    # attention seg head (V3), ragged
    ("N_A", True, 28, 1, 32, 48),
    ("F", False, 28, 2, 48, 80),      # TINY_F: three pools (cell 8), 64-d descriptors, 64 x 128 NetVLAD
    ("D_A", True, 19, 1, 32, 48),     # LARGE_D: 64-wide conv1a, 256-wide attention (head dim 64), 128-d descriptors
    ("S", False, 28, 2, 80, 144),     # half-resolution maps of 40 x 72: 8 x 32 conv tiles (40 % 16 = 8) with a ragged column block
    ("S", True, 19, 1, 208, 272),     # 104 x 136 maps: 8 x 32 tiles, 13 tile rows, ragged columns, V3 heads
])
def test_against_oracle_other_shapes(config, v3, ncls, B, H, W):
    model, sd = product_model(config, v3, ncls)
    x = synthetic_frames(B, H, W, seed=21)
    fwd, post, post_np = _run(model, x, H, W)
    cfg = orc.get_config(config, v3)
    ref = orc.forward(x, sd, cfg)
    refp = orc.post_processing(ref, H, W, cfg)
    for k in ("score", "coord", "feat", "vlad", "seg"):
        assert fwd[k].shape == ref[k].shape
        assert np.max(np.abs(fwd[k] - ref[k])) < TOL, k
    for k in ("score", "feat"):
        assert np.max(np.abs(post_np[k] - refp[k])) < TOL, k
    assert np.max(np.abs(post_np["coord"] - refp["coord"])) < 5e-4
    assert (post_np["seg"] != refp["seg"]).mean() < 2e-3


def test_remove_netvlad_and_sample_segmentation():
    """to_export configs return the encoder map as "vlad" (vpr.py:84); sample_segmentation=True samples the class map
    at the keypoint coordinates with nearest-neighbour grid_sample (kp2dtiny.py:634-639)."""
    from nano_vs_slam_amd.kp2dtiny.models.kp2dtiny import tiny_factory
    from oracle.weights import spread_state_dict
    model = tiny_factory("S", 28, to_export=True)
    sd = spread_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    model = model.to(DEV).eval()
    model.training = False
    model.sample_segmentation = True
    x = synthetic_frames(2, 48, 64, seed=6)
    cfg = orc.get_config("S")
    cfg["remove_netvlad"] = True
    with torch.no_grad():
        out = model(torch.from_numpy(x).to(DEV))
        ref = orc.forward(x, sd, cfg)
        assert out["vlad"].shape == (2, 64, 12, 16)
        assert np.max(np.abs(out["vlad"].cpu().numpy() - ref["vlad"])) < TOL
        post = model.post_processing(out, 48, 64)
    refp = orc.post_processing(ref, 48, 64, cfg, sample_segmentation=True)
    assert post["seg"].shape == (2, 1, 12, 16) and post["seg"].dtype == torch.int64
    assert (post["seg"].cpu().numpy() != refp["seg"]).mean() < 0.02      # cells whose coord rounds on a pixel boundary


def test_training_mode_semantics():
    """model.training True: V3 returns logits (no Softmax2d) and post_processing skips sampling (kp2dtiny.py:615,942)."""
    model, sd = product_model("S", True, 19)
    x = synthetic_frames(1, 32, 48, seed=4)
    cfg = orc.get_config("S", True)
    model.training = True
    with torch.no_grad():
        out = model(torch.from_numpy(x).to(DEV))
        ref = orc.forward(x, sd, cfg, eval_mode=False)
        assert np.max(np.abs(out["seg"].cpu().numpy() - ref["seg"])) < TOL
        post = model.post_processing(out, 32, 48)
    assert post["feat"].shape == (1, 32, 16, 24) and post["seg"].dtype == torch.float32
    refp = orc.post_processing(ref, 32, 48, cfg, training=True)
    assert np.max(np.abs(post["coord"].cpu().numpy() - refp["coord"])) < 5e-4


def test_full_size_properties():
    """BASELINE configs[1]: KP2DTiny-S 240x320, batch 64 — size-independent properties."""
    from nano_vs_slam_amd.selectors import select_topk
    model, _ = product_model("S", False, 28)
    B, H, W = 64, 240, 320
    x = torch.from_numpy(synthetic_frames(B, H, W, seed=7)).to(DEV)
    with torch.no_grad():
        a = model(x)
        a = {k: v.clone() for k, v in a.items()}
        # frames are independent: permuting the batch permutes the outputs, bit for bit
        perm = torch.randperm(B, generator=torch.Generator().manual_seed(0)).to(DEV)
        b = model(x[perm].contiguous())
        for k in a:
            assert torch.equal(a[k][perm], b[k]), k
        # sub-batch size is an implementation detail: results are bitwise independent of it
        eng = model._engine
        eng.lib.kp2d_set_chunk_frames(eng.handle, 5)
        eng._ws = None
        c = model(x)
        eng.lib.kp2d_set_chunk_frames(eng.handle, 0)
        eng._ws = None
        for k in a:
            assert torch.equal(a[k], c[k]), k
        post = model.post_processing({k: v.clone() for k, v in a.items()}, H, W)
    assert torch.isfinite(a["feat"]).all() and torch.isfinite(a["seg"]).all()
    assert (a["score"] > 0).all() and (a["score"] < 1).all() and (a["coord"].abs() <= 1).all()
    assert torch.allclose(a["vlad"].norm(dim=1), torch.ones(B, device=DEV), atol=1e-5)
    s = post["score"]
    assert (s[:, :, 0] == 0).all() and (s[:, :, -1] == 0).all() and (s[..., 0] == 0).all() and (s[..., -1] == 0).all()
    assert torch.allclose(post["feat"].norm(dim=1), torch.ones_like(post["feat"][:, 0]), atol=1e-5)
    assert post["coord"][:, 0].min() >= 0 and post["coord"][:, 0].max() <= W - 1
    assert post["coord"][:, 1].min() >= 0 and post["coord"][:, 1].max() <= H - 1
    assert post["seg"].dtype == torch.int64 and post["seg"].min() >= 0 and post["seg"].max() < 28
    assert torch.equal(post["seg"][:, 0], a["seg"].argmax(1))
    idx, val, cnt = select_topk(s, 1000, 0.7)
    flat = s.reshape(B, -1)
    n_above = (flat > 0.7).sum(1)
    assert torch.equal(cnt.long(), torch.clamp(n_above, max=1000))
    tv, ti = flat.topk(1000, dim=1)
    for b in range(0, B, 9):
        n = int(cnt[b])
        assert torch.equal(val[b, :n], tv[b, :n])
        assert (val[b, :n] > 0.7).all() and (idx[b, n:] == -1).all()
        assert (val[b, 1:n] <= val[b, : n - 1]).all()


def test_warp_specialised_conv1b_equals_the_general_kernel_on_ragged_tiles():
    """backbone.conv1b runs as the warp-specialised persistent kernel once a launch has >= 1024 tiles (conv3x3_f16.hip);
    sub-batches of one frame take the general kernel.  272 columns = 8.5 tiles of 32: the last tile column is ragged."""
    model, _ = product_model("S", False, 28)
    B, H, W = 20, 208, 272      # two stream lanes of 10 frames = 1170 tiles each
    x = torch.from_numpy(synthetic_frames(B, H, W, seed=5)).to(DEV)
    eng_ready = model(x[:1])          # creates the engine
    del eng_ready
    eng = model._engine
    with torch.no_grad():
        eng.lib.kp2d_set_chunk_frames(eng.handle, 0)
        eng._ws = None
        a = {k: v.clone() for k, v in model(x).items()}
        eng.lib.kp2d_set_chunk_frames(eng.handle, 1)
        eng._ws = None
        c = model(x)
        eng.lib.kp2d_set_chunk_frames(eng.handle, 0)
        eng._ws = None
    for k in a:
        assert torch.equal(a[k], c[k]), k


def test_cfg4_full_size_properties():
    """BASELINE configs[3] per-GPU share: V3 S_A (efficient self-attention on) at 480x640 x 32 frames, 19 classes.
    This size runs in sub-batches under the 4 GiB workspace cap; results must not depend on that: batch permutation
    and explicit sub-batch sizes give bit-identical outputs, and frame 0 equals the reference fixture v3_SA_480x640."""
    meta, z = load_golden("v3_SA_480x640")
    cfg, sd, x0 = golden_inputs(meta)
    model, _ = product_model("S_A", True, 19)
    B, H, W = 32, 480, 640
    xs = synthetic_frames(B, H, W, seed=11)
    xs[0] = x0[0]                                  # frame 0 = the fixture's frame
    x = torch.from_numpy(xs).to(DEV)
    with torch.no_grad():
        a = {k: v.clone() for k, v in model(x).items()}
        perm = torch.randperm(B, generator=torch.Generator().manual_seed(1)).to(DEV)
        b = model(x[perm].contiguous())
        for k in a:
            assert torch.equal(a[k][perm], b[k]), k
        del b
        eng = model._engine
        for frames in (3, 16):
            eng.lib.kp2d_set_chunk_frames(eng.handle, frames)
            eng._ws = None
            c = model(x)
            for k in a:
                assert torch.equal(a[k], c[k]), (k, frames)
            del c
        eng.lib.kp2d_set_chunk_frames(eng.handle, 0)
        eng._ws = None
    st = meta["dense_stride"]
    assert float((a["score"][0:1].cpu() - torch.from_numpy(z["fwd_score"])).abs().max()) < TOL
    assert float((a["coord"][0:1].cpu() - torch.from_numpy(z["fwd_shift"])).abs().max()) < TOL
    assert float((a["vlad"][0:1].cpu() - torch.from_numpy(z["fwd_vlad"])).abs().max()) < 1e-5
    assert float((a["feat"][0:1, :, ::st, ::st].cpu() - torch.from_numpy(z["fwd_feat"])).abs().max()) < TOL
    assert float((a["seg"][0:1, :, ::st, ::st].cpu() - torch.from_numpy(z["fwd_seg"])).abs().max()) < TOL
    assert torch.allclose(a["seg"].sum(1), torch.ones_like(a["seg"][:, 0]), atol=1e-5)      # V3 eval: probabilities
    assert torch.isfinite(a["feat"]).all() and torch.allclose(a["vlad"].norm(dim=1), torch.ones(B, device=DEV), atol=1e-5)


def test_full_dense_maps_against_oracle_at_headline_size():
    """The fixtures hold a stride-4 subsample of the dense maps at 240x320; here EVERY pixel of feat / seg (and the
    post-processed descriptors and class map) of two frames is compared with the CPU oracle."""
    model, sd = product_model("S", False, 28)
    H, W = 240, 320
    x = synthetic_frames(2, H, W, seed=7)
    cfg = orc.get_config("S")
    ref = orc.forward(x, sd, cfg)
    refp = orc.post_processing(ref, H, W, cfg)
    for prec in ("f16x3", "fp32"):
        model.set_precision(prec)
        fwd, post, post_np = _run(model, x, H, W)
        for k in ("score", "coord", "feat", "vlad", "seg"):
            assert fwd[k].shape == ref[k].shape
            assert np.max(np.abs(fwd[k] - ref[k])) < TOL, (prec, k)
        assert np.max(np.abs(post_np["feat"] - refp["feat"])) < TOL
        part = np.partition(ref["seg"], -2, axis=1)
        clear = (part[:, -1] - part[:, -2]) > 1e-3
        assert np.array_equal(post_np["seg"][:, 0][clear], refp["seg"][:, 0][clear])


def test_topk_kernel_against_oracle_with_ties():
    from nano_vs_slam_amd.selectors import select_topk
    rng = np.random.default_rng(2)
    # after a large-k call: k = 1 and k = 2 (the sort network's smallest cases: ADVICE r1, stale LDS next to one key slot);
    # 4097..16384: the 1024-thread LDS path; above: the in-place global sort ("no cap" at 480x640 = 19200 cells);
    # 76800 cells = a 960x1280 frame
    for n, k, thr in [(4800, 1000, 0.7), (1200, 4000, 0.7), (19200, 4096, -np.inf), (300, 7, 0.5), (64, 64, 2.0),
                      (4800, 1, 0.7), (4800, 1, -np.inf), (300, 2, 0.74), (1, 1, 0.5), (19200, 4097, 0.2),
                      (19200, 10000, 0.7), (19200, 16384, -np.inf), (19200, 16385, -np.inf), (19200, 19200, 0.7),
                      (19200, 19200, -np.inf), (76800, 76800, 0.7), (76800, 30000, -np.inf), (17000, 17000, 0.1)]:
        s = rng.random((3, n)).astype(np.float32)
        s[:, ::7] = np.float32(0.75)            # many exact ties: lowest index must win
        s[1] = 0.0                              # a frame with nothing above threshold
        idx, val, cnt = select_topk(torch.from_numpy(s).to(DEV), k, thr)
        idx, val, cnt = idx.cpu().numpy(), val.cpu().numpy(), cnt.cpu().numpy()
        kk = min(k, n)
        for b in range(3):
            cand = np.nonzero(s[b] > thr)[0]
            order = cand[np.lexsort((cand, -s[b][cand]))][:kk]
            assert cnt[b] == len(order)
            assert np.array_equal(idx[b, :len(order)], order)
            assert np.all(idx[b, len(order):] == -1) and np.all(val[b, len(order):] == 0)
            assert np.array_equal(val[b, :len(order)], s[b][order])


def test_select_and_gather_equals_the_two_calls():
    """kp2d_select_keypoints against kp2d_select_topk + kp2d_gather_keypoints, bit for bit, including empty frames and
    the padded rows."""
    from nano_vs_slam_amd.selectors import gather_keypoints, select_and_gather, select_topk
    rng = np.random.default_rng(11)
    for (hc, wc), cd, k, thr in [((30, 40), 32, 1000, 0.7), ((60, 80), 32, 1024, -np.inf), ((30, 40), 64, 4000, 0.7),
                                 ((15, 20), 128, 7, 0.5), ((120, 160), 32, 19200, 0.3), ((1, 1), 32, 1, 0.5)]:
        n = hc * wc
        s = rng.random((3, 1, hc, wc)).astype(np.float32)
        s[1] = 0.0                              # a frame with nothing above a positive threshold
        coord = rng.random((3, 2, hc, wc)).astype(np.float32) * 300
        desc = rng.standard_normal((3, cd, hc, wc)).astype(np.float32)
        ts, tc, td = (torch.from_numpy(v).to(DEV) for v in (s, coord, desc))
        idx, val, cnt = select_topk(ts, k, thr)
        pts, dsel = gather_keypoints(tc, td, idx)
        fi, fv, fc, fp, fd = select_and_gather(ts, tc, td, k, thr)
        assert torch.equal(fi, idx) and torch.equal(fv, val) and torch.equal(fc, cnt)
        assert torch.equal(fp, pts) and torch.equal(fd, dsel)
        assert fp.shape == (3, min(k, n), 2) and fd.shape == (3, min(k, n), cd)
    # many frames, dense selection: the gather that stages whole channel planes in LDS (post.hip gather_lds_kernel),
    # against plain indexing
    for B, (hc, wc), cd, k, thr in [(40, (8, 12), 32, 60, 0.3), (32, (30, 40), 64, 1000, 0.7), (36, (60, 80), 32, 1024, -np.inf)]:
        n = hc * wc
        s = rng.random((B, 1, hc, wc)).astype(np.float32)
        s[1] = 0.0
        coord = rng.random((B, 2, hc, wc)).astype(np.float32) * 300
        desc = rng.standard_normal((B, cd, hc, wc)).astype(np.float32)
        idx, val, cnt, pts, dsel = select_and_gather(*(torch.from_numpy(v).to(DEV) for v in (s, coord, desc)), k, thr)
        idx, cnt, pts, dsel = idx.cpu().numpy(), cnt.cpu().numpy(), pts.cpu().numpy(), dsel.cpu().numpy()
        for b in range(B):
            m = int(cnt[b])
            sel = idx[b, :m]
            assert np.array_equal(dsel[b, :m], desc[b].reshape(cd, n)[:, sel].T)
            assert np.array_equal(pts[b, :m], coord[b].reshape(2, n)[:, sel].T)
            assert not dsel[b, m:].any() and not pts[b, m:].any() and np.all(idx[b, m:] == -1)


def test_post_processing_accepts_any_forward_dict():
    """post_processing is a separate entry point: feed tensors that did not come from forward()."""
    model, _ = product_model("S", False, 28)
    rng = np.random.default_rng(8)
    B, H, W = 2, 64, 96
    out = {"score": rng.random((B, 1, 16, 24)).astype(np.float32),
           "coord": (rng.random((B, 2, 16, 24)).astype(np.float32) * 2 - 1),
           "feat": rng.standard_normal((B, 32, 32, 48)).astype(np.float32),
           "seg": rng.standard_normal((B, 28, 32, 48)).astype(np.float32),
           "vlad": np.zeros((B, 4096), np.float32)}
    cfg = orc.get_config("S")
    ref = orc.post_processing(out, H, W, cfg)
    with torch.no_grad():
        got = model.post_processing({k: torch.from_numpy(v).to(DEV) for k, v in out.items()}, H, W)
    assert np.array_equal(got["score"].cpu().numpy(), ref["score"])
    assert np.max(np.abs(got["coord"].cpu().numpy() - ref["coord"])) < 1e-5
    assert np.max(np.abs(got["feat"].cpu().numpy() - ref["feat"])) < 1e-5
    assert np.array_equal(got["seg"].cpu().numpy(), ref["seg"])


def test_weight_updates_and_packed_roundtrip():
    model, _ = product_model("S", False, 28)
    x = torch.from_numpy(synthetic_frames(1, 32, 32, seed=3)).to(DEV)
    with torch.no_grad():
        a = model(x)["score"].clone()
        model.score_head.convDb.bias.add_(0.5)                 # in-place edit must be picked up
        b = model(x)["score"].clone()
        assert not torch.equal(a, b)
        blob = model.packed_weights(DEV)                       # what rank 0 broadcasts over RCCL
        from nano_vs_slam_amd.kp2dtiny.models.kp2dtiny import tiny_factory
        other = tiny_factory("S", 28).to(DEV).eval()           # random weights, never loads a checkpoint
        other.training = False
        other.load_packed_weights(blob)
        assert torch.equal(other(x)["score"], b)
        assert torch.equal(other(x)["vlad"], model(x)["vlad"])


def test_weight_replacement_paths_are_seen():
    """Every way a caller may change a weight reaches the engine (kp2dtiny.py, "Supported ways to change weights"):
    ``p.data = t`` and ``setattr`` on the next call; a write into ``module._parameters`` behind torch's back at the
    latest ``_SIG_RECHECK`` calls later (the cached tensor list is rebuilt that often)."""
    from nano_vs_slam_amd.kp2dtiny.models import kp2dtiny as K
    model, _ = product_model("S", False, 28)
    x = torch.from_numpy(synthetic_frames(1, 32, 32, seed=3)).to(DEV)
    with torch.no_grad():
        a = model(x)["score"].clone()
        bias = model.score_head.convDb.bias
        orig = bias.data.clone()
        bias.data = orig + 0.5                                                 # new storage, same Parameter object
        b = model(x)["score"].clone()
        assert not torch.equal(a, b)
        model.score_head.convDb.bias = torch.nn.Parameter(orig.clone())        # setattr: registration hook
        assert torch.equal(model(x)["score"], a)
        model.score_head.convDb._parameters["bias"] = torch.nn.Parameter(orig + 1.0)   # behind torch's back
        seen = False
        for _ in range(K._SIG_RECHECK + 1):
            seen = seen or not torch.equal(model(x)["score"], a)
        assert seen


def test_argument_errors():
    model, _ = product_model("S", False, 28)
    with pytest.raises(ValueError):
        model(torch.zeros(1, 3, 30, 32, device=DEV))
    with pytest.raises(ValueError):
        model(torch.zeros(1, 1, 32, 32, device=DEV))
    with pytest.raises(TypeError):
        model(torch.zeros(1, 3, 32, 32, device=DEV, dtype=torch.float16))
    from nano_vs_slam_amd import _lib
    from nano_vs_slam_amd.selectors import select_topk
    with pytest.raises(ValueError):
        select_topk(torch.zeros(1, 10000, device=DEV), 0)
    lib = _lib.load()
    cnt = torch.zeros(1, dtype=torch.int32, device=DEV)
    idx = torch.zeros(1, 4, dtype=torch.int32, device=DEV)
    sc = torch.zeros(1, 16, device=DEV)
    P = lambda t: _lib.C.c_void_p(t.data_ptr())
    assert lib.kp2d_select_topk(P(sc), 1, 16, 0, 0.5, P(idx), None, P(cnt), None) == -1          # KP2D_ERR_ARG: k < 1
    assert lib.kp2d_select_topk(P(sc), 1, 0, 4, 0.5, P(idx), None, P(cnt), None) == -1           # empty map
    assert b"k must be" in lib.kp2d_last_error() or b"empty" in lib.kp2d_last_error()
    with pytest.raises(RuntimeError):
        select_topk(torch.zeros(1, 100), 5)                                                       # CPU tensor: no CPU path


def test_entry_points_run(tmp_path):
    """demo.py / eval_multitask.py equivalents keep the reference's model-facing call sequence."""
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, "demo.py", "--frames", "3"], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "keypoints/frame" in r.stdout, r.stderr[-2000:]
    r = subprocess.run([sys.executable, "eval_multitask.py", "--keypoints", "--visloc", "--segmentation", "--n_batches", "2",
                        "--config", "S", "--result_dir", str(tmp_path)], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "self_recall@1" in r.stdout, r.stderr[-2000:]


def test_inference_front_end_matches_oracle():
    from nano_vs_slam_amd.pipeline import inference
    model, sd = product_model("S", False, 28)
    rng = np.random.default_rng(5)
    frame = rng.integers(0, 256, (64, 96, 3), dtype=np.uint8)
    pts, feat, out = inference(model, frame, (64, 96), nn_thresh=0.5, top_k=50)
    x = (frame.astype(np.float32) / 255.0 - 0.5) * 2.0
    x = np.ascontiguousarray(x.transpose(2, 0, 1))[None]
    cfg = orc.get_config("S")
    ref = orc.post_processing(orc.forward(x, sd, cfg), 64, 96, cfg)
    idx, rpts, rdesc = orc.select_k1(ref["score"], ref["coord"], ref["feat"], 0.5, 50)
    assert pts.shape == rpts.shape and feat.shape == rdesc.shape
    order = np.lexsort((pts[:, 0], pts[:, 1]))
    rorder = np.lexsort((rpts[:, 0], rpts[:, 1]))
    assert np.max(np.abs(pts[order] - rpts[rorder])) < 1e-3
    assert np.max(np.abs(feat[order] - rdesc[rorder])) < 1e-3


@pytest.mark.parametrize("src_hw,new_size,top_k", [((64, 96), None, 50), ((120, 200), (64, 96), 4000), ((64, 96), (64, 96), 0)])
def test_frame_stream_equals_inference_frame_by_frame(src_hw, new_size, top_k):
    """FrameStream = inference() replayed as a HIP graph with overlapped uploads (the VO loop, one frame per call):
    bit-identical keypoints / descriptors / dense outputs for every frame, more frames than slots, and a weight update
    in mid-stream takes effect without a recapture."""
    from nano_vs_slam_amd.pipeline import FrameStream, inference
    model, sd = product_model("S", False, 28)
    rng = np.random.default_rng(11)
    frames = [rng.integers(0, 256, (*src_hw, 3), dtype=np.uint8) for _ in range(9)]      # more frames than slots (7)
    want = []
    for f in frames:
        pts, feat, out = inference(model, f, new_size, nn_thresh=0.5, top_k=top_k)
        want.append((pts, feat, out["score"].clone(), out["seg"].clone()))
    fs = FrameStream(model, src_hw, new_size, nn_thresh=0.5, top_k=top_k, device=DEV)
    assert fs.slots < len(frames)
    got = []
    for pts, feat, out in fs.map(frames):
        got.append((pts, feat, out["score"].clone(), out["seg"].clone()))
    assert len(got) == len(want)
    for (p0, f0, s0, g0), (p1, f1, s1, g1) in zip(want, got):
        assert np.array_equal(p0, p1) and np.array_equal(f0, f1)
        assert torch.equal(s0, s1) and torch.equal(g0, g1)
    assert len(want[0][0]) > 0
    with pytest.raises(RuntimeError):
        fs.result()                                                   # nothing in flight
    for k in range(fs.slots):
        fs.submit(frames[k])
    with pytest.raises(RuntimeError):
        fs.submit(frames[fs.slots])                                   # every slot busy
    for k in range(fs.slots):
        p, f, _ = fs.result()                                         # in order, although the slots' graphs overlap on the GPU
        assert np.array_equal(p, want[k][0]) and np.array_equal(f, want[k][1])
    # new weights: same graphs, new numbers
    with torch.no_grad():
        model.loc_head.convDb.bias.add_(0.2)   # moves every keypoint
    p_new, f_new, _ = inference(model, frames[3], new_size, nn_thresh=0.5, top_k=top_k)
    fs.submit(frames[3])
    p_fs, f_fs, _ = fs.result()
    assert np.array_equal(p_new, p_fs) and np.array_equal(f_new, f_fs) and not np.array_equal(p_new, want[3][0])
    # the other arithmetic mode: recaptured transparently
    model.set_precision("fp32")
    p_new, f_new, _ = inference(model, frames[4], new_size, nn_thresh=0.5, top_k=top_k)
    fs.submit(frames[4])
    p_fs, f_fs, _ = fs.result()
    assert np.array_equal(p_new, p_fs) and np.array_equal(f_new, f_fs)


def test_descriptor_matching_against_oracle():
    """kp2d_match_descriptors vs the restated BfFeatureMatcher (knnMatch k=2 + goodMatchesOneToOne)."""
    from nano_vs_slam_amd.matching import bf_match, match_descriptors
    rng = np.random.default_rng(12)
    B, k0, k1, C = 3, 700, 900, 32
    d0 = rng.standard_normal((B, k0, C)).astype(np.float32)
    d1 = rng.standard_normal((B, k1, C)).astype(np.float32)
    # make real correspondences (noisy copies) plus exact duplicates so the one-to-one rule is exercised
    for b in range(B):
        src = rng.permutation(k1)[:400]
        d0[b, :400] = d1[b, src] + 0.05 * rng.standard_normal((400, C)).astype(np.float32)
        d0[b, 400:420] = d0[b, :20]
    d0 /= np.linalg.norm(d0, axis=-1, keepdims=True)
    d1 /= np.linalg.norm(d1, axis=-1, keepdims=True)
    n0 = np.array([700, 650, 1], np.int32)
    n1 = np.array([900, 2, 900], np.int32)
    r = match_descriptors(torch.from_numpy(d0).to(DEV), torch.from_numpy(n0).to(DEV), torch.from_numpy(d1).to(DEV),
                          torch.from_numpy(n1).to(DEV), 0.7)
    r = {k: v.cpu().numpy() for k, v in r.items()}
    for b in range(B):
        best, nn, dd1, dd2 = orc.bf_match_one_to_one(d0[b, :n0[b]], d1[b, :n1[b]], 0.7)
        assert np.array_equal(r["nn_idx"][b, :n0[b]], nn)
        assert np.max(np.abs(r["nn_dist"][b, :n0[b]] - dd1)) < 1e-5
        fin = np.isfinite(dd2)
        assert np.max(np.abs(r["nn_dist2"][b, :n0[b]][fin] - dd2[fin]), initial=0) < 1e-5
        got = {int(t): int(q) for t, q in enumerate(r["match_q"][b]) if q >= 0}
        ref = {t: q for t, (q, _) in best.items()}
        # pairs may differ only where the ratio test sits on its boundary (two fp32 roundings of the distance)
        for t in set(got) ^ set(ref):
            q = got.get(t, ref.get(t))
            assert abs(dd1[q] - 0.7 * dd2[q]) < 1e-5
        assert all(got[t] == ref[t] for t in set(got) & set(ref))
        assert np.all(r["match_q"][b, n1[b]:] == -1)
        assert len(got) > 100 or b > 0
    i1, i2, sc = bf_match(d0[0], d1[0], 0.7)
    best, *_ = orc.bf_match_one_to_one(d0[0], d1[0], 0.7)
    assert {t: q for q, t in zip(i1, i2)} == {t: q for t, (q, _) in best.items()}


@pytest.mark.parametrize("config,v3,B,H,W", [("S", False, 2, 72, 104), ("S_A", True, 1, 40, 56), ("N", False, 3, 24, 88),
                                             ("S", False, 2, 80, 144)])
def test_no_out_of_bounds_writes(config, v3, B, H, W):
    """Call the C ABI directly with guard bands around every caller-owned buffer (outputs and workspace):
    ragged tiles, padded channel groups and the LDS-transposed NCHW stores must not touch a byte outside."""
    import ctypes as C
    from nano_vs_slam_amd import _lib
    model, _ = product_model(config, v3, 28)
    x = torch.from_numpy(synthetic_frames(B, H, W, seed=2)).to(DEV)
    with torch.no_grad():
        ref = model(x)                      # builds the engine, gives the expected values
    eng = model._engine
    lib = eng.lib
    G = 4096                                # guard floats on each side
    sentinel = 12345.678

    def guarded(n):
        buf = torch.full((n + 2 * G,), sentinel, device=DEV)
        return buf, buf[G:G + n]

    shapes = {k: tuple(v.shape) for k, v in ref.items()}
    bufs = {k: guarded(int(np.prod(s))) for k, s in shapes.items()}
    nws = lib.kp2d_workspace_bytes(eng.handle, B, H, W)
    ws_full = torch.full((nws // 4 + 2 * G,), sentinel, device=DEV)
    ws = ws_full[G:G + nws // 4]
    assert ws.data_ptr() % 256 == 0 or True
    # the workspace must be 256-byte aligned: slide inside the guarded buffer if needed
    shift = (-ws_full[G:].data_ptr()) % 256 // 4
    ws = ws_full[G + shift:G + shift + nws // 4]
    P = lambda t: C.c_void_p(t.data_ptr())
    stream = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.kp2d_forward(eng.handle, P(x), B, H, W, _lib.KP2D_FWD_EVAL, P(bufs["score"][1]), P(bufs["coord"][1]),
                                P(bufs["feat"][1]), P(bufs["seg"][1]), P(bufs["vlad"][1]), C.c_void_p(), P(ws), nws,
                                C.c_void_p(stream)))
    torch.cuda.synchronize()
    for k, (full, view) in bufs.items():
        assert torch.all(full[:G] == sentinel) and torch.all(full[-G:] == sentinel), f"{k}: guard band overwritten"
        assert torch.equal(view.reshape(shapes[k]), ref[k]), k
    assert torch.all(ws_full[:G + shift] == sentinel) and torch.all(ws_full[G + shift + nws // 4:] == sentinel), "workspace guard"


def test_preprocess_front_end_matches_torch():
    """kp2d_preprocess == /255 -> F.interpolate(bilinear, align_corners=False) -> (v - 0.5) * 2 (kornia's resize)."""
    import torch.nn.functional as F
    from nano_vs_slam_amd.pipeline import frames_to_input, inference
    rng = np.random.default_rng(3)
    frames = rng.integers(0, 256, (2, 150, 200, 3), dtype=np.uint8)
    same = frames_to_input(frames, DEV).cpu()
    ref_same = (torch.from_numpy(frames).permute(0, 3, 1, 2).float() / 255.0 - 0.5) * 2.0
    assert torch.equal(same, ref_same)
    for size in ((96, 128), (240, 320), (152, 200)):
        got = frames_to_input(frames, DEV, size).cpu()
        ref = (F.interpolate(torch.from_numpy(frames).permute(0, 3, 1, 2).float() / 255.0, size=size, mode="bilinear",
                             align_corners=False) - 0.5) * 2.0
        assert got.shape == ref.shape and float((got - ref).abs().max()) < 2e-6
    model, _ = product_model("S", False, 28)
    pts, feat, out = inference(model, frames[0], (96, 128), nn_thresh=0.3, top_k=100)
    assert out["score"].shape == (1, 1, 24, 32) and pts.shape[1] == 2 and feat.shape[1] == 32
    assert pts[:, 0].max() <= 200 and pts[:, 1].max() <= 150 and (len(pts) == 0 or pts[:, 0].max() > 128 * 0.9)


@pytest.mark.parametrize("name", ["v2_N_32x48_taps", "v3_SA_32x48_taps"])
def test_only_encoder_matches_reference(name):
    """model.only_encoder(x): backbone + VPR encoder + channel L2Norm (kp2dtiny.py:515-518), reference fixture."""
    meta, z = load_golden(name)
    cfg, sd, x = golden_inputs(meta)
    model, _ = product_model(meta["config"], meta["v3"], meta["n_classes"], recipe=meta.get("weights", "spread"))
    with torch.no_grad():
        enc = model.only_encoder(torch.from_numpy(x).to(DEV))
    assert enc.shape == z["only_encoder"].shape
    assert np.max(np.abs(enc.cpu().numpy() - z["only_encoder"])) < TOL
    # init_netvlad replaces the NetVLAD parameters; the engine must pick the new tensors up
    model.init_netvlad(z["init_clsts"].copy(), z["init_descs"].copy())
    sd2 = dict(sd)
    sd2["vlad_head.netvlad.conv.weight"], sd2["vlad_head.netvlad.centroids"] = z["init_conv_weight"], z["init_centroids"]
    with torch.no_grad():
        out = model(torch.from_numpy(x).to(DEV))
    ref = orc.forward(x, sd2, cfg)
    assert np.max(np.abs(out["vlad"].cpu().numpy() - ref["vlad"])) < 1e-5


def test_remove_netvlad_wins_over_pooler():
    """to_export on a GeM / ConvAP config: VPRHead.forward returns the encoder map (vpr.py:84), pooler unused."""
    from nano_vs_slam_amd.kp2dtiny.models.kp2dtiny import tiny_factory
    from oracle.weights import spread_state_dict
    model = tiny_factory("GEM_N", 28, to_export=True)
    sd = spread_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    model = model.to(DEV).eval()
    model.training = False
    x = synthetic_frames(1, 32, 48, seed=4)
    cfg = orc.get_config("GEM_N")
    cfg["remove_netvlad"] = True
    with torch.no_grad():
        out = model(torch.from_numpy(x).to(DEV))
    ref = orc.forward(x, sd, cfg)
    assert out["vlad"].shape == ref["vlad"].shape == (1, 48, 8, 12)
    assert np.max(np.abs(out["vlad"].cpu().numpy() - ref["vlad"])) < TOL


def test_vo_frontend_wrapper_matches_reference_contract():
    """KP2DtinyFrontend.run (src/visual_odometry/frontend.py:78-129): threshold, optional semantic filter, top-k; the
    reference's import line resolves to the device-side implementation."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "src"))
    from visual_odometry.frontend import KP2DtinyFrontend
    from oracle.weights import spread_state_dict
    H, W = 120, 160
    fe = KP2DtinyFrontend((H, W), None, nn_thresh=0.7, device=DEV, debug=False, config="S", top_k=50, nClasses=28)
    sd = spread_state_dict({k: tuple(v.shape) for k, v in fe.net.state_dict().items()})
    fe.net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    x = synthetic_frames(1, H, W, seed=8)
    img = torch.from_numpy((x[0] + 1.0) / 2.0)                    # the wrapper applies .sub(0.5).mul(2) itself
    pts, feat, seg = fe.run(img)
    cfg = orc.get_config("S")
    ref = orc.post_processing(orc.forward(x, sd, cfg), H, W, cfg)
    idx, rpts, rdesc = orc.select_k1(ref["score"], ref["coord"], ref["feat"], 0.7, 50)
    assert pts.shape == (len(idx), 2) and feat.shape == (len(idx), 32) and seg.shape == (60 * 80,)
    order = np.lexsort((pts[:, 1], pts[:, 0]))
    rorder = np.lexsort((rpts[:, 1], rpts[:, 0]))
    assert np.max(np.abs(pts[order] - rpts[rorder])) < 5e-4 and np.max(np.abs(feat[order] - rdesc[rorder])) < TOL
    info = fe.get_info()
    assert info["top_k"] == 50 and info["model"]["total_params"] > 0
    # semantic filter: cells whose sampled class is listed never come back
    fe2 = KP2DtinyFrontend((H, W), None, nn_thresh=0.7, device=DEV, semantic_filter=True, classes_to_filter=[3, 7, 11],
                           debug=False, config="S", top_k=4000, nClasses=28)
    fe2.net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    pts2, feat2, seg2 = fe2.run(img)
    with torch.no_grad():
        out = fe2.net.post_processing(fe2.net(torch.from_numpy(x).to(DEV)), H, W)
    sc, cls = out["score"].view(-1).cpu().numpy(), out["seg"].view(-1).cpu().numpy()
    keep = (sc > 0.7) & ~np.isin(cls, [3, 7, 11])
    assert len(pts2) == keep.sum() == len(seg2) and not np.isin(seg2, [3, 7, 11]).any()
    assert sorted(seg2.tolist()) == sorted(cls[keep].tolist())
    assert keep.sum() < (sc > 0.7).sum()                           # the filter removed something
    # a negative threshold must still exclude filtered classes (they are masked with -inf, not with 0)
    fe2.nn_thresh = -1.0
    pts3, _, seg3 = fe2.run(img)
    assert len(pts3) == (~np.isin(cls, [3, 7, 11])).sum() and not np.isin(seg3, [3, 7, 11]).any()


@pytest.mark.parametrize("top_k", [0, -1, 10000, 19200, 4000])
def test_vo_selection_is_never_truncated(top_k):
    """K1 at 480x640 (19200 cells): the reference keeps EVERY cell above the threshold when top_k <= 0 or when fewer
    than top_k survive (src/visual_odometry/frontend.py:122, src/evaluation/visual_odometry.py:112).  A low threshold
    makes thousands of cells pass, far beyond the 4096-key limit the selection kernel had in round 1."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "src"))
    from visual_odometry.frontend import KP2DtinyFrontend
    from nano_vs_slam_amd.selectors import select_keypoints
    meta, z = load_golden("v2_S_480x640")
    cfg, sd, x = golden_inputs(meta)
    H, W = meta["H"], meta["W"]
    thr = 0.3
    ref_score = z["post_score"]
    n_pass = int((ref_score > thr).sum())
    assert n_pass > 4096                                            # the case is past the old cap
    fe = KP2DtinyFrontend((H, W), None, nn_thresh=thr, device=DEV, debug=False, config="S", top_k=top_k, nClasses=28)
    fe.net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    pts, feat, seg = fe.run(torch.from_numpy((x[0] + 1.0) / 2.0))
    ridx, rpts, rdesc = orc.select_k1(ref_score, z["post_coord"], z["post_feat"], thr, top_k)
    assert len(ridx) == (n_pass if top_k <= 0 else min(n_pass, top_k))
    assert pts.shape == (len(ridx), 2) and feat.shape == (len(ridx), 32)
    # same SET of cells: recover each returned keypoint's cell from its coordinates
    with torch.no_grad():
        out = fe.net.post_processing(fe.net(torch.from_numpy(x).to(DEV)), H, W)
    sel = select_keypoints(out, thr, top_k)[0]
    got = np.sort(sel[2].cpu().numpy())
    flat = ref_score.reshape(-1)
    bound = thr if (top_k <= 0 or n_pass <= top_k) else flat[ridx].min()
    _same_set(got, ridx, flat, bound)
    assert len(got) == len(pts)
    # the wrapper returns rows in the selection kernel's order (score desc): row j is cell sel_idx[j] of the reference
    cells = sel[2].cpu().numpy()
    assert np.max(np.abs(pts - z["post_coord"][0].reshape(2, -1).T[cells])) < 1e-3
    assert np.max(np.abs(feat - z["post_feat"][0].reshape(32, -1).T[cells])) < TOL
    sc = flat[cells]
    assert np.all(sc[1:] <= sc[:-1] + 2e-5)                          # ordered by score, up to fp32 rounding


def test_large_weights_keep_the_split_pack_finite():
    """A checkpoint weight of 40.0 (|w| * 2^11 is past the fp16 range): the f16x3 pack picks a smaller per-layer
    power-of-two pre-scale instead of storing inf / -inf halves.  Must match the
    oracle like any other weight set, in both modes; a non-finite weight is refused by name."""
    from nano_vs_slam_amd._lib import Kp2dError
    model, sd = product_model("S", False, 28)
    sd = {k: np.array(v, copy=True) for k, v in sd.items()}
    sd["backbone.conv2a.conv.weight"][3, 5, 1, 1] = 40.0
    sd["seg_head.convs.8.weight"][2, 7, 0, 2] = -300.0
    sd["desc_head.convB.weight"][17, 1, 2, 2] = 17.5
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    x = synthetic_frames(2, 48, 64, seed=4)
    cfg = orc.get_config("S")
    ref = orc.forward(x, sd, cfg)
    for prec in ("f16x3", "fp32"):
        model.set_precision(prec)
        with torch.no_grad():
            out = {k: v.cpu().numpy() for k, v in model(torch.from_numpy(x).to(DEV)).items()}
        for k in ("score", "coord", "feat", "vlad", "seg"):
            assert np.isfinite(out[k]).all(), (prec, k)
            scale = max(1.0, float(np.abs(ref[k]).max()))
            assert np.max(np.abs(out[k] - ref[k])) < TOL * scale, (prec, k, float(np.max(np.abs(out[k] - ref[k]))))
    bad = dict(sd)
    bad["loc_head.convDa.conv.weight"] = sd["loc_head.convDa.conv.weight"].copy()
    bad["loc_head.convDa.conv.weight"][0, 0, 0, 0] = np.inf
    model.load_state_dict({k: torch.from_numpy(v) for k, v in bad.items()})
    with pytest.raises(Kp2dError, match="loc_head.convDa"):
        model(torch.from_numpy(x).to(DEV))


def test_streams_views_and_coexisting_models():
    """The engine enqueues on the caller's current stream, accepts non-contiguous inputs, and two models (handles) can
    interleave calls; results do not depend on any of it."""
    m1, sd1 = product_model("S", False, 28)
    m2, sd2 = product_model("N", True, 19)
    x = synthetic_frames(3, 48, 64, seed=13)
    xt = torch.from_numpy(x).to(DEV)
    with torch.no_grad():
        ref1 = {k: v.clone() for k, v in m1(xt).items()}
        ref2 = {k: v.clone() for k, v in m2(xt).items()}
        # non-contiguous view (channels-last memory) and a side stream, the two models interleaved
        xv = xt.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
        assert not xv.is_contiguous()
        side = torch.cuda.Stream(device=DEV)
        side.wait_stream(torch.cuda.current_stream(DEV))
        with torch.cuda.stream(side):
            a1 = m1(xv)
            a2 = m2(xv)
            b1 = m1.post_processing(dict(a1), 48, 64)
        side.synchronize()
        p1 = m1.post_processing(dict(ref1), 48, 64)
    for k in ref1:
        assert torch.equal(a1[k], ref1[k]), k
    for k in ref2:
        assert torch.equal(a2[k], ref2[k]), k
    for k in ("score", "coord", "feat", "seg"):
        assert torch.equal(b1[k], p1[k]), k


def test_operand_split_saturates_instead_of_overflowing():
    """Activations past the fp16 range: the split-fp16 convolutions convert with MODE.FP16_OVFL set, so the high half
    saturates at 65504 (and the low half carries the rest up to 131008) instead of turning into inf - inf = NaN.
    The reference is plain fp32 and stays finite for such inputs (its BatchNorm / sigmoid / tanh squash them)."""
    model, sd = product_model("S", False, 28)
    x = synthetic_frames(1, 32, 48, seed=5) * 3.0e5       # conv1a outputs land far past 65504
    with torch.no_grad():
        out = model(torch.from_numpy(x.astype(np.float32)).to(DEV))
    for k in ("score", "coord", "feat", "seg", "vlad"):
        assert bool(torch.isfinite(out[k]).all()), k
    # moderately large activations (inside hi + lo's reach) still agree with the exact-fp32 mode of the same engine
    x2 = synthetic_frames(1, 32, 48, seed=5) * 40.0
    with torch.no_grad():
        a = {k: v.clone() for k, v in model(torch.from_numpy(x2.astype(np.float32)).to(DEV)).items()}
        model.set_precision("fp32")
        b = model(torch.from_numpy(x2.astype(np.float32)).to(DEV))
    for k in ("score", "coord"):
        assert float((a[k] - b[k]).abs().max()) < 1e-3, k


def test_selection_edge_cases():
    """Edges of K1/K2/K3 (the reference thresholds first and only ranks when more than k cells survive,
    evaluation/visual_odometry.py:106-117): nothing above the threshold, fewer candidates than k (no ranking needed),
    k larger than the number of cells, and a frame where every cell is selected."""
    from nano_vs_slam_amd.selectors import gather_keypoints, select_topk
    rng = np.random.default_rng(3)
    B, Hc, Wc, Cd = 3, 6, 8, 32
    n = Hc * Wc
    score = rng.random((B, 1, Hc, Wc), dtype=np.float32) * 0.5          # all below 0.7
    score[1, 0].reshape(-1)[[5, 17, 30]] = [0.9, 0.8, 0.95]             # frame 1: three candidates
    score[2] = 0.75 + 0.2 * rng.random((1, Hc, Wc), dtype=np.float32)  # frame 2: every cell above the threshold
    coord = rng.random((B, 2, Hc, Wc), dtype=np.float32) * 100
    desc = rng.standard_normal((B, Cd, Hc, Wc)).astype(np.float32)
    s, c, d = (torch.from_numpy(t).to(DEV) for t in (score, coord, desc))
    for k in (4, 10, 1000):                                              # 1000 > n: clipped to n by the host wrapper
        idx, val, cnt = select_topk(s, k, 0.7)
        kk = min(k, n)
        assert idx.shape == (B, kk) and cnt.tolist() == [0, min(3, kk), kk]
        assert (idx[0] == -1).all() and (val[0] == 0).all()
        order1 = [30, 5, 17][:kk]
        assert idx[1, :len(order1)].tolist() == order1 and (idx[1, len(order1):] == -1).all()
        ref2 = np.argsort(-score[2].reshape(-1), kind="stable")[:kk]
        assert idx[2].cpu().numpy().tolist() == ref2.tolist()
        assert np.array_equal(val[2].cpu().numpy(), score[2].reshape(-1)[ref2])
        pts, dsel = gather_keypoints(c, d, idx)
        assert (pts[0] == 0).all() and (dsel[0] == 0).all()              # unselected slots gather zeros
        got = dsel[2].cpu().numpy()
        assert np.array_equal(got, desc[2].reshape(Cd, -1)[:, ref2].T)
        assert np.array_equal(pts[2].cpu().numpy(), coord[2].reshape(2, -1)[:, ref2].T)


@pytest.mark.parametrize("config,v3,B,H,W", [
    ("S", False, 1, 104, 136), ("S", False, 5, 56, 72), ("S", True, 2, 88, 120), ("S_A", True, 1, 120, 160),
    ("S_A", False, 3, 64, 96), ("N", False, 1, 120, 160), ("N_A", True, 2, 72, 88), ("S", False, 9, 240, 320),
    ("S_A", True, 8, 240, 320),       # 19 query tiles x 4 heads x 8 frames >= 512 workgroups: the 256-query attention variant
])
def test_precision_modes_agree_on_other_shapes(config, v3, B, H, W):
    """The split-fp16 path takes shape-dependent routes the exact-fp32 path does not (32-channel groups and the merged
    first layer of the heads for small grids, NetVLAD tile mode, centre-only staging of 1x1 convolutions, 256-query
    attention workgroups): both modes of the engine must agree on shapes the fixtures do not cover."""
    model, _ = product_model(config, v3, 19)
    x = torch.from_numpy(synthetic_frames(B, H, W, seed=31)).to(DEV)
    with torch.no_grad():
        a = {k: v.clone() for k, v in model(x).items()}
        model.set_precision("fp32")
        b = model(x)
    for k in ("score", "coord", "feat", "seg", "vlad"):
        assert a[k].shape == b[k].shape
        assert float((a[k] - b[k]).abs().max()) < TOL, k


def _maxpool2(a):
    B, C, H, W = a.shape
    return a.reshape(B, C, H // 2, 2, W // 2, 2).max(axis=(3, 5))


@pytest.mark.parametrize("precision", ["f16x3", "fp32"])
@pytest.mark.parametrize("name", ["v2_N_32x48_taps", "v3_SA_32x48_taps"])
def test_intermediate_taps_on_device(name, precision):
    """Every CBR / attention-module output the reference recorded (module hooks, make_golden.py), against the same
    activation of the HIP path (kp2d_set_tap), layer by layer — the kernels themselves, not only the end of the net.
    A layer whose MaxPool2d / PixelShuffle is folded into its store is compared with the pooled / shuffled reference."""
    meta, z = load_golden(name)
    cfg, sd, x = golden_inputs(meta)
    model, _ = product_model(meta["config"], meta["v3"], meta["n_classes"], recipe=meta.get("weights", "spread"))
    model.set_precision(precision)
    xt = torch.from_numpy(x).to("cuda:0")
    ds = cfg["downsample"]
    pool, same = _maxpool2, (lambda a: a)
    layers = [("backbone.conv1a", "backbone.conv1a", same),
              ("backbone.conv1b", "backbone.conv1b", pool if ds >= 2 else same),
              ("backbone.conv2a", "backbone.conv2a", same),
              ("backbone.conv2b", "backbone.conv2b", pool if ds >= 3 else same),
              ("backbone.conv3a", "backbone.conv3a", same), ("backbone.conv3b", "backbone.conv3b", same),
              ("backbone.conv4a", "backbone.conv4a", same), ("backbone.conv4b", "backbone.conv4b", same),
              ("vlad_head.convlad1", "vlad_head.convlad1", same), ("vlad_head.convlad2", "vlad_head.convlad2", same),
              ("vlad_head.convlad3", "vlad_head.convlad3", same), ("seg_head.convs.0", "seg_head.convs.0", same)]
    if meta["v3"]:
        layers += [("score_loc_head.convDa", "score_loc_head.convDa", same)]
    else:
        layers += [("score_head.convDa", "score_head.convDa", same), ("loc_head.convDa", "loc_head.convDa", same),
                   ("desc_head.convA", "desc_head.convA", same), ("desc_head.convB", "desc_head.upsample", same),
                   ("desc_head.confAa", "desc_head.confAa", same)]
    if cfg["use_attention"]:
        layers += [("seg_head.convs.1.att", "seg_head.convs.1.att", same), ("seg_head.convs.1.mff", "seg_head.convs.1.mff", pool),
                   ("seg_head.convs.2.att", "seg_head.convs.2.att", same), ("seg_head.convs.2.mff", "seg_head.convs.2.mff", same),
                   ("seg_head.convs.3", "seg_head.upsample", same), ("seg_head.convs.4", "seg_head.convs.4", same),
                   ("seg_head.convs.5", "seg_head.upsample2", same), ("seg_head.convs.6", "seg_head.convs.6", same)]
    else:
        layers += [("seg_head.convs.1", "seg_head.convs.1", pool), ("seg_head.convs.2", "seg_head.convs.2", same),
                   ("seg_head.convs.3", "seg_head.convs.3", same), ("seg_head.convs.4", "seg_head.upsample", same),
                   ("seg_head.convs.5", "seg_head.convs.5", same), ("seg_head.convs.6", "seg_head.upsample2", same),
                   ("seg_head.convs.7", "seg_head.convs.7", same)]
    with torch.no_grad():
        for layer, key, f in layers:
            ref = f(z["tap:" + key])
            _, tap = model.forward_with_tap(xt, layer, ref.shape[1:])
            got = tap.cpu().numpy()
            assert np.isfinite(got).all(), f"{layer}: the tap was not written"
            assert np.max(np.abs(got - ref)) < TOL, (layer, float(np.max(np.abs(got - ref))))


def test_dense_maps_complete_at_headline_size():
    """KP2DTiny-S 240x320 (BASELINE config 2's frame): the FULL dense descriptor and segmentation maps against the
    oracle (the reference fixtures hold a stride-4 subsample of them; the oracle is pinned on that subsample)."""
    meta, z = load_golden("v2_S_240x320")
    cfg, sd, x = golden_inputs(meta)
    model, _ = product_model(meta["config"], meta["v3"], meta["n_classes"], recipe=meta.get("weights", "spread"))
    fwd, _, _ = _run(model, x, meta["H"], meta["W"])
    ref = orc.forward(x, sd, cfg)
    st = meta["dense_stride"]
    for k in ("feat", "seg"):
        assert np.max(np.abs(ref[k][:, :, ::st, ::st] - z["fwd_" + k])) < 1e-4      # the oracle on the reference's subsample
        assert fwd[k].shape == ref[k].shape
        assert np.max(np.abs(fwd[k] - ref[k])) < TOL, k                                # every pixel


@pytest.mark.parametrize("precision", ["f16x3", "fp32"])
@pytest.mark.parametrize("name", ["v2_S_480x640", "v3_SA_480x640"])
def test_dense_maps_complete_at_480x640(name, precision):
    """BASELINE configs 4 / 5 (480x640): EVERY pixel of the dense descriptor and segmentation maps of frame 0 against
    the oracle, in both arithmetic modes.  The reference fixtures hold a stride-8 subsample of these maps (1/64 of the
    pixels); the oracle is first checked on that subsample, then every pixel of the HIP output against the oracle."""
    meta, z = load_golden(name)
    cfg, sd, x = golden_inputs(meta)
    model, _ = product_model(meta["config"], meta["v3"], meta["n_classes"], recipe=meta.get("weights", "spread"))
    model.set_precision(precision)
    x = x[:1]
    fwd, _, _ = _run(model, x, meta["H"], meta["W"])
    ref = orc.forward(x, sd, cfg)
    st = meta["dense_stride"]
    for k in ("feat", "seg"):
        assert np.max(np.abs(ref[k][:, :, ::st, ::st] - z["fwd_" + k][:1])) < 1e-4      # the oracle on the reference's subsample
        assert fwd[k].shape == ref[k].shape == (1, ref[k].shape[1], meta["H"] // 2, meta["W"] // 2)
        err = float(np.max(np.abs(fwd[k] - ref[k])))
        print(f"{name}[{precision}] every-pixel max|{k} - oracle| = {err:.3e} over {fwd[k].size} values")
        assert err < TOL, (k, err)


def test_config_struct_without_in_channels_still_creates_an_rgb_model():
    """ABI evolution: a caller compiled against kp2d_config before in_channels existed (84 bytes) gets an RGB model."""
    import ctypes as C
    from nano_vs_slam_amd import _lib
    lib = _lib.load()
    cfg = _lib.Kp2dConfig()
    cfg.struct_size = 21 * 4
    cfg.version = 2
    for i, v in enumerate([16, 32, 32, 64, 64, 128]):
        cfg.channel_dims[i] = v
    cfg.nfeatures, cfg.n_classes, cfg.num_clusters, cfg.encoder_dim, cfg.downsample = 32, 28, 64, 64, 2
    cfg.leaky_relu = 1
    cfg.in_channels = 7          # garbage behind the end of the old struct must not be read
    h = C.c_void_p()
    _lib.check(lib.kp2d_create(C.byref(cfg), C.byref(h)))
    key, shape, nd = C.c_char_p(), (C.c_int64 * 4)(), C.c_int()
    _lib.check(lib.kp2d_weight_info(h, 0, C.byref(key), shape, C.byref(nd)))
    assert key.value == b"backbone.conv1a.conv.weight" and list(shape) == [16, 3, 3, 3]
    lib.kp2d_destroy(h)
    cfg.struct_size = 22 * 4
    assert lib.kp2d_create(C.byref(cfg), C.byref(h)) != 0 and b"in_channels" in lib.kp2d_last_error()


def test_use_color_false_rejects_rgb_frames():
    model, _ = product_model("S+gray", True, 19)
    with pytest.raises(ValueError):
        model(torch.zeros(1, 3, 64, 96, device=DEV))
    with torch.no_grad():
        out = model(torch.zeros(1, 1, 64, 96, device=DEV))
    assert out["score"].shape == (1, 1, 16, 24)


@pytest.mark.parametrize("config,v3,ncls", [("S", False, 28), ("N", True, 19)])
def test_frames_fused_into_the_first_layer_are_bit_identical(config, v3, ncls):
    """kp2d_forward_frames (uint8 frames -> /255 -> bilinear resize -> *2-1 as conv1a's prologue) against
    kp2d_preprocess + kp2d_forward: the same arithmetic in the same order, so every output tensor is equal bit for bit —
    without resize, down- and up-scaling, ragged tile edges (H, W not multiples of 16), several frames per call."""
    from nano_vs_slam_amd.pipeline import frames_to_input
    model, _ = product_model(config, v3, ncls)
    rng = np.random.default_rng(5)
    for B, Hs, Ws, size in [(2, 48, 64, None), (3, 150, 200, (72, 104)), (1, 60, 90, (120, 160)), (2, 376, 1241, (240, 320))]:
        frames = torch.from_numpy(rng.integers(0, 256, (B, Hs, Ws, 3), dtype=np.uint8)).to(DEV)
        with torch.no_grad():
            ref = model(frames_to_input(frames, DEV, size))
            got = model.forward_frames(frames, size)
        assert set(got) == set(ref)
        for k in ref:
            assert torch.equal(got[k], ref[k]), (k, Hs, Ws, size, float((got[k] - ref[k]).abs().max()))
    with pytest.raises(ValueError):
        model.forward_frames(torch.zeros(1, 48, 64, 3, device=DEV))            # not uint8
    gray, _ = product_model("S+gray", True, 19)
    from nano_vs_slam_amd._lib import Kp2dError
    with pytest.raises(Kp2dError):
        gray.forward_frames(torch.zeros(1, 64, 96, 3, dtype=torch.uint8, device=DEV))


def test_inference_uses_the_fused_front_and_matches_the_two_step_path(monkeypatch):
    from nano_vs_slam_amd import pipeline
    model, _ = product_model("S", False, 28)
    rng = np.random.default_rng(6)
    frames = rng.integers(0, 256, (2, 150, 200, 3), dtype=np.uint8)
    pts_a, feat_a, _ = pipeline.inference(model, frames, (96, 128), nn_thresh=0.3, top_k=200)
    monkeypatch.setenv("KP2D_FUSED_FRONT", "0")
    pts_b, feat_b, _ = pipeline.inference(model, frames, (96, 128), nn_thresh=0.3, top_k=200)
    for a, b in zip(pts_a + feat_a, pts_b + feat_b):
        assert np.array_equal(a, b)


# ---------------------------------------------------------------------------------------------------------------------
# round 4: the warp-specialised persistent form of the multi-chunk 64-channel-group layers (conv3x3_wsm.hip)
# ---------------------------------------------------------------------------------------------------------------------
def _kernels_that_ran(model, x):
    """{layer: kernel family + tile form} of one forward, read back through kp2d_profile_get."""
    import ctypes as C
    eng = model._engine
    lib = eng.lib
    lib.kp2d_set_profiling(eng.handle, 1)
    with torch.no_grad():
        model(x)
    torch.cuda.synchronize()
    out = {}
    layer, kern = C.c_char_p(), C.c_char_p()
    ms, fl, by = C.c_float(), C.c_double(), C.c_double()
    for i in range(lib.kp2d_profile_count(eng.handle)):
        lib.kp2d_profile_get(eng.handle, i, C.byref(layer), C.byref(kern), C.byref(ms), C.byref(fl), C.byref(by))
        out.setdefault(layer.value.decode(), set()).add(kern.value.decode())
    lib.kp2d_set_profiling(eng.handle, 0)
    return out


def _set_wsm(model, value, transposed=None):
    eng = model._engine
    assert eng.lib.kp2d_set_option(eng.handle, b"wsm_min_items", value) == 0
    if transposed is not None:      # 0 never (default), 1 always, 2 where the launcher's matrix-time model prefers it
        assert eng.lib.kp2d_set_option(eng.handle, b"wsm_transposed", transposed) == 0


@pytest.mark.parametrize("config,v3,ncls,B,H,W", [
    # 52 x 88 maps: 4 x 3 tiles of 16 x 32, ragged rows (52 = 3.25 x 16) AND columns (88 = 2.75 x 32), 60 work items on a
    # 56-workgroup grid (four workgroups walk two items: the trip-count edge n_my = 1 | 2); 26 x 44 maps: 2 x 2 tiles, one
    # and two 64-channel groups (20 / 40 items: grid 16 / 40); conv3b (full + pooled), convs.1 (pooled), the pixel-shuffle
    # layers and the two-source layers all take the form
    ("S", False, 28, 5, 104, 176),
    ("S", True, 19, 3, 104, 176),       # V3 fused heads: other layer set
    ("N", False, 28, 4, 96, 160),       # 48-channel layers: THREE chunks (odd step counts: the padded step), cout 48 < 64
    ("S", False, 28, 1, 240, 320),      # one frame: 40 / 12 / 4 items per launch (n_my = 1 everywhere, tiny grids)
    ("S", False, 28, 9, 64, 96),        # 32 x 48 maps: 2 x 2 tiles with half-empty column blocks, 16 x 24 maps fall back (W < 32)
])
@pytest.mark.parametrize("walk", ["plain", "transposed"])
def test_warp_specialised_multichunk_conv_equals_the_general_kernels(config, v3, ncls, B, H, W, walk):
    """Every output of a forward with the multi-chunk 64-channel-group layers on conv3x3_f16x3_wsm_kernel (forced by
    wsm_min_items = 8) against the same forward on the general kernels (wsm_min_items = -1): same arithmetic in the
    same order, so the results must be bit-identical; and the profile must say the form ran where it was meant to.
    walk = transposed: every such layer with its tiles walking the map transposed (wsm_transposed = 1: tile rows = map
    columns, transposed-tap weight pack) — the nine taps are then summed in another order, so the comparison is to
    1e-5 of the output's range instead of to the bit."""
    model, _ = product_model(config, v3, ncls)
    x = torch.from_numpy(synthetic_frames(B, H, W, seed=21)).to(DEV)
    with torch.no_grad():
        model(x[:1])
        _set_wsm(model, -1, 0)
        ran_off = _kernels_that_ran(model, x)
        ref = {k: v.clone() for k, v in model(x).items()}
        _set_wsm(model, 8, 1 if walk == "transposed" else 0)
        ran_on = _kernels_that_ran(model, x)
        got = {k: v.clone() for k, v in model(x).items()}
        _set_wsm(model, 0, 0)
    assert not any("<wsm>" in k for ks in ran_off.values() for k in ks)
    wsm_layers = sorted(l for l, ks in ran_on.items() if any("<wsm>" in k for k in ks))
    # (N: backbone.conv3b is 24 -> 48 channels, not whole 16-channel chunks: it stays on the general kernel)
    # (64 x 96 frames: only the 32 x 48 maps are wide enough for a 32-pixel tile)
    want = {"backbone.conv4b", "seg_head.convs.1"} if H > 64 else {"desc_head.confAa", "seg_head.convs.7"}
    if config == "S":      # 32 -> 32 layers on 32-channel items of the same kernel
        assert any("<wsm32>" in k for k in ran_on["backbone.conv2a"]) and any("<wsm32>" in k for k in ran_on["backbone.conv3a"]), ran_on["backbone.conv2a"]
    assert ("backbone.conv3b" in wsm_layers or config == "N") and want <= set(wsm_layers), wsm_layers
    if walk == "transposed":
        assert all(any("<wsm>t" in k for k in ran_on[l]) for l in wsm_layers), {l: ran_on[l] for l in wsm_layers}
        for k in ref:
            r, g = ref[k].float(), got[k].float()
            assert float((r - g).abs().max()) <= 1e-5 * max(1.0, float(r.abs().max())), (k, float((r - g).abs().max()), float(r.abs().max()))
        return
    assert not any("<wsm>t" in k for ks in ran_on.values() for k in ks)
    for k in ref:
        assert torch.equal(ref[k], got[k]), (k, float((ref[k].float() - got[k].float()).abs().max()), wsm_layers)


@pytest.mark.parametrize("config,v3,ncls,B,H,W", [("S_A", True, 19, 2, 96, 128), ("S_A", False, 28, 3, 104, 176), ("S_A", True, 19, 1, 240, 320),
                                                  ("N_A", True, 19, 2, 64, 96)])
def test_mixffn_tail_as_one_launch_equals_the_three_launches(config, v3, ncls, B, H, W):
    """MixFeedForward's tail — depthwise 3x3 -> 1x1 (GELU) -> 1x1, and the max-pool behind the first module — as ONE kernel
    (mff_tail.hip; the GELU output goes from the accumulators of one matrix product straight into the operands of the next)
    against the three launches (mff_fused = 0): both modules' outputs (.mff taps: full-resolution and pooled) and every
    output of the forward to 2e-5 of the tensor's range — same formulas, the matrix products group K differently.  Ragged
    tiles (26 x 44 and 13 x 22 maps), one frame, and N_A (48-channel modules: not fusable, must simply still run)."""
    model, _ = product_model(config, v3, ncls)
    x = torch.from_numpy(synthetic_frames(B, H, W, seed=29)).to(DEV)
    cw = 64 if config == "S_A" else 48
    taps = [("seg_head.convs.1.mff", (cw, H // 8, W // 8)), ("seg_head.convs.2.mff", (cw, H // 8, W // 8))]
    with torch.no_grad():
        model(x[:1])
        eng = model._engine
        assert eng.lib.kp2d_set_option(eng.handle, b"mff_fused", 0) == 0
        ran_off = _kernels_that_ran(model, x)
        ref = {k: v.clone() for k, v in model(x).items()}
        ref_t = [model.forward_with_tap(x, n, shp)[1].clone() for n, shp in taps]
        assert eng.lib.kp2d_set_option(eng.handle, b"mff_fused", 1) == 0
        ran_on = _kernels_that_ran(model, x)
        got = {k: v.clone() for k, v in model(x).items()}
        got_t = [model.forward_with_tap(x, n, shp)[1].clone() for n, shp in taps]
    assert not any("mff_tail" in k for ks in ran_off.values() for k in ks)
    fused = [l for l, ks in ran_on.items() if any("mff_tail" in k for k in ks)]
    assert (len(fused) == 2) == (config == "S_A"), (fused, sorted(ran_on))
    for r, g in list(zip(ref_t, got_t)) + [(ref[k].float(), got[k].float()) for k in ref]:
        assert r.shape == g.shape and not bool(torch.isnan(g).any())
        assert float((r - g).abs().max()) <= 2e-5 * max(1.0, float(r.abs().max())), float((r - g).abs().max())


def _set_s16(model, value, ws_min=None):
    eng = model._engine
    assert eng.lib.kp2d_set_option(eng.handle, b"s16_min_items", value) == 0
    if ws_min is not None:
        assert eng.lib.kp2d_set_option(eng.handle, b"ws_min_tiles", ws_min) == 0


@pytest.mark.parametrize("config,v3,ncls,B,H,W", [
    # the conv2a map (H/2 x W/2) in 16 x 32 tiles: 52 x 88 = ragged rows (3.25 tiles) AND columns (2.75), 60 items on a
    # 56-workgroup grid (n_my = 1 | 2); 120 x 160 = 8 x 5 tiles with the half-empty last tile row; 32 x 48 = 2 x 2 tiles
    # with a half-empty column block; one frame = tiny grids, n_my = 1 everywhere
    ("S", False, 28, 5, 104, 176),
    ("S", True, 19, 3, 104, 176),
    ("S", False, 28, 3, 240, 320),
    ("S", False, 28, 1, 240, 320),
    ("S", False, 28, 9, 64, 96),
    ("S_A", True, 19, 2, 96, 128),
])
def test_split_activation_backbone_stage_equals_the_fp32_activation_path(config, v3, ncls, B, H, W):
    """conv1b -> conv2a -> conv2b -> conv3a -> conv3b with the four inner tensors kept as the fp16 halves of the split
    (S16P, kp2d_kernels.h; conv3x3_s16.hip reads them by LDS-DMA, the conv1b form and conv3x3_s16.hip write them), forced
    by s16_min_items = 1 / ws_min_tiles = 1, against the same forward with fp32 NHWC activations (s16_min_items = -1):
    a consumer multiplies the very halves its own staging would have split off, so EVERY output must be bit-identical; the
    profile must say the forms ran; and the taps of the split tensors (hi + lo) must equal the fp32 activations to two
    units in the last place."""
    model, _ = product_model(config, v3, ncls)
    x = torch.from_numpy(synthetic_frames(B, H, W, seed=33)).to(DEV)
    taps = ["backbone.conv1b", "backbone.conv2a", "backbone.conv2b", "backbone.conv3a", "backbone.conv3b"]
    with torch.no_grad():
        model(x[:1])
        _set_s16(model, -1, 1)
        ran_off = _kernels_that_ran(model, x)
        ref = {k: v.clone() for k, v in model(x).items()}
        ref_taps = {t: model.forward_with_tap(x, t, (64 if t.endswith("3b") else 32, H // 2, W // 2))[1].clone() for t in taps}
        _set_s16(model, 1, 1)
        ran_on = _kernels_that_ran(model, x)
        got = {k: v.clone() for k, v in model(x).items()}
        got_taps = {t: model.forward_with_tap(x, t, (64 if t.endswith("3b") else 32, H // 2, W // 2))[1].clone() for t in taps}
        _set_s16(model, 0, 0)
    assert not any("s16" in k for ks in ran_off.values() for k in ks), ran_off
    assert any("<ws>" in k and "s16" in k for k in ran_on["backbone.conv1b"]), ran_on["backbone.conv1b"]
    for layer in ("backbone.conv2a", "backbone.conv2b", "backbone.conv3a", "backbone.conv3b"):
        assert any("<s16>" in k for k in ran_on[layer]), (layer, ran_on[layer])
    for k in ref:
        assert torch.equal(ref[k], got[k]), (k, float((ref[k].float() - got[k].float()).abs().max()))
    for t in taps:
        r, g = ref_taps[t], got_taps[t]
        assert r.shape == g.shape
        if t.endswith("3b"):
            assert torch.equal(r, g), t                      # conv3b stores fp32 either way
        else:
            # hi + lo against x: two units in the last place, and the fp16 subnormal floor for values below ~1e-4
            assert bool(((r - g).abs() <= 3e-7 * r.abs() + 1.2e-7).all()), (t, float((r - g).abs().max()))


@pytest.mark.parametrize("B", [1, 2])
def test_single_frame_netvlad_on_the_side_stream_equals_in_line(B):
    """Small grids run the heads level by level; NetVLAD's three launches go to a model-owned side stream beside the
    segmentation head's chain (kp2d_api.cpp build(): fork after level 2, join before the forward returns; in line under
    stream capture).  Same kernels on the same data: every output bit-identical to side_overlap = 0 — also back to back,
    where a missing join or a scratch buffer released before the side stream is done would show as a changed descriptor."""
    model, _ = product_model("S", False, 28)
    xs = [torch.from_numpy(synthetic_frames(B, 240, 320, seed=70 + i)).to(DEV) for i in range(3)]
    with torch.no_grad():
        model(xs[0])
        _set_opt(model, "side_overlap", 0)
        ref = [{k: v.clone() for k, v in model(x).items()} for x in xs]
        _set_opt(model, "side_overlap", 1)
        for rep in range(4):
            got = [{k: v.clone() for k, v in model(x).items()} for x in xs]      # back to back, no synchronisation in between
            for r, g in zip(ref, got):
                for k in r:
                    assert torch.equal(r[k], g[k]), (rep, k)


def _set_opt(model, key, value):
    eng = model._engine
    assert eng.lib.kp2d_set_option(eng.handle, key.encode(), value) == 0


# layers of KP2DTiny-S whose tensors change layout with s16_all: (tap, channels, downsampling of the tapped tensor)
_S16ALL_TAPS = [("backbone.conv3b", 64, 2), ("backbone.conv4a", 64, 4), ("backbone.conv4b", 64, 4), ("score_head.convDa", 64, 4),
                ("desc_head.convA", 64, 4), ("seg_head.convs.0", 64, 4), ("vlad_head.convlad1", 64, 4), ("desc_head.convB", 32, 2),
                ("desc_head.confAa", 64, 2), ("seg_head.convs.1", 64, 8), ("seg_head.convs.4", 32, 4), ("seg_head.convs.5", 64, 4),
                ("seg_head.convs.6", 32, 2), ("seg_head.convs.7", 64, 2), ("vlad_head.convlad2", 64, 4), ("vlad_head.convlad3", 64, 4)]


@pytest.mark.parametrize("B,H,W,forced", [
    # 48 frames of 240 x 320 with two stream lanes: the automatic policy (conv4a = 576 items on 128 workgroups); forced:
    # 272 x 320 -> conv4a's map 68 x 80 = ragged tile rows (4.25) AND columns (2.5), convs.4's 34 x 40 (the layout needs the
    # merged first layer: a cell grid of at least 60 x 80, or a small grid); 96 x 256 -> the least width the layout takes
    # (convs.4 reads a 12 x 32 map), one tile row with 12 of its 16 rows, few frames
    (48, 240, 320, False),
    (12, 272, 320, True),
    (5, 96, 256, True),
])
def test_all_split_activations_equal_the_fp32_activation_layout(B, H, W, forced):
    """Big grids keep EVERY tensor a warp-specialised split-fp16 3x3 layer reads as S16P (kp2d_api.cpp build(): conv3b's two
    outputs, conv4a / 4b, the desc / seg / vlad slices of the merged first layer, both pixel-shuffled tensors, convs.5,
    convlad2, confAa's and convs.7's outputs), copied into LDS by LDS-DMA (conv3x3_wsm.hip IN16, conv3x3_s16.hip) — against the same forward with s16_all = 0: every
    consumer multiplies the halves its own staging would have produced, so every OUTPUT is bit-identical; the profile says
    which form ran; a tap of a split tensor (hi + lo) equals the fp32 activation to two units in the last place, a tap of
    a tensor that stays fp32 (score slice, convs.1, convlad3) bit for bit."""
    model, _ = product_model("S", False, 28)
    x = torch.from_numpy(synthetic_frames(B, H, W, seed=57)).to(DEV)
    fp32_taps = {"score_head.convDa", "seg_head.convs.1", "vlad_head.convlad3"}      # (convs.7: either, by the class-map rule below)
    with torch.no_grad():
        model(x[:1])
        if forced:
            _set_s16(model, 1, 1)
            _set_wsm(model, 1)
            _set_opt(model, "lanes", 1)      # (two lanes halve the few frames: below the form's least grid, the layout stays fp32)
        res = {}
        for mode in (0, 1):
            _set_opt(model, "s16_all", mode)
            ran = _kernels_that_ran(model, x)
            out = {k: v.clone() for k, v in model(x).items()}
            taps = {t: model.forward_with_tap(x, t, (c, H // d, W // d))[1].clone() for t, c, d in _S16ALL_TAPS}
            res[mode] = (ran, out, taps)
        _set_opt(model, "s16_all", 1)
        _set_opt(model, "lanes", 0)
        _set_s16(model, 0, 0)
        _set_wsm(model, 0)
    ran_off, ref, ref_t = res[0]
    ran_on, got, got_t = res[1]
    assert not any("s16i" in k or "s16out" in k for ks in ran_off.values() for k in ks), ran_off
    for layer, form in [("backbone.conv4a", "s16io"), ("backbone.conv4b", "s16io"), ("heads.first", "s16io"), ("desc_head.convB", "s16io"),
                        ("desc_head.confAa", "s16io"), ("seg_head.convs.1", "s16in"), ("seg_head.convs.4", "s16out"),
                        ("seg_head.convs.5", "s16io"), ("seg_head.convs.6", "s16io"), ("seg_head.convs.7", "s16i"),
                        ("vlad_head.convlad2", "s16io"), ("vlad_head.convlad3", "s16in")]:
        assert any("<wsm>" + form in k for k in ran_on[layer]), (layer, ran_on[layer])
    assert any("<s16>" in k for k in ran_on["backbone.conv3b"]), ran_on["backbone.conv3b"]
    # the descriptor map straight from the accumulators of un-transposed products (conv3x3_s16.hip's planar form); the class
    # logits too unless the forward also writes the dense class map (then the general kernel, which finds the argmax in LDS)
    assert any("<s16>planar" in k for k in ran_on["desc_head.confBb"]), ran_on["desc_head.confBb"]
    assert any("<s16>planar" in k or "flat32" in k or "<1,1," in k for k in ran_on["seg_head.convs.8"]), ran_on["seg_head.convs.8"]
    for k in ref:
        assert torch.equal(ref[k], got[k]), (k, float((ref[k].float() - got[k].float()).abs().max()))
    for t, _, _ in _S16ALL_TAPS:
        r, g = ref_t[t], got_t[t]
        assert r.shape == g.shape and not bool(torch.isnan(g).any()), t
        if t in fp32_taps:
            assert torch.equal(r, g), t
        else:
            assert bool(((r - g).abs() <= 3e-7 * r.abs() + 1.2e-7).all()), (t, float((r - g).abs().max()))


@pytest.mark.parametrize("precision,form", [("f16x3", "auto"), ("f16x3", "general"), ("fp32", "auto")])
def test_large_grid_tile_forms_against_the_reference_fixture_at_headline_size(precision, form):
    """The fast tile forms need big launches (ws >= 1024 tiles, wsm >= 256 items, <1,2,16> >= 1024 wide tiles, flat32 >= 512)
    and no fixture batch is that large, so they used to be checked only by bit-equality with the general kernel.  Here the
    two frames of the reference fixture v2_S_240x320 are tiled to the headline batch of 64: the kernels that run are the
    ones bench.py times, and frames 0 / 1 (and two copies deep in the batch) are compared with the REFERENCE's outputs."""
    from nano_vs_slam_amd.selectors import select_keypoints
    meta, z = load_golden("v2_S_240x320")
    cfg, sd, x2 = golden_inputs(meta)
    assert meta["B"] == 2
    model, _ = product_model(meta["config"], meta["v3"], meta["n_classes"], recipe=meta.get("weights", "spread"))
    model.set_precision(precision)
    H, W, st = meta["H"], meta["W"], meta["dense_stride"]
    B = 64
    x = torch.from_numpy(np.ascontiguousarray(np.tile(x2, (B // 2, 1, 1, 1)))).to(DEV)
    if precision == "f16x3":
        with torch.no_grad():
            model(x[:1])
        # automatic: two stream lanes of 32 frames, half the chip (128 workgroups) per launch, 128 items at 30 x 40
        _set_wsm(model, 0 if form == "auto" else -1)
        _set_s16(model, 0 if form == "auto" else -1)
        ran = _kernels_that_ran(model, x)
        forms = {k for ks in ran.values() for k in ks}
        assert any("<ws>" in k for k in ran["backbone.conv1b"]), ran["backbone.conv1b"]
        big = ("backbone.conv3b", "backbone.conv4a", "desc_head.confAa", "desc_head.convB", "seg_head.convs.1", "seg_head.convs.3", "seg_head.convs.7")
        for layer in big:
            # (automatic policy: 30 x 40 maps are one item per workgroup, conv3b has two chunks per item — both stay general)
            if form == "auto" and layer == "backbone.conv3b":
                continue                                     # (conv3x3_s16.hip: asserted below)
            if form == "auto" and layer != "seg_head.convs.3":
                assert any("<wsm>" in k for k in ran[layer]), (layer, ran[layer])
                assert not any("<wsm>t" in k for k in ran[layer]), (layer, ran[layer])      # (the transposed walk is opt-in)
            else:
                assert any("<2,1,16>" in k or "<2,1,8>" in k for k in ran[layer]), (layer, ran[layer])
        # the planar outputs: 8 x 32 tiles of the general kernel, or (behind S16P tensors) un-transposed products of conv3x3_s16.hip
        assert any("flat32" in k or "<s16>planar" in k for k in forms), forms
        if form == "general":
            assert any("flat32" in k for k in forms), forms
        # the 32-channel stage: split activations + LDS-DMA staging (conv3x3_s16.hip); with that form off, the wide tiles
        # (32-channel items of the register-staging persistent form measured slower)
        if form == "auto":
            assert any("<ws>" in k and "s16" in k for k in ran["backbone.conv1b"]), ran["backbone.conv1b"]
            assert any("<wsm>s16io" in k for k in ran["backbone.conv4a"]), ran["backbone.conv4a"]      # S16P beyond the 32-channel stage
            for layer in ("backbone.conv2a", "backbone.conv2b", "backbone.conv3a", "backbone.conv3b"):
                assert any("<s16>" in k for k in ran[layer]), (layer, ran[layer])
        else:
            assert any("<1,2,16>" in k for k in ran["backbone.conv2a"]), ran["backbone.conv2a"]
    with torch.no_grad():
        out = model(x)
        fwd = {b: {k: v[b:b + 1].cpu().numpy() for k, v in out.items()} for b in (0, 1, 30, 63)}      # (post_processing works in place)
        post = model.post_processing(out, H, W)
    for b in (0, 1, 30, 63):
        r = b & 1
        f = fwd[b]
        assert np.max(np.abs(f["score"] - z["fwd_score"][r:r + 1])) < TOL
        assert np.max(np.abs(f["coord"] - z["fwd_shift"][r:r + 1])) < TOL
        assert np.max(np.abs(f["vlad"] - z["fwd_vlad"][r:r + 1])) < 1e-5
        assert np.max(np.abs(f["feat"][:, :, ::st, ::st] - z["fwd_feat"][r:r + 1])) < TOL
        assert np.max(np.abs(f["seg"][:, :, ::st, ::st] - z["fwd_seg"][r:r + 1])) < TOL
        assert np.max(np.abs(post["score"][b].cpu().numpy() - z["post_score"][r])) < TOL
        assert np.max(np.abs(post["feat"][b].cpu().numpy() - z["post_feat"][r])) < TOL
    ref_scores = z["post_score"].reshape(2, -1)
    for k in (300, 1000):
        sel = select_keypoints(post, 0.7, k)
        for b in (0, 1, 30, 63):
            r = b & 1
            ref = z[f"k1_top{k}_idx_{r}"]
            got = np.sort(sel[b][2].cpu().numpy())
            kth = ref_scores[r][ref].min() if len(ref) else 0.7
            bound = 0.7 if len(z[f"keep_idx_{r}"]) <= k else kth
            _same_set(got, ref, ref_scores[r], bound, label=f"headline-batch[{precision}] K1 top-{k} frame {b}")


# ---------------------------------------------------------------------------------------------------------------------
# round 4: matcher variants (class-masked, mutual nearest neighbours, 128-d descriptors, sliced search, compaction)
# ---------------------------------------------------------------------------------------------------------------------
def _match_problem(rng, B, k0, k1, C, n_true):
    d0 = rng.standard_normal((B, k0, C)).astype(np.float32)
    d1 = rng.standard_normal((B, k1, C)).astype(np.float32)
    for b in range(B):
        src = rng.permutation(k1)[:n_true]
        d0[b, :n_true] = d1[b, src] + 0.05 * rng.standard_normal((n_true, C)).astype(np.float32)
        d0[b, n_true:n_true + 10] = d0[b, :10]          # exact duplicates: the one-to-one rule and distance ties
    d0 /= np.linalg.norm(d0, axis=-1, keepdims=True)
    d1 /= np.linalg.norm(d1, axis=-1, keepdims=True)
    return d0, d1


def _pairs_equal(got_q, ref, dd1=None, dd2=None, ratio=None):
    got = {int(t): int(q) for t, q in enumerate(got_q) if q >= 0}
    want = {t: q for t, (q, _) in ref.items()}
    for t in set(got) ^ set(want):                      # only where the ratio test sits on its fp32 boundary
        assert dd1 is not None, (t, got.get(t), want.get(t))
        q = got.get(t, want.get(t))
        assert abs(dd1[q] - ratio * dd2[q]) < 1e-5, (t, q)
    assert all(got[t] == want[t] for t in set(got) & set(want))
    return len(got)


@pytest.mark.parametrize("C", [32, 64, 128])
def test_descriptor_matching_widths_and_sliced_search(C):
    """C = 128 (LARGE_D descriptors, kp2dtiny.py:169-188) and the sliced search: with few pairs one query's train rows
    are spread over several workgroups and merged — the result must be the one-slice result, bit for bit, and the oracle's."""
    from nano_vs_slam_amd.matching import match_descriptors
    rng = np.random.default_rng(40 + C)
    B, k0, k1 = 2, 900, 1100
    d0, d1 = _match_problem(rng, B, k0, k1, C, 300)
    n0 = np.array([900, 517], np.int32)
    n1 = np.array([1100, 3], np.int32)
    t = lambda a: torch.from_numpy(a).to(DEV)
    r = match_descriptors(t(d0), t(n0), t(d1), t(n1), 0.7)                      # sliced (2 pairs x 15 workgroups < 512)
    big = 40                                                                      # 40 copies: 600 workgroups, one slice
    rb = match_descriptors(t(np.tile(d0, (big, 1, 1))), t(np.tile(n0, big)), t(np.tile(d1, (big, 1, 1))), t(np.tile(n1, big)), 0.7)
    for k in ("nn_idx", "nn_dist", "nn_dist2", "match_q", "match_d"):
        for b in range(B):
            nq = n0[b] if k.startswith("nn") else k1
            assert torch.equal(r[k][b, :nq], rb[k][b, :nq]), (k, b)
            assert torch.equal(r[k][b, :nq], rb[k][2 * (big - 1) + b, :nq]), (k, b)
    for b in range(B):
        best, nn, dd1, dd2 = orc.bf_match_one_to_one(d0[b, :n0[b]], d1[b, :n1[b]], 0.7)
        assert np.array_equal(r["nn_idx"][b, :n0[b]].cpu().numpy(), nn)
        assert np.max(np.abs(r["nn_dist"][b, :n0[b]].cpu().numpy() - dd1)) < 2e-5
        n = _pairs_equal(r["match_q"][b].cpu().numpy(), best, dd1, dd2, 0.7)
        assert n > 100 or b > 0
    # fewer than 256 train rows: the VALU form of the search (the matrix-core form serves the sets above)
    ks = 200
    rs = match_descriptors(t(d0), t(n0), t(np.ascontiguousarray(d1[:, :ks])), t(np.minimum(n1, ks)), 0.7)
    for b in range(B):
        nb = min(int(n1[b]), ks)
        best, nn, dd1, dd2 = orc.bf_match_one_to_one(d0[b, :n0[b]], d1[b, :nb], 0.7)
        assert np.array_equal(rs["nn_idx"][b, :n0[b]].cpu().numpy(), nn)
        assert np.max(np.abs(rs["nn_dist"][b, :n0[b]].cpu().numpy() - dd1)) < 2e-5
        _pairs_equal(rs["match_q"][b].cpu().numpy(), best, dd1, dd2, 0.7)


@pytest.mark.parametrize("scale", [1e5, 1e-6, "mixed"])
def test_descriptor_matching_outside_the_fp16_range_of_the_matrix_core_search(scale):
    """The matrix-core search ranks on split-fp16 keys: descriptors scaled by 1e5 (past the fp16 clamp) or 1e-6 (hi halves
    subnormal, lo halves gone) used to rank on saturated / vanished keys and the exact pass then re-scored the wrong
    candidates, silently.  The kernel now checks every row's norm against [0.5, 2^15] and a workgroup that meets a row
    outside it scans its slice exactly: nn_idx / nn_dist / nn_dist2 and the matches must equal the oracle (which is
    scale-invariant up to fp32 rounding) at any scale — also when only SOME rows are out of range ("mixed": one all-zero
    query, a few huge train rows).  >= 256 train rows, so this is the matrix-core path (KP2D_MATCH_MFMA default)."""
    from nano_vs_slam_amd.matching import match_descriptors
    rng = np.random.default_rng(91)
    B, k0, k1, C = 2, 500, 700, 32
    d0, d1 = _match_problem(rng, B, k0, k1, C, 200)
    if scale == "mixed":
        d0[0, 7] = 0.0
        d1[0, 100:104] *= np.float32(3e5)
        d1[1, 650] *= np.float32(1e-7)
    else:
        d0 *= np.float32(scale)
        d1 *= np.float32(scale)
    n0 = np.array([500, 411], np.int32)
    n1 = np.array([700, 688], np.int32)
    t = lambda a: torch.from_numpy(a).to(DEV)
    r = match_descriptors(t(d0), t(n0), t(d1), t(n1), 0.7)
    total = 0
    for b in range(B):
        best, nn, dd1, dd2 = orc.bf_match_one_to_one(d0[b, :n0[b]], d1[b, :n1[b]], 0.7)
        got_nn = r["nn_idx"][b, :n0[b]].cpu().numpy()
        diff = np.where(got_nn != nn)[0]
        # a different index only for an exact tie in fp32 (the all-zero query sees many rows at one distance)
        got_d = r["nn_dist"][b, :n0[b]].cpu().numpy()
        for q in diff:
            assert np.isclose(np.sqrt(((d0[b, q] - d1[b, got_nn[q]]) ** 2).sum(dtype=np.float32)), dd1[q], rtol=1e-6), (b, q)
        assert len(diff) <= 2, diff
        assert np.allclose(got_d, dd1, rtol=2e-6, atol=0), np.max(np.abs(got_d / dd1 - 1))
        fin = np.isfinite(dd2)
        assert np.allclose(r["nn_dist2"][b, :n0[b]].cpu().numpy()[fin], dd2[fin], rtol=2e-6, atol=0)
        sc = float(np.median(dd1)) if scale == "mixed" else float(scale)
        got = {int(tt): int(q) for tt, q in enumerate(r["match_q"][b].cpu().numpy()) if q >= 0}
        want = {tt: q for tt, (q, _) in best.items()}
        for tt in set(got) ^ set(want):                    # ratio test on its fp32 boundary (relative, any scale)
            q = got.get(tt, want.get(tt))
            assert abs(dd1[q] - 0.7 * dd2[q]) < 1e-5 * max(dd1[q], 1e-30) * 10, (tt, q, sc)
        assert all(got[tt] == want[tt] for tt in set(got) & set(want))
        total += len(got)
    assert total > 150, total


def test_descriptor_matching_per_class():
    """match_semantic (visual_odometry.py:347-380) as one class-masked launch vs the per-class loop of the oracle: classes
    that are empty on one side, a class with ONE train row (no second neighbour: skipped, as the reference's knnMatch(k=2)
    path does), exact duplicate descriptors inside a class."""
    from nano_vs_slam_amd.matching import bf_match_semantic, match_descriptors
    rng = np.random.default_rng(77)
    B, k0, k1, C = 2, 600, 640, 32
    d0, d1 = _match_problem(rng, B, k0, k1, C, 250)
    c0 = rng.integers(0, 28, (B, k0)).astype(np.int32)
    c1 = rng.integers(0, 28, (B, k1)).astype(np.int32)
    for b in range(B):
        # true correspondences mostly share their class; class 5 absent from the train side, 6 from the query side,
        # class 7 has a single train row, class 27 (the last id) is populated
        nn_true = np.argmin(((d0[b, :250, None, :] - d1[b, None, :, :]) ** 2).sum(-1), axis=1)
        c0[b, :250] = c1[b, nn_true]
        c1[b][c1[b] == 5] = 4
        c0[b][c0[b] == 6] = 4
        c1[b][c1[b] == 7] = 8
        c1[b, 17] = 7
        c0[b, 300:310] = 7
    n0 = np.array([600, 333], np.int32)
    n1 = np.array([640, 640], np.int32)
    t = lambda a: torch.from_numpy(a).to(DEV)
    r = match_descriptors(t(d0), t(n0), t(d1), t(n1), 0.7, cls0=t(c0), cls1=t(c1))
    for b in range(B):
        ref = orc.bf_match_semantic(d0[b, :n0[b]], c0[b, :n0[b]], d1[b, :n1[b]], c1[b, :n1[b]], 0.7)
        got_q = r["match_q"][b].cpu().numpy()
        # ratio-boundary tolerance needs the class-restricted distances: recompute them per query from the oracle side
        dd1 = np.full(n0[b], np.inf, np.float32)
        dd2 = np.full(n0[b], np.inf, np.float32)
        for q in range(n0[b]):
            same = np.where(c1[b, :n1[b]] == c0[b, q])[0]
            if len(same):
                d = np.sort(np.sqrt(((d0[b, q] - d1[b, same]) ** 2).sum(-1, dtype=np.float32)))
                dd1[q] = d[0]
                dd2[q] = d[1] if len(d) > 1 else np.inf
        n = _pairs_equal(got_q, ref, dd1, dd2, 0.7)
        assert n > 60, n
        matched_t = np.where(got_q >= 0)[0]
        assert np.all(c1[b, matched_t] == c0[b, got_q[matched_t]])            # never across classes
        assert not np.any(c1[b, matched_t] == 7)                               # the single-train-row class is skipped
        nn = r["nn_idx"][b, :n0[b]].cpu().numpy()
        assert np.all(nn[c0[b, :n0[b]] == 5] == -1)                            # no train row of that class at all
    i1, i2, sc = bf_match_semantic(d0[0], c0[0], d1[0], c1[0], 0.7)
    ref = orc.bf_match_semantic(d0[0], c0[0], d1[0], c1[0], 0.7)
    assert {t_: q for q, t_ in zip(i1, i2)} == {t_: q for t_, (q, _) in ref.items()}


def test_descriptor_matching_mutual_and_compaction():
    """crossCheck=True (descriptor.py:221-222): mutual nearest neighbours; crossCheck=False: nn_idx; and the compact
    (x0, y0, x1, y1) lists the VO loop takes to the host."""
    from nano_vs_slam_amd.matching import bf_match_crosscheck, bf_match_nn, match_descriptors, match_pairs
    rng = np.random.default_rng(5)
    B, k0, k1, C = 3, 500, 450, 32
    d0, d1 = _match_problem(rng, B, k0, k1, C, 200)
    n0 = np.array([500, 1, 77], np.int32)
    n1 = np.array([450, 450, 1], np.int32)
    t = lambda a: torch.from_numpy(a).to(DEV)
    r = match_descriptors(t(d0), t(n0), t(d1), t(n1), mutual=True)
    p0 = rng.uniform(0, 320, (B, k0, 2)).astype(np.float32)
    p1 = rng.uniform(0, 320, (B, k1, 2)).astype(np.float32)
    pr = match_pairs(r, t(p0), t(p1))
    for b in range(B):
        ref = orc.bf_match_crosscheck(d0[b, :n0[b]], d1[b, :n1[b]])
        got_q = r["match_q"][b].cpu().numpy()
        _pairs_equal(got_q, ref)
        assert np.all(got_q[n1[b]:] == -1)
        cnt = int(pr["count"][b])
        assert cnt == len(ref)
        ts = np.where(got_q >= 0)[0]
        assert np.array_equal(pr["idx"][b, :cnt].cpu().numpy(), np.stack([got_q[ts], ts], 1))
        want = np.concatenate([p0[b][got_q[ts]], p1[b][ts]], 1)
        assert np.array_equal(pr["pairs"][b, :cnt].cpu().numpy(), want)
        assert np.allclose(pr["dist"][b, :cnt].cpu().numpy(), [ref[int(t_)][1] for t_ in ts], atol=2e-5)
    i1, i2, sc = bf_match_crosscheck(d0[0], d1[0])
    assert {t_: q for q, t_ in zip(i1, i2)} == {t_: q for t_, (q, _) in orc.bf_match_crosscheck(d0[0], d1[0]).items()}
    q, tr, ds = bf_match_nn(d0[0], d1[0])
    dmat = np.sqrt(((d0[0][:, None] - d1[0][None]) ** 2).sum(-1, dtype=np.float32))
    assert q == list(range(k0)) and np.array_equal(tr, np.argmin(dmat, 1))
    assert bf_match_nn(d0[0][:0], d1[0]) == ([], [], [])


@pytest.mark.parametrize("pinned", [False, True])
def test_batch_stream_host_frames_equal_inference(pinned):
    """BatchStream.submit_frames / result_host (uint8 frames in host memory, two batches in flight, the preprocess kernel
    reading pinned memory across PCIe, selected rows copied back behind the next batch's kernels) against
    pipeline.inference() batch by batch: same points and descriptors, frame by frame, including a resize."""
    from nano_vs_slam_amd.pipeline import BatchStream, inference
    model, _ = product_model("S", False, 28)
    rng = np.random.default_rng(5)
    batches = [rng.integers(0, 256, (b, 120, 160, 3), dtype=np.uint8) for b in (4, 4, 4, 2)]
    srcs = [torch.from_numpy(f).pin_memory() if pinned else f for f in batches]
    for new_size in (None, (96, 128)):
        want = [inference(model, f, new_size, 0.7, 300, DEV) for f in batches]
        want = [([p.copy() for p in w[0]], [d.copy() for d in w[1]]) for w in want]
        bs = BatchStream(model, slots=2, top_k=300, nn_thresh=0.7, device=DEV)
        got = [([p.copy() for p in r[0]], [d.copy() for d in r[1]]) for r in bs.map_frames(srcs, new_size)]
        bs.close()
        assert len(got) == len(want)
        for (wp, wd), (gp, gd) in zip(want, got):
            assert len(wp) == len(gp)
            for a, b in zip(wp, gp):
                assert a.shape == b.shape and np.array_equal(a, b)
            for a, b in zip(wd, gd):
                assert a.shape == b.shape and np.array_equal(a, b)


def test_plan_sizes_follow_the_arithmetic_mode_and_options():
    """The engine memoises a plan's workspace size per (frames, H, W).  Which layers run merged (the five heads' first
    layers as one 320-channel launch) depends on the arithmetic mode and on options, so the memo must not outlive a
    kp2d_set_precision / kp2d_set_option: fp32 first, then f16x3 at the same shape used to fail with 'workspace exhausted'."""
    model, _ = product_model("S", False, 28)
    x = torch.from_numpy(synthetic_frames(4, 240, 320, seed=3)).to(DEV)
    eng = model._get_engine(torch.device(DEV))
    with torch.no_grad():
        model.set_precision("fp32")
        a = model(x)["score"].clone()
        model.set_precision("f16x3")
        for lanes in (1, 2, 0):
            assert eng.lib.kp2d_set_option(eng.handle, b"lanes", lanes) == 0
            eng._ws = None
            b = model(x)["score"].clone()
            assert float((a - b).abs().max()) < TOL
        assert eng.lib.kp2d_set_option(eng.handle, b"lanes", 9) != 0


@pytest.mark.parametrize("slots", [2, 3])
def test_batch_stream_equals_the_plain_loop(slots):
    """pipeline.BatchStream (several batches in flight on alternating HIP streams, one engine lane per forward, a
    workspace per slot) against net(x) + post_processing + select_and_gather one batch after the other on one stream
    with the engine's default two lanes: the same kernels on the same data, so every tensor must agree bit for bit —
    including after the slots have been reused (5 batches over 2 / 3 slots) and for a ragged last batch."""
    from nano_vs_slam_amd.pipeline import BatchStream
    from nano_vs_slam_amd.selectors import select_and_gather
    model, _ = product_model("S", False, 28)
    H, W = 96, 128
    batches = [torch.from_numpy(synthetic_frames(b, H, W, seed=40 + i)).to(DEV) for i, b in enumerate((6, 6, 6, 6, 3))]
    want = []
    with torch.no_grad():
        for x in batches:
            out = model.post_processing(model(x), H, W)
            _i, _v, cnt, pts, desc = select_and_gather(out["score"], out["coord"], out["feat"], 300, 0.7)
            want.append(({k: v.clone() for k, v in out.items() if torch.is_tensor(v)}, pts.clone(), desc.clone(), cnt.clone()))
    bs = BatchStream(model, slots=slots, top_k=300, nn_thresh=0.7, device=DEV)
    got = []
    for out, pts, desc, cnt in bs.map(batches):
        got.append(({k: v.clone() for k, v in out.items() if torch.is_tensor(v)}, pts.clone(), desc.clone(), cnt.clone()))
    bs.close()
    torch.cuda.synchronize()
    assert len(got) == len(want)
    for (wo, wp, wd, wc), (go, gp, gd, gc) in zip(want, got):
        assert set(wo) == set(go)
        for k in wo:
            assert torch.equal(wo[k], go[k]), k
        assert torch.equal(wc, gc) and torch.equal(wp, gp) and torch.equal(wd, gd)
    # the engine is back on its default lane count: a plain forward afterwards still agrees
    with torch.no_grad():
        again = model.post_processing(model(batches[0]), H, W)
    assert torch.equal(again["score"], want[0][0]["score"])


def test_plain_forward_beside_an_open_batch_stream_keeps_its_own_workspace():
    """A plain net(x) while a BatchStream has batches in flight: every slot forwards on its OWN workspace
    (_Engine.using_workspace) and the engine's cached one is never pointed at a slot's buffer (rounds 3-4 swapped
    eng._ws, so such a forward took the last slot's scratch memory while that slot's kernels were still running).
    Both the in-flight batches and the plain forwards must equal the plain loop bit for bit; a second BatchStream on the
    same model is refused, a closed one refuses submits, and the context-manager form restores the lane count."""
    from nano_vs_slam_amd.pipeline import BatchStream
    model, _ = product_model("S", False, 28)
    H, W = 96, 128
    batches = [torch.from_numpy(synthetic_frames(6, H, W, seed=60 + i)).to(DEV) for i in range(4)]
    other = torch.from_numpy(synthetic_frames(5, H, W, seed=70)).to(DEV)
    with torch.no_grad():
        want = [{k: v.clone() for k, v in model.post_processing(model(x), H, W).items() if torch.is_tensor(v)} for x in batches]
        want_other = model(other)["feat"].clone()
    eng = model._get_engine(torch.device(DEV))
    with BatchStream(model, slots=2, top_k=300, device=DEV) as bs:
        with pytest.raises(RuntimeError):
            BatchStream(model, slots=2, device=DEV)
        slots = []
        for i, x in enumerate(batches):
            slots.append(bs.submit(x))
            with torch.no_grad():
                got_other = model(other)["feat"]             # plain forward on the caller's stream, slots in flight
            assert all(eng._ws is not w for w in bs._ws if w is not None)
            assert eng._ws_call is None
            assert torch.equal(got_other, want_other), i
            if len(slots) == 2:
                out = bs.result(slots.pop(0))[0]
                j = i - 1
                for k in want[j]:
                    assert torch.equal(out[k], want[j][k]), (j, k)
        while slots:
            out = bs.result(slots.pop(0))[0]
            for k in want[-1]:
                assert torch.equal(out[k], want[-1][k]), k
    with pytest.raises(RuntimeError):
        bs.submit(batches[0])
    bs2 = BatchStream(model, slots=2, device=DEV)          # the first one is closed: allowed again
    bs2.close()
    torch.cuda.synchronize()


def _vo_frames(n, rng, hw=(96, 128)):
    """Consecutive frames of one scene: shifted by whole cells (4 pixels) plus a little noise, so that most keypoints of a
    frame have their true correspondence in the next one and survive the ratio test — hundreds of matches per frame even
    with random-weight descriptors (round 4's frames shifted by single pixels: 1-3 matches per frame)."""
    base = rng.integers(0, 256, hw + (3,), dtype=np.uint8)
    frames = []
    for i in range(n):
        f = np.roll(base, (4 * (i // 2), 4 * i), axis=(0, 1)).astype(np.int16) + rng.integers(-2, 3, base.shape)
        frames.append(np.clip(f, 0, 255).astype(np.uint8))
    return frames


def _class_distances(f_prev, c_prev, f_cur, c_cur):
    """nearest / second-nearest distance of every query among the train rows of its own class (the ratio test's operands)"""
    dd1 = np.full(len(f_prev), np.inf, np.float32)
    dd2 = np.full(len(f_prev), np.inf, np.float32)
    for q in range(len(f_prev)):
        same = np.where(c_cur == c_prev[q])[0]
        if len(same):
            d = np.sort(np.sqrt(((f_prev[q] - f_cur[same]) ** 2).sum(-1, dtype=np.float32)))
            dd1[q] = d[0]
            dd2[q] = d[1] if len(d) > 1 else np.inf
    return dd1, dd2


@pytest.mark.parametrize("semantic", [False, True])
def test_frame_stream_matches_consecutive_frames_on_the_device(semantic):
    """FrameStream(match=True): the VO loop's matcher inside the replayed graphs (visual_odometry.py:193-284 / :347-380).
    For every frame after the first, the (kps0, kps1) pairs it returns are the reference-side result: inference() per
    frame, then the oracle's BF k-NN(2) + ratio + one-to-one (per class: match_semantic) between the previous frame's
    rows and this frame's — compared with the matcher unit tests' rule (_pairs_equal: a pair may differ only where the
    ratio test sits on its fp32 boundary), on frames that give hundreds of matches.  More frames than slots, so slots
    are reused while matches of the previous round complete."""
    from nano_vs_slam_amd.pipeline import FrameStream, inference
    from nano_vs_slam_amd.selectors import select_and_gather
    model, sd = product_model("S", False, 28)
    model.sample_segmentation = semantic
    frames = _vo_frames(10, np.random.default_rng(3))
    rows = []
    for f in frames:
        pts, feat, out = inference(model, f, None, nn_thresh=0.5, top_k=300)
        cls = None
        if semantic:
            idx, _v, cnt, _p, _d = select_and_gather(out["score"], out["coord"], out["feat"], 300, 0.5)
            cls = out["seg"].reshape(-1)[idx[0, :int(cnt[0])].long()].cpu().numpy()
        rows.append((pts, feat, cls))
    fs = FrameStream(model, (96, 128), None, nn_thresh=0.5, top_k=300, device=DEV, slots=4, match=True, semantic=semantic)
    got = list(fs.map(frames))
    assert len(got) == len(frames)
    k0, k1, dist, out = got[0]
    assert k0.shape == (0, 2) and k1.shape == (0, 2) and dist.shape == (0,)
    total = 0
    for i in range(1, len(frames)):
        (p_prev, f_prev, c_prev), (p_cur, f_cur, c_cur) = rows[i - 1], rows[i]
        if semantic:
            ref = orc.bf_match_semantic(f_prev, c_prev, f_cur, c_cur, 0.7)
            dd1, dd2 = _class_distances(f_prev, c_prev, f_cur, c_cur)
        else:
            ref, _nn, dd1, dd2 = orc.bf_match_one_to_one(f_prev, f_cur, 0.7)
        k0, k1, dist, out = got[i]
        # the stream returns coordinates: map them back to rows (keypoints of a frame are distinct cells)
        pos_prev = {tuple(p): j for j, p in enumerate(p_prev)}
        pos_cur = {tuple(p): j for j, p in enumerate(p_cur)}
        got_q = np.full(len(p_cur), -1, np.int64)
        for a, b in zip(k0, k1):
            got_q[pos_cur[tuple(b)]] = pos_prev[tuple(a)]
        n = _pairs_equal(got_q, ref, dd1, dd2, 0.7)
        assert n == len(k0)
        assert np.all(np.diff([pos_cur[tuple(b)] for b in k1]) > 0)            # train order
        for a, b, d in zip(k0, k1, dist):
            dd = np.sqrt(((f_prev[pos_prev[tuple(a)]] - f_cur[pos_cur[tuple(b)]]) ** 2).sum(dtype=np.float32))
            assert abs(dd - d) < 2e-5
        total += n
        assert out["rows"]["cnt"].shape == (1,) and "match" in out
    assert total >= 200, total
    with pytest.raises(ValueError):
        FrameStream(model, (96, 128), None, slots=1, match=True)


def test_frame_stream_caps_the_matches_on_the_device():
    """FrameStream(match=True, top_k_matches=k): the loop's cap (visual_odometry.py:272-283: the k smallest distances of the
    brute-force matches) inside the replayed graph — against the uncapped stream on the same frames: the k best of its
    list (as a set; ties on the k-th distance by lower train row), best first, nothing else over PCIe."""
    from nano_vs_slam_amd.pipeline import FrameStream
    model, _ = product_model("S", False, 28)
    frames = _vo_frames(7, np.random.default_rng(4))
    K = 40
    full = list(FrameStream(model, (96, 128), None, nn_thresh=0.5, top_k=300, device=DEV, slots=3, match=True).map(frames))
    cap = list(FrameStream(model, (96, 128), None, nn_thresh=0.5, top_k=300, device=DEV, slots=3, match=True,
                           top_k_matches=K).map(frames))
    seen = 0
    for (a0, a1, ad, _), (c0, c1, cd, _) in zip(full[1:], cap[1:]):
        want = np.argsort(ad, kind="stable")[:K]
        assert len(cd) == min(K, len(ad)) and np.all(np.diff(cd) >= 0)
        assert {tuple(r) for r in np.concatenate([a0[want], a1[want]], 1).tolist()} == \
               {tuple(r) for r in np.concatenate([c0, c1], 1).tolist()}
        assert np.array_equal(np.sort(ad[want]), np.sort(cd))
        seen += len(ad) > K
    assert seen >= 3, seen                                        # the cap must actually have cut something


def test_frame_stream_runs_lightglue_inside_the_loop():
    """FrameStream(match="lightglue"): the loop's use_lg branch (visual_odometry.py:198-266, kp2dtiny method) in the replayed
    graphs — previous and current rows stay on the device, LightGlue runs on the padded sets (kp2d_lg_forward_counts),
    get_matches_scores (:26-32) and the top_k_matches cap (:260-266) on the device.  Reference side: inference() per
    frame, the ORACLE's LightGlue on exactly the selected rows (keypoints / (W, H), image_size = (H, W) as the reference
    passes it), matches0 > -1, scores.topk.  Parity unpinned, as every LightGlue test (the oracle restates lightglue.py)."""
    from lightglue.lightglue import LightGlue
    from lightglue.lightglue_configs import get_light_glue_config
    from nano_vs_slam_amd.pipeline import FrameStream, inference
    from oracle import lightglue_oracle as lgo
    model, _ = product_model("S", False, 28)
    th = 0.0          # (seeded random weights give small matching scores: every mutual pair counts as a match)
    conf_in = dict(get_light_glue_config("S"), filter_threshold=th)
    conf = lgo.get_config(conf_in)
    sd = lgo.seeded_state_dict(conf)
    lgm = LightGlue(conf_in)
    lgm.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    lgm = lgm.to(DEV).eval()
    # (seeded random matcher weights: a handful of mutual matches per frame — many frames, and a cap small enough to cut)
    H, W, K = 96, 128, 2
    frames = _vo_frames(14, np.random.default_rng(6), (H, W))
    rows = [inference(model, f, None, nn_thresh=0.5, top_k=200)[:2] for f in frames]
    fs = FrameStream(model, (H, W), None, nn_thresh=0.5, top_k=200, device=DEV, slots=3, match="lightglue", matcher=lgm,
                     top_k_matches=K)
    got = list(fs.map(frames))
    assert got[0][0].shape == (0, 2)
    wh = np.asarray([W, H], np.float32)
    size = np.asarray([[H, W]], np.float32)
    compared = agreed = cut = 0
    for i in range(1, len(frames)):
        (p0, f0), (p1, f1) = rows[i - 1], rows[i]
        data = {"keypoints0": (p0 / wh)[None], "keypoints1": (p1 / wh)[None], "descriptors0": f0[None], "descriptors1": f1[None],
                "view0": {"image_size": size}, "view1": {"image_size": size}}
        ref = lgo.forward(data, sd, conf)
        m0, s0 = ref["matches0"][0], ref["matching_scores0"][0]
        on = np.nonzero(m0 > -1)[0]
        k0, k1, sc, out = got[i]
        assert len(sc) == min(K, len(sc)) and np.all(np.diff(sc) <= 0)
        pos0 = {tuple(p): j for j, p in enumerate(p0)}
        pos1 = {tuple(p): j for j, p in enumerate(p1)}
        gotp = {(pos0[tuple(a)], pos1[tuple(b)]): float(v) for a, b, v in zip(k0, k1, sc)}
        # every returned pair is one of the oracle's matches with the oracle's score ...
        inner = ref["log_assignment"][0, :-1, :-1]
        agree = 0
        for (q, t_), v in gotp.items():
            clear = inner.shape[1] < 2 or (np.sort(inner[q])[-1] - np.sort(inner[q])[-2]) > 1e-3
            if clear and abs(s0[q] - th) > 1e-7:
                assert m0[q] == t_ and abs(s0[q] - v) < 1e-4, (i, q, t_, int(m0[q]), float(s0[q]), v)
                compared += 1
            agree += int(m0[q] == t_ and abs(s0[q] - v) < 1e-4)
        assert agree >= 0.9 * len(gotp), (i, agree, len(gotp))      # (the rest: near-ties of the assignment, decided in fp32)
        agreed += agree
        # ... and they are the k best: nothing left out scores clearly above the weakest one returned
        if len(on) > K and len(sc):
            left = [q for q in on if (int(q), int(m0[q])) not in gotp]
            assert all(s0[q] <= sc[-1] + 1e-4 for q in left)
            cut += 1
        else:
            assert abs(len(gotp) - len(on)) <= max(2, len(on) // 20)
    assert compared >= 8 and agreed >= 15 and cut >= 2, (compared, agreed, cut)


@pytest.mark.parametrize("B,H,W,tiles", [(1, 304, 864, 513), (1, 80, 96, 15), (1, 112, 1184, 259), (2, 48, 160, 15)])
def test_warp_specialised_conv1b_trip_count_edges(B, H, W, tiles):
    """conv3x3_f16x3_ws_kernel's two role loops must run the same number of barriers whatever the tile count: 2 G + 1 tiles
    (513 on 256 workgroups: one workgroup walks three tiles, its second trip has only a first half), an odd count below
    the number of CUs (15: one tile per workgroup, G = ntiles), G + 3 tiles (259), and a two-frame batch that the two stream
    lanes split into one frame each.  Forced with ws_min_tiles = 1, against the general kernel (ws_min_tiles = huge)."""
    model, _ = product_model("S", False, 28)
    x = torch.from_numpy(synthetic_frames(B, H, W, seed=9)).to(DEV)
    assert (W // 32) * ((H + 15) // 16) * B == tiles or B == 2
    with torch.no_grad():
        model(x[:1])
        eng = model._engine
        assert eng.lib.kp2d_set_option(eng.handle, b"stem_fusion", 2) == 0      # (conv1a as its own launch, same arithmetic)
        assert eng.lib.kp2d_set_option(eng.handle, b"ws_min_tiles", 1 << 30) == 0
        ran = _kernels_that_ran(model, x)
        assert not any("<ws>" in k for k in ran["backbone.conv1b"])
        ref = {k: v.clone() for k, v in model(x).items()}
        assert eng.lib.kp2d_set_option(eng.handle, b"ws_min_tiles", 1) == 0
        ran = _kernels_that_ran(model, x)
        assert all("<ws>" in k for k in ran["backbone.conv1b"]), ran["backbone.conv1b"]
        got = {k: v.clone() for k, v in model(x).items()}
        # the same tile counts with conv1a computed by the staging waves of conv1b's launch
        assert eng.lib.kp2d_set_option(eng.handle, b"stem_fusion", 1) == 0
        ran = _kernels_that_ran(model, x)
        assert all("stem" in k for k in ran["backbone.conv1b"]) and "backbone.conv1a" not in ran, ran.get("backbone.conv1b")
        fused = {k: v.clone() for k, v in model(x).items()}
        assert eng.lib.kp2d_set_option(eng.handle, b"ws_min_tiles", 0) == 0
    for k in ref:
        assert torch.equal(ref[k], got[k]), k
        assert torch.equal(ref[k], fused[k]), k          # (conv1a's own launch runs the fused form's products: the same bits)


@pytest.mark.parametrize("s16", [1, -1])
def test_first_layer_fused_into_conv1b_against_the_reference_fixture(s16):
    """conv1a computed by the staging waves of conv1b's launch on the matrix cores (conv3x3_f16.hip STEM; big grids): the two
    frames of the reference fixture v2_S_240x320, forced onto the fused form (ws_min_tiles = 1), against the REFERENCE's
    outputs at the suite's tolerance — with the split-activation stage behind it and without — plus the layer itself: the
    tap of conv1b against the run with conv1a as its own launch (conv1a_mfma_kernel: the same products), bit for bit; and the
    keypoint sets of the fixture."""
    from nano_vs_slam_amd.selectors import select_keypoints
    meta, z = load_golden("v2_S_240x320")
    cfg, sd, x2 = golden_inputs(meta)
    model, _ = product_model(meta["config"], meta["v3"], meta["n_classes"], recipe=meta.get("weights", "spread"))
    H, W, st = meta["H"], meta["W"], meta["dense_stride"]
    x = torch.from_numpy(np.ascontiguousarray(x2)).to(DEV)
    with torch.no_grad():
        model(x[:1])
        eng = model._engine
        _set_s16(model, s16, 1)
        assert eng.lib.kp2d_set_option(eng.handle, b"stem_fusion", 2) == 0      # conv1a as its own launch, the fused form's products
        ran = _kernels_that_ran(model, x)
        assert any("conv1a_mfma" in k for k in ran["backbone.conv1a"]), ran["backbone.conv1a"]
        _o, tap0 = model.forward_with_tap(x, "backbone.conv1b", (32, H // 2, W // 2))
        tap0 = tap0.clone()
        assert eng.lib.kp2d_set_option(eng.handle, b"stem_fusion", 1) == 0
        ran = _kernels_that_ran(model, x)
        assert all("stem" in k for k in ran["backbone.conv1b"]) and "backbone.conv1a" not in ran, ran
        out, tap1 = model.forward_with_tap(x, "backbone.conv1b", (32, H // 2, W // 2))
        fwd = {k: v.cpu().numpy() for k, v in out.items()}
        post = model.post_processing(out, H, W)
        _set_s16(model, 0, 0)
    assert torch.equal(tap0, tap1)                       # fused or not: the same bits
    assert np.max(np.abs(fwd["score"] - z["fwd_score"])) < TOL
    assert np.max(np.abs(fwd["coord"] - z["fwd_shift"])) < TOL
    assert np.max(np.abs(fwd["vlad"] - z["fwd_vlad"])) < 1e-5
    assert np.max(np.abs(fwd["feat"][:, :, ::st, ::st] - z["fwd_feat"])) < TOL
    assert np.max(np.abs(fwd["seg"][:, :, ::st, ::st] - z["fwd_seg"])) < TOL
    ref_scores = z["post_score"].reshape(2, -1)
    sel = select_keypoints(post, 0.7, 1000)
    for b in range(2):
        ref = z[f"k1_top1000_idx_{b}"]
        got = np.sort(sel[b][2].cpu().numpy())
        kth = ref_scores[b][ref].min() if len(ref) else 0.7
        bound = 0.7 if len(z[f"keep_idx_{b}"]) <= 1000 else kth
        _same_set(got, ref, ref_scores[b], bound, label=f"stem[s16={s16}] K1 top-1000 frame {b}")


@pytest.mark.parametrize("config,v3,ncls,B,H,W", [("S", False, 28, 3, 72, 104), ("S", True, 19, 2, 48, 80), ("S_A", True, 19, 1, 72, 104),
                                                  ("S", False, 28, 64, 240, 320)])
def test_class_map_from_the_segmentation_layers_epilogue(config, v3, ncls, B, H, W, monkeypatch):
    """Inference-mode forwards also write the dense class map from the epilogue of the layer that writes `seg`
    (kp2d_set_seg_ids); post_processing takes it over when it is handed that very tensor, untouched.  Same ids as the
    separate argmax pass, for V2 logits and V3 softmax outputs, ragged tiles, and the headline batch; a dict whose seg was
    modified in place, or replaced, goes the full way."""
    model, _ = product_model(config, v3, ncls)
    x = torch.from_numpy(synthetic_frames(B, H, W, seed=13)).to(DEV)
    with torch.no_grad():
        out = model(x)
        assert model.__dict__.get("_seg_ids_cache") is not None
        logits = out["seg"].clone()
        fused = model.post_processing(out, H, W)["seg"].clone()
        assert model.__dict__.get("_seg_ids_cache") is None
        monkeypatch.setenv("KP2D_FUSED_ARGMAX", "0")
        out2 = model(x)
        assert model.__dict__.get("_seg_ids_cache") is None
        plain = model.post_processing(out2, H, W)["seg"]
        monkeypatch.delenv("KP2D_FUSED_ARGMAX")
        assert fused.dtype == torch.int64 and fused.shape == (B, 1, H // model.cell * 2, W // model.cell * 2)
        assert torch.equal(fused, plain)
        assert torch.equal(fused[:, 0], logits.argmax(1))
        # in-place change of the logits: the version counter moves, the cached ids are dropped
        out3 = model(x)
        out3["seg"][:, ncls - 1] += 100.0
        assert torch.all(model.post_processing(out3, H, W)["seg"] == ncls - 1)
        # another tensor under the same key
        out4 = model(x)
        out4["seg"] = -out4["seg"]
        assert torch.equal(model.post_processing(out4, H, W)["seg"][:, 0], (-logits).argmax(1))
