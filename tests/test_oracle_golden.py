"""The oracle (oracle/kp2d_oracle.py) against every golden fixture produced by the reference.

This is what pins the oracle: if these pass, a HIP result that matches the oracle on the GPU box
(where the reference cannot travel) matches the reference.
"""
import numpy as np
import pytest

from conftest import assert_topk_equivalent, golden_inputs, golden_names, load_golden
from oracle import kp2d_oracle as orc

TOL = 1e-4   # observed <= 3e-5 (make_golden.py records it in each fixture's meta)


@pytest.mark.parametrize("name", golden_names())
def test_forward_and_post_match_reference(name):
    meta, z = load_golden(name)
    cfg, sd, x = golden_inputs(meta)
    st = meta["dense_stride"]
    out = orc.forward(x, sd, cfg)
    assert np.max(np.abs(out["score"] - z["fwd_score"])) < TOL
    assert np.max(np.abs(out["coord"] - z["fwd_shift"])) < TOL
    assert np.max(np.abs(out["vlad"] - z["fwd_vlad"])) < 1e-5   # NetVLAD: ~2e-8; GeM (powf): ~3e-6
    assert np.max(np.abs(out["feat"][:, :, ::st, ::st] - z["fwd_feat"])) < TOL
    assert np.max(np.abs(out["seg"][:, :, ::st, ::st] - z["fwd_seg"])) < TOL
    if "fwd_depth" in z:
        assert np.max(np.abs(out["depth"] - z["fwd_depth"])) < TOL
    post = orc.post_processing(out, meta["H"], meta["W"], cfg)
    assert np.max(np.abs(post["score"] - z["post_score"])) < TOL
    assert np.max(np.abs(post["coord"] - z["post_coord"])) < 2e-4
    assert np.max(np.abs(post["feat"] - z["post_feat"])) < TOL
    assert post["seg"].dtype == np.int64 and post["seg"].shape == (meta["B"], 1, 2 * (meta["H"] >> cfg["downsample"]), 2 * (meta["W"] >> cfg["downsample"]))
    clear = z["seg_margin_f16"].astype(np.float32) > 1e-3
    assert np.array_equal(post["seg"][:, 0][clear], z["post_seg_u8"][:, 0][clear].astype(np.int64))
    # selectors: the kept / top-k SETS are identical
    for b in range(meta["B"]):
        sc, co, ft = post["score"][b:b + 1], post["coord"][b:b + 1], post["feat"][b:b + 1]
        assert np.array_equal(np.nonzero(sc.reshape(-1) > 0.7)[0], z[f"keep_idx_{b}"])
        for k in (300, 1000, 4000):
            idx, pts, desc = orc.select_k1(sc, co, ft, 0.7, k)
            assert np.array_equal(idx, z[f"k1_top{k}_idx_{b}"])
            assert pts.shape == (len(idx), 2) and desc.shape == (len(idx), cfg["nfeatures"])
    k3 = orc.select_k3(post["score"], post["coord"], post["feat"], k=z["k3_idx"].shape[1])
    for b in range(meta["B"]):
        assert_topk_equivalent(k3[0][b], z["post_score"][b].reshape(-1), z["k3_idx"][b])


@pytest.mark.parametrize("name", ["v2_N_32x48_taps", "v3_SA_32x48_taps"])
def test_intermediate_taps(name):
    """Every recorded intermediate activation of the reference, layer by layer."""
    meta, z = load_golden(name)
    cfg, sd, x = golden_inputs(meta)
    taps = {}
    orc.forward(x, sd, cfg, taps=taps)
    assert np.max(np.abs(taps["backbone.conv1a"] - z["tap:backbone.conv1a"])) < 1e-5
    assert np.max(np.abs(taps["backbone.skip"] - z["tap:backbone.conv3b"])) < 1e-4
    assert np.max(np.abs(taps["backbone.x"] - z["tap:backbone.conv4b"])) < 1e-4
    assert np.max(np.abs(taps["vlad_head.enc"] - z["tap:vlad_head.convlad3"])) < 1e-4
    if cfg["use_attention"]:
        for i in (1, 2):
            assert np.max(np.abs(taps[f"seg_head.convs.{i}.att"] - z[f"tap:seg_head.convs.{i}.att"])) < 1e-4
            assert np.max(np.abs(taps[f"seg_head.convs.{i}.mff"] - z[f"tap:seg_head.convs.{i}.mff"])) < 1e-4


def test_netvlad_gemm_restatement_equals_literal_form():
    """V = A X^T - rowsum(A) cent  ==  sum_s a (x - cent)   (netvlad.py:94-100)."""
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 48, 6, 9)).astype(np.float32)
    p = {"vlad_head.netvlad.conv.weight": rng.standard_normal((32, 48, 1, 1)).astype(np.float32),
         "vlad_head.netvlad.centroids": rng.standard_normal((32, 48)).astype(np.float32) * 0.3}
    a = orc.netvlad(x, p)
    b = orc.netvlad(x, p, literal=True)
    assert a.shape == (2, 32 * 48)
    assert np.max(np.abs(a - b)) < 1e-6


def test_selector_edge_cases():
    sc = np.zeros((1, 1, 4, 5), np.float32)
    co = np.zeros((1, 2, 4, 5), np.float32)
    ft = np.ones((1, 32, 4, 5), np.float32)
    idx, pts, d = orc.select_k1(sc, co, ft)              # nothing above threshold
    assert idx.size == 0 and pts.shape == (0, 2) and d.shape == (0, 32)
    sc.reshape(-1)[[3, 7, 11]] = 0.9                     # exact ties: lowest index wins
    idx, _, _ = orc.select_k1(sc, co, ft, 0.7, 2)
    assert idx.tolist() == [3, 7]
    k3 = orc.select_k3(sc, co, ft, k=4)[0]
    assert k3[0].tolist() == [3, 7, 11, 0]


def test_get_config_errors():
    with pytest.raises(ValueError):
        orc.get_config("nope")


@pytest.mark.parametrize("name", ["v2_N_32x48_taps", "v3_SA_32x48_taps"])
def test_only_encoder_and_netvlad_init_match_reference(name):
    """only_encoder (kp2dtiny.py:515-518) and NetVLAD.init_params (aggregators/netvlad.py:51-63), reference outputs."""
    meta, z = load_golden(name)
    cfg, sd, x = golden_inputs(meta)
    assert np.max(np.abs(orc.only_encoder(x, sd, cfg) - z["only_encoder"])) < 1e-5
    alpha, cent, w = orc.netvlad_init_params(z["init_clsts"].copy(), z["init_descs"].copy())
    assert abs(alpha - float(z["init_alpha"])) < 1e-9 * abs(alpha)
    assert np.array_equal(w, z["init_conv_weight"]) and np.array_equal(cent, z["init_centroids"])


def test_matcher_variants_on_hand_made_cases():
    """Known answers for the matcher restatements of the oracle (parity unpinned otherwise: cv2 is not installed here and
    the reference holds no match vectors — the restatements follow feature_matcher.py:179-209, visual_odometry.py:347-380
    and OpenCV's documented BFMatcher semantics)."""
    from oracle import kp2d_oracle as orc
    e = np.eye(4, dtype=np.float32)
    q = np.stack([e[0], e[1], e[0] * 0.9 + e[1] * 0.1, e[3]])
    t = np.stack([e[0], e[1], e[2]])
    # one-to-one: queries 0 and 2 both point at train 0; query 0 (distance 0) keeps it.  Query 3 is equidistant from all
    # three train rows: ratio test 1.0 <= 0.7 fails
    best, nn, d1, d2 = orc.bf_match_one_to_one(q, t, 0.7)
    assert best == {0: (0, 0.0), 1: (1, 0.0)} and list(nn) == [0, 1, 0, 0]
    # mutual nearest neighbours: train 2's nearest query is query 0 (lowest index among ties), whose nearest train is 0
    assert orc.bf_match_crosscheck(q, t) == {0: (0, 0.0), 1: (1, 0.0)}
    # per class: class 0 = {q0, q2} x {t0, t2}, class 1 = {q1, q3} x {t1} -> a single train row, skipped as a whole
    sem = orc.bf_match_semantic(q, [0, 1, 0, 1], t, [0, 1, 0], 0.7)
    assert sem == {0: (0, 0.0)}
    # a class without queries or without train rows contributes nothing
    assert orc.bf_match_semantic(q, [2, 2, 2, 2], t, [0, 0, 0], 0.7) == {}
