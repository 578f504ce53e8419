"""HIP LightGlue (include/kp2d_lightglue.h) against the numpy oracle.  The oracle itself is only self-consistent
(parity unpinned: the reference's lightglue.py cannot be imported here), see oracle/lightglue_oracle.py."""
import numpy as np
import pytest
import torch

from oracle import lightglue_oracle as lg
from test_lightglue_oracle import make_data

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def product(conf_in, sd):
    from lightglue.lightglue import LightGlue
    m = LightGlue(conf_in)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return m.to(DEV).eval()


def to_dev(data):
    out = {}
    for k, v in data.items():
        out[k] = {"image_size": torch.from_numpy(v["image_size"]).to(DEV)} if isinstance(v, dict) else torch.from_numpy(v).to(DEV)
    return out


def check(pred, ref, conf, th):
    la, rla = pred["log_assignment"].cpu().numpy(), ref["log_assignment"]
    assert la.shape == rla.shape
    assert np.max(np.abs(la - rla) / (1.0 + np.abs(rla))) < 5e-5
    for k in ("ref_descriptors0", "ref_descriptors1"):
        assert pred[k].shape == ref[k].shape
        assert np.max(np.abs(pred[k].cpu().numpy() - ref[k])) < 2e-4 * max(1.0, float(np.abs(ref[k]).max()))
    # matches: identical wherever the oracle's decision is not within rounding of a tie / of the threshold
    inner = rla[:, :-1, :-1]
    top2 = np.sort(inner, axis=2)[:, :, -2:]
    clear0 = (top2[:, :, 1] - top2[:, :, 0]) > 1e-3
    m0, rm0 = pred["matches0"].cpu().numpy(), ref["matches0"]
    s0, rs0 = pred["matching_scores0"].cpu().numpy(), ref["matching_scores0"]
    near_th = (np.abs(rs0 - th) < 1e-4) & (rs0 > 0)
    ok = clear0 & ~near_th
    top2c = np.sort(inner, axis=1)[:, -2:, :]
    clear1 = (top2c[:, 1] - top2c[:, 0]) > 1e-3
    # a row's mutual test also depends on its column's argmax being clear
    ok &= np.take_along_axis(clear1, inner.argmax(2), 1)
    assert ok.mean() > 0.7          # the mask must leave most rows to compare
    assert np.array_equal(m0[ok], rm0[ok])
    assert np.max(np.abs(s0[ok] - rs0[ok])) < 1e-4
    m1, rm1 = pred["matches1"].cpu().numpy(), ref["matches1"]
    agree = (m1 == rm1).mean()
    assert agree > 0.97
    # consistency of the product's own outputs: matches are mutual and one-to-one
    for b in range(m0.shape[0]):
        i = np.nonzero(m0[b] >= 0)[0]
        assert np.array_equal(m1[b][m0[b][i]], i)
        assert len(set(m0[b][i].tolist())) == len(i)
    assert pred["matches0"].dtype == torch.int64 and pred["prune0"].shape == pred["matching_scores0"].shape
    assert float(pred["prune0"].min()) == conf["n_layers"]


@pytest.mark.parametrize("name,B,M,N,th", [("S", 2, 70, 53, 0.1), ("S", 1, 300, 257, 0.0), ("F", 1, 128, 96, 0.1),
                                           ("A", 3, 17, 64, 0.1),
                                           # M == N: both images' attention runs as one launch of 2B sequences
                                           ("S", 3, 96, 96, 0.1), ("S", 1, 200, 200, 0.0),
                                           # key-split attention: several 32-key rounds per wave, ragged last wave /
                                           # a wave without keys (T = 130), 64-row tail workgroups with a ragged last one
                                           ("S", 1, 520, 410, 0.1), ("S", 2, 130, 641, 0.0), ("S", 1, 1024, 1024, 0.1)])
def test_lightglue_matches_oracle(name, B, M, N, th):
    from lightglue.lightglue_configs import get_light_glue_config
    conf_in = dict(get_light_glue_config(name), filter_threshold=th)
    conf = lg.get_config(conf_in)
    sd = lg.seeded_state_dict(conf)
    data = make_data(B, M, N, conf["input_dim"], seed=11)
    ref = lg.forward(data, sd, conf)
    model = product(conf_in, sd)
    with torch.no_grad():
        pred = model(to_dev(data))
    check(pred, ref, conf, th)


def test_lightglue_input_proj_and_keypoint_extent_size():
    """input_dim != descriptor_dim adds input_proj; image_size None -> 1 + max - min of the keypoints (:140-141)."""
    conf_in = {"input_dim": 64, "descriptor_dim": 32, "n_layers": 2, "filter_threshold": 0.05}
    conf = lg.get_config(conf_in)
    sd = lg.seeded_state_dict(conf)
    data = make_data(2, 90, 75, 64, seed=5)
    data["view0"] = {"image_size": None}
    data["view1"] = {"image_size": None}
    ref = lg.forward(data, sd, conf)
    model = product(conf_in, sd)
    dd = {k: torch.from_numpy(v).to(DEV) for k, v in data.items() if not isinstance(v, dict)}
    dd["view0"], dd["view1"] = {"image_size": None}, {"image_size": None}
    with torch.no_grad():
        pred = model(dd)
    check(pred, ref, conf, 0.05)


def test_lightglue_full_size_properties():
    """BASELINE config 5 shape (1024 keypoints per image, K3 top-k): swapping the two images transposes the assignment,
    rows/cols of exp(log_assignment) are sub-stochastic, matches are mutual."""
    from lightglue.lightglue_configs import get_light_glue_config
    conf_in = dict(get_light_glue_config("S"), filter_threshold=0.1)
    conf = lg.get_config(conf_in)
    sd = lg.seeded_state_dict(conf)
    data = make_data(2, 1024, 1000, 32, seed=2, size=(640.0, 480.0))
    model = product(conf_in, sd)
    d = to_dev(data)
    sw = {"keypoints0": d["keypoints1"], "keypoints1": d["keypoints0"], "descriptors0": d["descriptors1"],
          "descriptors1": d["descriptors0"], "view0": d["view1"], "view1": d["view0"]}
    with torch.no_grad():
        p, q = model(d), model(sw)
    la, lb = p["log_assignment"], q["log_assignment"].transpose(1, 2)
    assert torch.max(torch.abs(la - lb) / (1 + la.abs())) < 5e-5
    pe = la[:, :-1, :].exp().sum(2)          # each keypoint's assignment mass incl. the dustbin <= ~1
    assert float(pe.max()) < 1.0 + 1e-3
    m0, m1 = p["matches0"], p["matches1"]
    for b in range(2):
        i = torch.nonzero(m0[b] >= 0)[:, 0]
        assert torch.equal(m1[b][m0[b][i]], i)
    assert torch.equal(p["matches0"], q["matches1"]) or (p["matches0"] != q["matches1"]).float().mean() < 0.01


def test_lightglue_refuses_unbuilt_modes_and_bad_weights():
    from lightglue.lightglue import LightGlue
    from nano_vs_slam_amd import _lib
    conf = lg.get_config("S")
    sd = lg.seeded_state_dict(conf)
    m = product({"input_dim": 32, "descriptor_dim": 32, "n_layers": 4, "depth_confidence": 0.9}, sd)
    data = to_dev(make_data(1, 20, 20, 32, seed=1))
    with pytest.raises(NotImplementedError):
        m(data)
    with pytest.raises(NotImplementedError):
        LightGlue({"input_dim": 32, "descriptor_dim": 32, "add_scale_ori": True})
    big = LightGlue({"input_dim": 256, "descriptor_dim": 256}).to(DEV).eval()      # the upstream default width
    with pytest.raises(_lib.Kp2dError):
        big({"keypoints0": torch.zeros(1, 4, 2, device=DEV), "keypoints1": torch.zeros(1, 4, 2, device=DEV),
             "descriptors0": torch.zeros(1, 4, 256, device=DEV), "descriptors1": torch.zeros(1, 4, 256, device=DEV)})
    ok = product({"input_dim": 32, "descriptor_dim": 32, "n_layers": 4}, sd)
    keys = ok.expected_weights()
    assert keys == list(lg.state_dict_shapes(conf).items())


def test_two_view_pipeline_extractor_plus_matcher():
    """gluefactory-style two-view sequence: extractor -> K3 top-k -> LightGlue, all on the device; matching an image
    against itself must pair (almost) every keypoint with itself."""
    from lightglue.lightglue import LightGlue
    from lightglue.lightglue_configs import get_light_glue_config
    from nano_vs_slam_amd.pipeline import two_view_match
    from conftest import product_model
    net, _ = product_model("S", False, 28)
    conf = dict(get_light_glue_config("S"), filter_threshold=0.0)
    sd = lg.seeded_state_dict(lg.get_config(conf))
    matcher = product(conf, sd)
    g = torch.Generator(device=DEV).manual_seed(3)
    img = torch.rand(2, 3, 125, 163, device=DEV, generator=g)          # cropped to 120 x 160 by the extractor
    p0, p1, m = two_view_match(net, matcher, img, img, max_num_keypoints=256)
    assert p0["keypoints"].shape == (2, 256, 2) and p0["descriptors"].shape == (2, 256, 32)
    assert torch.equal(p0["keypoints"], p1["keypoints"])
    assert m["log_assignment"].shape == (2, 257, 257)
    ar = torch.arange(256, device=DEV)[None].expand(2, -1)
    matched = m["matches0"] >= 0
    assert torch.equal(m["matches0"][matched], ar[matched])            # a keypoint can only match itself
    ref = lg.forward({"keypoints0": p0["keypoints"].cpu().numpy(), "keypoints1": p1["keypoints"].cpu().numpy(),
                      "descriptors0": p0["descriptors"].cpu().numpy(), "descriptors1": p1["descriptors"].cpu().numpy(),
                      "view0": {"image_size": np.array([[160.0, 120.0]] * 2, np.float32)},
                      "view1": {"image_size": np.array([[160.0, 120.0]] * 2, np.float32)}}, sd, lg.get_config(conf))
    la, rla = m["log_assignment"].cpu().numpy(), ref["log_assignment"]
    assert np.max(np.abs(la - rla) / (1.0 + np.abs(rla))) < 5e-5


def test_cfg5_two_view_pipeline_at_full_size():
    """BASELINE configs[4]: KP2DTiny-S keypoints + LightGlue on 480x640 pairs, 1024 keypoints per view.  The extractor
    half is pinned by the reference fixture v2_S_480x640 (frame 0 of view 0 is the fixture's frame: K3 indices and
    sampled descriptors must be the reference's); the matcher half (parity unpinned) is checked
    against the oracle on the extractor's outputs and through size-independent properties."""
    from conftest import golden_inputs, load_golden, product_model, assert_topk_equivalent
    from lightglue.lightglue import LightGlue  # noqa: F401  (the reference's import line resolves)
    from lightglue.lightglue_configs import get_light_glue_config
    from nano_vs_slam_amd.pipeline import two_view_match
    meta, z = load_golden("v2_S_480x640")
    cfg, ksd, x0 = golden_inputs(meta)
    net, _ = product_model("S", False, 28)
    conf = dict(get_light_glue_config("S"), filter_threshold=0.1)
    sd = lg.seeded_state_dict(lg.get_config(conf))
    matcher = product(conf, sd)
    H, W, K = 480, 640, 1024
    img0 = torch.from_numpy((x0 + 1.0) / 2.0).to(DEV)                     # the extractor applies .sub(0.5).mul(2)
    shift = torch.roll(img0, shifts=(8, 12), dims=(2, 3))                  # view 1: the same scene, translated
    g = torch.Generator(device=DEV).manual_seed(5)
    other = torch.rand(1, 3, H, W, device=DEV, generator=g)
    image0 = torch.cat([img0, other], 0)
    image1 = torch.cat([shift, other], 0)                                  # pair 1 matches an image with itself
    p0, p1, m = two_view_match(net, matcher, image0, image1, max_num_keypoints=K)
    assert p0["keypoints"].shape == (2, K, 2) and p0["descriptors"].shape == (2, K, 32)
    assert m["log_assignment"].shape == (2, K + 1, K + 1) and m["matches0"].shape == (2, K)
    # extractor half against the reference fixture
    assert_topk_equivalent(p0["indices"][0].cpu().numpy(), z["post_score"][0].reshape(-1), z["k3_idx"][0])
    ref_desc = z["post_feat"][0].reshape(32, -1)
    idx0 = p0["indices"][0].long().cpu().numpy()
    assert np.max(np.abs(p0["descriptors"][0].cpu().numpy() - ref_desc[:, idx0].T)) < 2e-4
    ref_pts = z["post_coord"][0].reshape(2, -1)
    assert np.max(np.abs(p0["keypoints"][0].cpu().numpy() - ref_pts[:, idx0].T)) < 1e-3
    # matcher half against the oracle on the same keypoints / descriptors
    ref = lg.forward({"keypoints0": p0["keypoints"].cpu().numpy(), "keypoints1": p1["keypoints"].cpu().numpy(),
                      "descriptors0": p0["descriptors"].cpu().numpy(), "descriptors1": p1["descriptors"].cpu().numpy(),
                      "view0": {"image_size": np.array([[float(W), float(H)]] * 2, np.float32)},
                      "view1": {"image_size": np.array([[float(W), float(H)]] * 2, np.float32)}}, sd, lg.get_config(conf))
    la, rla = m["log_assignment"].cpu().numpy(), ref["log_assignment"]
    assert np.max(np.abs(la - rla) / (1.0 + np.abs(rla))) < 1e-4
    # properties: mutual one-to-one matches, sub-stochastic assignment, identical pair matches itself
    m0, m1 = m["matches0"].cpu().numpy(), m["matches1"].cpu().numpy()
    for b in range(2):
        v = np.nonzero(m0[b] >= 0)[0]
        assert np.array_equal(m1[b][m0[b][v]], v)
        assert len(np.unique(m0[b][v])) == len(v)
    P = np.exp(la[:, :-1, :-1])
    assert P.sum(2).max() <= 1 + 1e-4 and P.sum(1).max() <= 1 + 1e-4
    same = m0[1] >= 0
    assert np.array_equal(m0[1][same], np.arange(K)[same])          # identical views: a keypoint can only match itself


@pytest.mark.parametrize("name,cap,counts", [("S", 256, [(200, 131), (256, 1), (17, 256)]), ("S", 96, [(96, 96), (40, 0)]),
                                             ("F", 160, [(100, 160)])])
def test_lightglue_on_padded_keypoint_sets_equals_the_sliced_sets(name, cap, counts):
    """kp2d_lg_forward_counts (num_keypoints0 / num_keypoints1: the rows of each set that exist, the rest of the capacity is
    padding) against the ORACLE run on exactly the existing rows, pair by pair — what the reference's VO loop hands its
    matcher (visual_odometry.py:198-258: the selected rows, a new shape every frame).  Padding rows must be no keys in any
    attention and carry no assignment mass: valid rows' matches / scores as the sliced run, padding rows -1 / 0.
    (Parity unpinned as every LightGlue test: the oracle restates lightglue.py, which cannot be imported here.)"""
    from lightglue.lightglue_configs import get_light_glue_config
    th = 0.1
    conf_in = dict(get_light_glue_config(name), filter_threshold=th)
    conf = lg.get_config(conf_in)
    sd = lg.seeded_state_dict(conf)
    B = len(counts)
    data = make_data(B, cap, cap, conf["input_dim"], seed=23)
    # garbage in the padding rows (what a replayed graph's static buffers hold from earlier frames)
    rng = np.random.default_rng(5)
    for b, (n0, n1) in enumerate(counts):
        data["keypoints0"][b, n0:] = rng.random((cap - n0, 2)).astype(np.float32) * 500
        data["descriptors0"][b, n0:] = rng.standard_normal((cap - n0, conf["input_dim"])).astype(np.float32)
        data["keypoints1"][b, n1:] = rng.random((cap - n1, 2)).astype(np.float32) * 500
        data["descriptors1"][b, n1:] = rng.standard_normal((cap - n1, conf["input_dim"])).astype(np.float32)
    model = product(conf_in, sd)
    dd = to_dev(data)
    dd["num_keypoints0"] = torch.tensor([c[0] for c in counts], dtype=torch.int32, device=DEV)
    dd["num_keypoints1"] = torch.tensor([c[1] for c in counts], dtype=torch.int32, device=DEV)
    with torch.no_grad():
        pred = model(dd)
    m0, s0 = pred["matches0"].cpu().numpy(), pred["matching_scores0"].cpu().numpy()
    m1, s1 = pred["matches1"].cpu().numpy(), pred["matching_scores1"].cpu().numpy()
    compared = 0
    for b, (n0, n1) in enumerate(counts):
        assert np.all(m0[b, n0:] == -1) and np.all(s0[b, n0:] == 0) and np.all(m1[b, n1:] == -1) and np.all(s1[b, n1:] == 0)
        assert np.all(m0[b, :n0] < n1) and np.all(m1[b, :n1] < n0)
        if n0 == 0 or n1 == 0:
            assert np.all(m0[b] == -1) and np.all(m1[b] == -1)
            continue
        one = {"keypoints0": data["keypoints0"][b:b + 1, :n0], "keypoints1": data["keypoints1"][b:b + 1, :n1],
               "descriptors0": data["descriptors0"][b:b + 1, :n0], "descriptors1": data["descriptors1"][b:b + 1, :n1],
               "view0": {"image_size": data["view0"]["image_size"][b:b + 1]}, "view1": {"image_size": data["view1"]["image_size"][b:b + 1]}}
        ref = lg.forward(one, sd, conf)
        inner = ref["log_assignment"][0, :-1, :-1]
        la = pred["log_assignment"][b, :n0, :n1].cpu().numpy()
        assert np.max(np.abs(la - inner) / (1.0 + np.abs(inner))) < 5e-5
        if n1 > 1:
            top2 = np.sort(inner, axis=1)[:, -2:]
            clear = (top2[:, 1] - top2[:, 0]) > 1e-3
        else:
            clear = np.ones(n0, bool)
        rm0, rs0 = ref["matches0"][0], ref["matching_scores0"][0]
        near_th = (np.abs(rs0 - th) < 1e-4) & (rs0 > 0)
        if n0 > 1:
            top2c = np.sort(inner, axis=0)[-2:, :]
            clear &= ((top2c[1] - top2c[0]) > 1e-3)[inner.argmax(1)]
        ok = clear & ~near_th
        assert np.array_equal(m0[b, :n0][ok], rm0[ok])
        assert np.max(np.abs(s0[b, :n0][ok] - rs0[ok]), initial=0.0) < 1e-4
        compared += int(ok.sum())
    assert compared > 0.5 * sum(min(c) and c[0] for c in counts)


def test_match_topk_pairs_caps_the_match_list_on_the_device():
    """kp2d_match_topk_pairs: the VO loop's top_k_matches cap (visual_odometry.py:272-283: np.argpartition(distance, k)[:k] on
    the brute-force matches; :26-32 + :260-266: get_matches_scores then scores.topk(k) on LightGlue's) against numpy on the
    same match tables: the same SET of pairs (the reference's order inside the set is unspecified), best first here, the
    coordinates gathered, k <= 0 / k > matches = every match."""
    from nano_vs_slam_amd.matching import match_topk_pairs
    rng = np.random.default_rng(9)
    B, n0, n1 = 3, 700, 650
    pts0 = rng.random((B, n0, 2)).astype(np.float32) * 300
    pts1 = rng.random((B, n1, 2)).astype(np.float32) * 300
    t = lambda a: torch.from_numpy(a).to(DEV)
    # brute force: match_q[t] = query kept for train row t
    mq = np.full((B, n1), -1, np.int32)
    md = rng.random((B, n1)).astype(np.float32)
    for b in range(B):
        rows = rng.permutation(n1)[:[400, 30, 0][b]]
        mq[b, rows] = rng.permutation(n0)[:len(rows)]
    md[0, :8] = 0.25                                            # equal distances: lower source row first
    match = {"match_q": t(mq), "match_d": t(md), "nn_idx": torch.empty(B, n0, dtype=torch.int32, device=DEV)}
    for k in (100, 0, 5000):
        r = match_topk_pairs(k, t(pts0), t(pts1), match=match)
        cnt = r["count"].cpu().numpy()
        for b in range(B):
            rows = np.nonzero(mq[b] >= 0)[0]
            keep = rows if (k <= 0 or len(rows) <= k) else rows[np.argsort(md[b, rows], kind="stable")[:k]]
            assert cnt[b] == len(keep)
            idx = r["idx"][b, :cnt[b]].cpu().numpy()
            assert set(map(tuple, idx.tolist())) == {(int(mq[b, tt]), int(tt)) for tt in keep}
            val = r["val"][b, :cnt[b]].cpu().numpy()
            assert np.all(np.diff(val) >= 0) and np.array_equal(val, md[b, idx[:, 1]])
            pr = r["pairs"][b, :cnt[b]].cpu().numpy()
            assert np.array_equal(pr[:, :2], pts0[b, idx[:, 0]]) and np.array_equal(pr[:, 2:], pts1[b, idx[:, 1]])
            assert np.all(r["idx"][b, cnt[b]:].cpu().numpy() == -1)
    # LightGlue: matches0[q] = train row matched to query q, matching_scores0[q]
    m0 = np.full((B, n0), -1, np.int64)
    sc = rng.random((B, n0)).astype(np.float32)
    for b in range(B):
        rows = rng.permutation(n0)[:[500, 12, 1][b]]
        m0[b, rows] = rng.permutation(n1)[:len(rows)]
    for k in (64, 0):
        r = match_topk_pairs(k, t(pts0), t(pts1), matches0=t(m0), scores0=t(sc))
        cnt = r["count"].cpu().numpy()
        for b in range(B):
            rows = np.nonzero(m0[b] > -1)[0]                    # get_matches_scores: m0 = matches0 > -1
            keep = rows if (k <= 0 or len(rows) <= k) else rows[np.argsort(-sc[b, rows], kind="stable")[:k]]
            assert cnt[b] == len(keep)
            idx = r["idx"][b, :cnt[b]].cpu().numpy()
            assert set(map(tuple, idx.tolist())) == {(int(q), int(m0[b, q])) for q in keep}
            val = r["val"][b, :cnt[b]].cpu().numpy()
            assert np.all(np.diff(val) <= 0) and np.array_equal(val, sc[b, idx[:, 0]])
            pr = r["pairs"][b, :cnt[b]].cpu().numpy()
            assert np.array_equal(pr[:, :2], pts0[b, idx[:, 0]]) and np.array_equal(pr[:, 2:], pts1[b, idx[:, 1]])
