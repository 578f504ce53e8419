"""LightGlue oracle self-consistency (CPU).  The reference's lightglue.py cannot be imported here (omegaconf is
missing), so the numpy restatement is cross-checked against an independent torch.nn.functional formulation of the same
published architecture — PARITY UNPINNED, see oracle/lightglue_oracle.py."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import lightglue_oracle as lg


def _torch_forward(data, sd, conf):
    P = {k: torch.from_numpy(v).double() for k, v in sd.items()}
    lin = lambda x, n: F.linear(x, P[n + ".weight"], P.get(n + ".bias"))
    H, L, D = conf["num_heads"], conf["n_layers"], conf["descriptor_dim"]

    def norm_kpts(k, size):
        size = torch.as_tensor(size, dtype=k.dtype)
        return (k - size[..., None, :] / 2) / (size.max(-1).values / 2)[..., None, None]

    def rot(t, cs):
        pairs = t.unflatten(-1, (-1, 2))
        swapped = torch.stack((-pairs[..., 1], pairs[..., 0]), -1).flatten(-2)
        return t * cs[0] + swapped * cs[1]

    def ffn(x, n):
        h = lin(x, n + ".0")
        h = F.gelu(F.layer_norm(h, h.shape[-1:], P[n + ".1.weight"], P[n + ".1.bias"]))
        return lin(h, n + ".3")

    def enc(k):
        pr = k @ P["posenc.Wr.weight"].T
        return torch.stack([pr.cos(), pr.sin()], 0).unsqueeze(-3).repeat_interleave(2, dim=-1)

    def selfb(x, e, n):
        qkv = lin(x, n + ".Wqkv").unflatten(-1, (H, -1, 3)).transpose(1, 2)
        q, k, v = rot(qkv[..., 0], e), rot(qkv[..., 1], e), qkv[..., 2]
        ctx = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).flatten(-2)
        return x + ffn(torch.cat([x, lin(ctx, n + ".out_proj")], -1), n + ".ffn")

    def crossb(x0, x1, n):
        heads = lambda t: t.unflatten(-1, (H, -1)).transpose(1, 2)
        a0, a1, v0, v1 = heads(lin(x0, n + ".to_qk")), heads(lin(x1, n + ".to_qk")), heads(lin(x0, n + ".to_v")), heads(lin(x1, n + ".to_v"))
        m0 = F.scaled_dot_product_attention(a0, a1, v1).transpose(1, 2).flatten(-2)
        m1 = F.scaled_dot_product_attention(a1, a0, v0).transpose(1, 2).flatten(-2)
        x0n = x0 + ffn(torch.cat([x0, lin(m0, n + ".to_out")], -1), n + ".ffn")
        x1n = x1 + ffn(torch.cat([x1, lin(m1, n + ".to_out")], -1), n + ".ffn")
        return x0n, x1n

    t = lambda a: torch.from_numpy(np.asarray(a)).double()
    k0 = norm_kpts(t(data["keypoints0"]), data["view0"]["image_size"])
    k1 = norm_kpts(t(data["keypoints1"]), data["view1"]["image_size"])
    d0, d1 = t(data["descriptors0"]), t(data["descriptors1"])
    if conf["input_dim"] != D:
        d0, d1 = lin(d0, "input_proj"), lin(d1, "input_proj")
    e0, e1 = enc(k0), enc(k1)
    for i in range(L):
        d0, d1 = selfb(d0, e0, f"transformers.{i}.self_attn"), selfb(d1, e1, f"transformers.{i}.self_attn")
        d0, d1 = crossb(d0, d1, f"transformers.{i}.cross_attn")
    n = f"log_assignment.{L - 1}"
    f0, f1 = lin(d0, n + ".final_proj") / D ** 0.25, lin(d1, n + ".final_proj") / D ** 0.25
    sim = f0 @ f1.transpose(1, 2)
    z0, z1 = lin(d0, n + ".matchability"), lin(d1, n + ".matchability")
    B, M, N = sim.shape
    sc = sim.new_zeros(B, M + 1, N + 1)
    sc[:, :M, :N] = F.log_softmax(sim, 2) + F.log_softmax(sim, 1) + F.logsigmoid(z0) + F.logsigmoid(z1).transpose(1, 2)
    sc[:, :M, N] = F.logsigmoid(-z0[..., 0])
    sc[:, M, :N] = F.logsigmoid(-z1[..., 0])
    return d0.numpy(), d1.numpy(), sc.numpy()


def make_data(B, M, N, din, seed, size=(320.0, 240.0)):
    g = np.random.default_rng(seed)
    d = lambda n: (lambda x: x / np.linalg.norm(x, axis=-1, keepdims=True))(g.standard_normal((B, n, din))).astype(np.float32)
    k = lambda n: (g.random((B, n, 2)) * np.asarray(size)).astype(np.float32)
    sz = np.broadcast_to(np.asarray(size, np.float32), (B, 2)).copy()
    return {"keypoints0": k(M), "keypoints1": k(N), "descriptors0": d(M), "descriptors1": d(N),
            "view0": {"image_size": sz}, "view1": {"image_size": sz}}


@pytest.mark.parametrize("name,B,M,N", [("S", 2, 70, 53), ("F", 1, 40, 64)])
def test_numpy_oracle_matches_torch_functional(name, B, M, N):
    conf = lg.get_config(name)
    sd = lg.seeded_state_dict(conf)
    data = make_data(B, M, N, conf["input_dim"], seed=3)
    p64 = {k: v.astype(np.float64) for k, v in sd.items()}
    d64 = {k: (v.astype(np.float64) if isinstance(v, np.ndarray) else {"image_size": v["image_size"].astype(np.float64)})
           for k, v in data.items()}
    out = lg.forward(d64, p64, conf)
    t0, t1, tsc = _torch_forward(data, sd, conf)
    assert np.max(np.abs(out["ref_descriptors0"][:, 0] - t0)) < 1e-9
    assert np.max(np.abs(out["ref_descriptors1"][:, 0] - t1)) < 1e-9
    assert np.max(np.abs(out["log_assignment"] - tsc)) < 1e-9
    # fp32 evaluation of the oracle stays close to the fp64 one (sizes the GPU tolerance)
    out32 = lg.forward(data, sd, conf)
    assert np.max(np.abs(out32["log_assignment"] - tsc) / (1.0 + np.abs(tsc))) < 2e-5


def test_state_dict_layout_and_filter_matches():
    conf = lg.get_config("S")
    keys = list(lg.state_dict_shapes(conf))
    assert keys[0] == "posenc.Wr.weight" and "input_proj.weight" not in keys
    assert keys[1] == "transformers.0.self_attn.Wqkv.weight" and keys[-1] == "token_confidence.2.token.0.bias"
    assert len(keys) == 1 + 4 * (4 + 6 + 6 + 6) + 4 * 4 + 3 * 2
    full = lg.state_dict_shapes(lg.get_config({"input_dim": 64, "descriptor_dim": 32, "n_layers": 2}))
    assert list(full)[:2] == ["input_proj.weight", "input_proj.bias"] and full["input_proj.weight"] == (32, 64)
    # filter_matches: mutual nearest neighbours above the threshold, everything else -1 / 0
    sc = np.log(np.array([[[0.6, 0.1, 0.3], [0.2, 0.7, 0.1], [0.5, 0.1, 0.05], [0, 0, 0]]], np.float32) + 1e-9)
    sc = np.concatenate([sc, np.zeros((1, 4, 1), np.float32)], 2)
    m0, m1, s0, s1 = lg.filter_matches(sc, 0.55)
    assert m0.tolist() == [[0, 1, -1]] and m1.tolist() == [[0, 1, -1]]
    assert np.allclose(s0, [[0.6, 0.7, 0.0]], atol=1e-6) and np.allclose(s1, [[0.6, 0.7, 0.0]], atol=1e-6)
    with pytest.raises(ValueError):
        lg.get_config("Z")
