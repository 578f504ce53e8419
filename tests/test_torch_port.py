"""bench.py's CPU baseline (oracle/torch_port.py) computes the same function as the numpy oracle."""
import numpy as np
import pytest
import torch

from oracle import kp2d_oracle as orc
from oracle import torch_port as tp
from oracle.weights import spread_state_dict, synthetic_frames


@pytest.mark.parametrize("config,v3,ncls", [("S", False, 28), ("N", False, 28), ("S_A", True, 19)])
def test_port_matches_oracle(config, v3, ncls):
    cfg = orc.get_config(config, v3)
    sd = spread_state_dict(orc.state_dict_shapes(cfg, ncls))
    x = synthetic_frames(2, 48, 64, seed=5)
    ref = orc.forward(x, sd, cfg)
    refp = orc.post_processing(ref, 48, 64, cfg)
    with torch.no_grad():
        out = tp.forward(torch.from_numpy(x), tp.to_torch(sd), cfg)
        post = tp.post_processing(out, 48, 64, cfg)
    for k in ("score", "coord", "feat", "vlad", "seg"):
        assert np.max(np.abs(out[k].numpy() - ref[k])) < 1e-4, k
    for k in ("score", "coord", "feat"):
        assert np.max(np.abs(post[k].numpy() - refp[k])) < 2e-4, k
    assert (post["seg"].numpy() != refp["seg"]).mean() < 1e-3
    sel = tp.select(post, 0.7, 50)
    for b in range(2):
        idx = orc.select_k1(refp["score"][b:b + 1], refp["coord"][b:b + 1], refp["feat"][b:b + 1], 0.7, 50)[0]
        assert sel[b][0].shape == (len(idx), 2)
