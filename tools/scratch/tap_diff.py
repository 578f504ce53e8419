"""Dev aid: first layer where the two precision modes disagree (kp2d_set_tap), for one config / shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from conftest import product_model
from oracle.weights import synthetic_frames
config, v3, H, W, B = sys.argv[1], sys.argv[2] == "1", int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
model, _ = product_model(config, v3, 28)
x = torch.from_numpy(synthetic_frames(B, H, W, seed=3)).to("cuda:0")
cd = model.channel_dims
ds = model.downsample
layers = [("backbone.conv1a", cd[0], 1), ("backbone.conv1b", cd[1], 2 if ds >= 2 else 1), ("backbone.conv2a", cd[1], 2),
          ("backbone.conv2b", cd[2], 4 if ds >= 3 else 2), ("backbone.conv3a", cd[2], 4 if ds >= 3 else 2),
          ("backbone.conv3b", cd[3], 4 if ds >= 3 else 2), ("backbone.conv4a", cd[3], 8 if ds >= 3 else 4),
          ("backbone.conv4b", cd[3], 8 if ds >= 3 else 4)]
with torch.no_grad():
    for name, c, div in layers:
        outs = {}
        for prec in ("fp32", "f16x3"):
            model.set_precision(prec)
            _, t = model.forward_with_tap(x, name, (c, H // div, W // div))
            outs[prec] = t.cpu().numpy()
        d = np.abs(outs["fp32"] - outs["f16x3"])
        print(f"{name:20s} max diff {np.nanmax(d):.3e}  nan {int(np.isnan(outs['f16x3']).sum())}  worst at {np.unravel_index(np.nanargmax(d), d.shape)}")
    for prec in ("fp32", "f16x3"):
        model.set_precision(prec)
        outs[prec] = {k: v.cpu().numpy() for k, v in model(x).items()}
    for k in outs["fp32"]:
        d = np.abs(outs["fp32"][k] - outs["f16x3"][k])
        print(f"out {k:8s} max diff {d.max():.3e} at {np.unravel_index(d.argmax(), d.shape)}")
