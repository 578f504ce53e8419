import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nano_vs_slam_amd.kp2dtiny.models.kp2dtiny import tiny_factory
from nano_vs_slam_amd.pipeline import FrameStream
from nano_vs_slam_amd.synthetic import spread_state_dict
net = tiny_factory("S", 28)
sd = spread_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()})
net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
net = net.to("cuda:0").eval(); net.training = False
frame = np.random.default_rng(0).integers(0, 256, (240, 320, 3), dtype=np.uint8)
fs = FrameStream(net, (240, 320), None, 0.7, 1000, "cuda:0")
def stats(name, fn, n=400):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ts=[]
    for _ in range(n):
        t0=time.perf_counter(); fn(); ts.append((time.perf_counter()-t0)*1e3)
    ts=np.array(ts); print(f"{name}: mean {ts.mean():.3f} median {np.median(ts):.3f} max {ts.max():.2f} n>2ms {(ts>2).sum()}", flush=True)
def sub_res():
    fs.submit(frame); fs.result()
def pipelined():
    fs.submit(frame)
    if len(fs._pending) == 2: fs.result()
print("zero_copy", fs.zero_copy)
t = torch.from_numpy(frame)
n = 1000
def rep(slot, sync="event"):
    with torch.cuda.stream(fs.compute_stream):
        fs.graphs[slot].replay(); fs.ev_done[slot].record(fs.compute_stream)
    if sync == "event": fs.ev_done[slot].synchronize()
    elif sync == "stream": fs.compute_stream.synchronize()
    else: torch.cuda.synchronize()
stats("E1 replay+sync", lambda: rep(0))
def e2():
    fs.pin_in[0][0].copy_(t); rep(0)
stats("E2 +pin write", e2)
def e3():
    rep(0); return fs.host[0][1][0, :n].numpy().copy(), fs.host[0][2][0, :n].numpy().copy()
stats("E3 +result read", e3)
k=[0]
def e4():
    k[0]^=1; rep(k[0])
stats("E4 alternate graphs", e4)
def e5():
    fs.submit(frame); fs.result()
stats("E5 submit+result", e5)
def e7():
    k[0]^=1; fs.pin_in[k[0]][0].copy_(t); rep(k[0], "stream"); return fs.host[k[0]][1][0, :n].numpy().copy()
stats("E7 stream.synchronize", e7)
def e7b():
    k[0]^=1; fs.pin_in[k[0]][0].copy_(t); rep(k[0], "device"); return fs.host[k[0]][1][0, :n].numpy().copy()
stats("E7b device synchronize", e7b)
npv = [p.numpy() for p in fs.pin_in]
def e8():
    k[0]^=1; npv[k[0]][0][...] = frame; rep(k[0]); return fs.host[k[0]][1][0, :n].numpy().copy()
stats("E8 numpy pin write", e8)
def e9():
    fs._weights_signature = None
    k[0]^=1; rep(k[0]); fs.net._weights_signature()
stats("E9 alternate + signature", e9)
stats("E1 again", lambda: rep(0))
