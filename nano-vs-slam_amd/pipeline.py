"""Frame -> keypoints front-end, the host mirror of the reference's per-frame ``inference()``
(src/evaluation/visual_odometry.py:74-122) and ``KP2DtinyFrontend.run`` (src/visual_odometry/frontend.py:78-129).

Differences, all on purpose:
  * the image is converted and normalised on the device (``/255``, ``*2-1``), forward + post_processing +
    threshold/top-k selection all run as HIP kernels; only the selected rows are copied to the host
    (the reference copies every cell's score/coord/descriptor and selects with numpy);
  * whole batches are accepted (the reference's ``.view(3, -1)`` idiom assumes B == 1);
  * the uint8 frame is resized (bilinear, align_corners=False, as kornia's resize) and normalised by one HIP kernel
    (kp2d_preprocess); keypoints are scaled back to the original frame as the reference does.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from .selectors import select_keypoints, select_keypoints_host


def _as_frames(frames) -> torch.Tensor:
    t = torch.as_tensor(np.asarray(frames) if not torch.is_tensor(frames) else frames)
    if t.dim() == 3:
        t = t.unsqueeze(0)
    if t.dtype != torch.uint8 or t.shape[-1] != 3:
        raise ValueError("expected uint8 frames of shape [H,W,3] or [B,H,W,3]")
    return t


def _pinned_zero_copy(t: torch.Tensor) -> bool:
    """Pinned host frames are read in place by the preprocess kernel (device-visible host memory: one pass over PCIe,
    at the kernel's own rate) instead of an asynchronous copy first: the hipMemcpyAsync of a 14.7 MB pinned batch runs on
    an SDMA engine and measured SLOWER than torch's staged copy of pageable memory (tools/bench_frontend.py)."""
    return (t.device.type == "cpu" and t.is_pinned() and t.is_contiguous()
            and os.environ.get("KP2D_PINNED_ZERO_COPY", "1") != "0")


def _frames_on_device(frames, device) -> torch.Tensor:
    t = _as_frames(frames)
    t = t.to(device, non_blocking=True).contiguous()
    if t.device.type != "cuda":
        raise RuntimeError("the frame front-end runs on the HIP device only")
    return t


def _fused_front(net) -> bool:
    """kp2d_forward_frames covers RGB models whose first layer has 16 channels (every S / N / F config)."""
    return (os.environ.get("KP2D_FUSED_FRONT", "1") != "0" and getattr(net, "use_color", True)
            and hasattr(net, "forward_frames") and int(net.channel_dims[0]) == 16)


def frames_to_input(frames, device, size=None) -> torch.Tensor:
    """uint8 [Hs,Ws,3] / [B,Hs,Ws,3] (numpy or torch) -> float32 [B,3,H,W] in [-1,1] on ``device``;
    ``size`` = (H, W) resizes (visual_odometry.py:77-87: /255, kornia resize, .sub(0.5).mul(2))."""
    t = _as_frames(frames)
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("the frame front-end runs on the HIP device only")
    if not _pinned_zero_copy(t):
        t = _frames_on_device(t, dev)
        dev = t.device
    elif dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    B, Hs, Ws, _ = t.shape
    H, W = (Hs, Ws) if size is None else (int(size[0]), int(size[1]))
    x = torch.empty(B, 3, H, W, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(_lib.load().kp2d_preprocess(C.c_void_p(t.data_ptr()), B, Hs, Ws, C.c_void_p(x.data_ptr()), H, W,
                                           C.c_void_p(stream)))
    return x


@torch.no_grad()
def inference(net, image, new_size=None, nn_thresh=0.7, top_k=4000, device="cuda"):
    """Returns (pts, feat, out) like the reference: for a single frame ``pts`` [n,2] and ``feat`` [n,C] numpy
    arrays; for a batch, lists of them.  ``out`` is the post-processed dict (device tensors)."""
    src_hw = tuple(image.shape[-3:-1])
    if _fused_front(net) and not (torch.is_tensor(image) and _pinned_zero_copy(image)):
        # uint8 frames straight into the first layer (kp2d_forward_frames): no float frame in between
        t = _frames_on_device(image, device)
        H, W = (t.shape[1], t.shape[2]) if new_size is None else (int(new_size[0]), int(new_size[1]))
        out = net.forward_frames(t, (H, W))
    else:
        x = frames_to_input(image, device, new_size)
        _, _, H, W = x.shape
        out = net(x)
    out = net.post_processing(out, H, W)
    scale = None
    if new_size is not None and (H, W) != src_hw:     # pts / scale: visual_odometry.py:81-83, 119-121
        scale = (W / float(src_hw[1]), H / float(src_hw[0]))
    sel = select_keypoints_host(out, nn_thresh, top_k, scale)
    pts = [p for p, _ in sel]
    feat = [d for _, d in sel]
    single = (not torch.is_tensor(image) and np.asarray(image).ndim == 3) or (torch.is_tensor(image) and image.dim() == 3)
    if single:
        return pts[0], feat[0], out
    return pts, feat, out


class FrameStream:
    """The per-frame VO front-end (``inference()`` one frame at a time, src/evaluation/visual_odometry.py:409-495 calls
    it in a loop) as a REPLAYED HIP GRAPH with overlapped transfers.

    One frame's step is ~60 dependent launches of a few microseconds each: enqueued one by one the host is the
    bottleneck (0.59 ms per frame).  Here the whole step — kp2d_preprocess, forward, post_processing, threshold/top-k
    selection, gather and the copies of the selected rows to pinned host memory — is captured once per slot into a HIP
    graph over static buffers and replayed with one call; frames go through ``slots`` pinned staging buffers (default 7;
    measured on one MI355X box, frames/s by slot count: 2: 5.1k, 3: 7.1k, 4: 5.9k, 5: 7.5k, 6: 8.5k, 7: 9.4k, 8: 7.8k, 10: 9.0k,
    12: 8.4k — multiples of the runtime's four hardware queues pair streams badly; a frame's latency is about ``slots``
    periods, 0.75 ms at 7 — pass ``slots=3`` for 0.42 ms at 7.1k frames/s)
    that the preprocess kernel reads in place, so the host's staging of frame n+1 and the caller's work on frame n-1's
    keypoints run while frame n computes.  Every slot has its OWN compute stream and its OWN engine workspace: one
    frame's ~60 small launches fill a tenth of the chip, so the graphs of consecutive frames run side by side on the
    GPU (on one shared stream they ran back to back and the stream's rate was one frame's latency).
    Same results as ``inference()``, bit for bit (same kernels, same order within a frame).

        fs = FrameStream(net, (Hs, Ws), new_size=(240, 320))
        for pts, feat, out in fs.map(frames): ...          # or fs.submit(frame) ... fs.result()

    ``out`` holds the slot's static device tensors: valid until ``slots`` further frames have been submitted (in ``map``:
    until the next iteration).  Work enqueued on the caller's current stream before that submit still sees them intact:
    ``submit`` orders the slot's replay behind the caller's stream.

    ``match=True`` puts the VO loop's matcher on the device too (``VisualOdometry.match``,
    src/visual_odometry/visual_odometry.py:193-284: ``self.matcher.match(self.prev_descriptors, feat_cur)`` = BF k-NN(2)
    + ratio test + one-to-one, feature_matcher.py:89-98,179-209; ``semantic=True``: per class, :347-380).  Every slot
    keeps its frame's selected keypoints and descriptors (and classes) in HBM; a second replayed graph per slot matches
    the PREVIOUS frame's rows (query) against this frame's (train) — it waits for the previous slot's extraction only,
    so the frames' extractions still run side by side — and compacts the matched coordinates.  ``result()`` then
    returns ``(kps0, kps1, dist, out)``: matched keypoints of the previous and of this frame ([m,2] each, the arrays the
    reference hands to its pose estimation) and the match distances; descriptors never leave the device (8 + 4 bytes
    per match over PCIe instead of (8 + 4 C) bytes per keypoint).  The first frame has no predecessor: empty arrays.

    ``match="lightglue", matcher=LightGlue(...)``: the loop's ``use_lg`` branch (visual_odometry.py:198-266, kp2dtiny
    method): keypoints divided by (W, H) of the network input, ``image_size`` = (H, W) as the reference passes it,
    ``self.lg(data)`` on the previous and the current frame's rows — padded to the selection's capacity, the counts
    handed over as ``num_keypoints0/1`` (kp2d_lg_forward_counts) — then ``get_matches_scores`` (:26-32) on the device;
    ``result()`` returns ``(kps0, kps1, scores, out)``.

    ``top_k_matches=k`` (> 0): the loop's cap — the k smallest distances (:272-283) or, with LightGlue, the k largest
    matching scores (:260-266) — applied on the device (kp2d_match_topk_pairs), best first; the D2H payload is the capped
    list.
    """

    def __init__(self, net, frame_hw, new_size=None, nn_thresh=0.7, top_k=4000, device="cuda", slots=7, match=False,
                 ratio_test=0.7, semantic=False, matcher=None, top_k_matches=0):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("the frame front-end runs on the HIP device only")
        self.dev = torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())
        self.net, self.thr = net, float(nn_thresh)
        Hs, Ws = int(frame_hw[0]), int(frame_hw[1])
        self.H, self.W = (Hs, Ws) if new_size is None else (int(new_size[0]), int(new_size[1]))
        self.scale = None
        if (self.H, self.W) != (Hs, Ws):           # pts / scale: visual_odometry.py:81-83, 119-121
            self.scale = torch.tensor([self.W / float(Ws), self.H / float(Hs)], device=self.dev)
        self.top_k = int(top_k)
        self.slots = int(slots)
        self.lg = None
        if isinstance(match, str):
            if match != "lightglue":
                raise ValueError("match is True (brute force), False, or 'lightglue'")
            if matcher is None:
                raise ValueError("match='lightglue' needs matcher=LightGlue(...)")
            if semantic:
                raise ValueError("the LightGlue branch of the loop has no semantic variant (visual_odometry.py:194-198)")
            self.lg = matcher
        elif matcher is not None:
            raise ValueError("matcher= goes with match='lightglue'")
        self.top_k_matches = int(top_k_matches)
        self._lg_ws = [None] * int(slots)
        self._lg_wh = torch.tensor([float(self.W), float(self.H)], device=self.dev)
        self._lg_size = torch.tensor([float(self.H), float(self.W)], device=self.dev)
        self.match, self.ratio, self.semantic = bool(match), float(ratio_test), bool(semantic)
        if self.semantic and not self.match:
            raise ValueError("semantic=True is a mode of match=True")
        if self.match and self.slots < 2:
            raise ValueError("match=True needs at least two slots (the previous frame's rows live in the previous slot)")
        if self.semantic and not getattr(net, "sample_segmentation", False):
            raise ValueError("semantic matching needs the class at every cell: set net.sample_segmentation = True "
                             "(as KP2DtinyFrontend does for its semantic filter)")
        self.copy_stream = torch.cuda.Stream(self.dev)
        # one compute stream per slot (KP2D_FS_SHARED_STREAM=1: the single shared stream of the first version, for A/B)
        shared = os.environ.get("KP2D_FS_SHARED_STREAM", "0") == "1"
        fs_prio = int(os.environ.get("KP2D_FS_PRIORITY", "0"))      # (A/B knob: -1 = high-priority slot streams)
        one = torch.cuda.Stream(self.dev, priority=fs_prio)
        self.compute_streams = [one if shared else (one if s == 0 else torch.cuda.Stream(self.dev, priority=fs_prio)) for s in range(self.slots)]
        self.compute_stream = self.compute_streams[0]
        self._slot_ws = [None] * self.slots        # per-slot engine workspace (the engine caches ONE otherwise)
        self.pin_in = [torch.empty(1, Hs, Ws, 3, dtype=torch.uint8).pin_memory() for _ in range(self.slots)]
        self._pin_np = [p.numpy() for p in self.pin_in]
        # the preprocess kernel reads the frame straight out of the pinned slot (device-visible host memory: 230 KB over
        # PCIe inside the graph) instead of a separate H2D copy + event + cross-stream wait per frame
        self.zero_copy = os.environ.get("KP2D_FS_ZERO_COPY", "1") != "0"
        self.dev_in = (self.pin_in if self.zero_copy else
                       [torch.empty(1, Hs, Ws, 3, dtype=torch.uint8, device=self.dev) for _ in range(self.slots)])
        self.ev_up = [torch.cuda.Event() for _ in range(self.slots)]
        self.ev_done = [torch.cuda.Event() for _ in range(self.slots)]
        self.ev_extract = [torch.cuda.Event() for _ in range(self.slots)]      # slot's keypoint rows are in HBM
        self.ev_match = [None] * self.slots                                     # slot's match graph is done (reads the previous slot)
        self._prev_slot = None        # slot of the frame submitted last (the query side of the next match)
        self._has_match = {}          # slot -> its frame had a predecessor
        self.match_graphs, self.rows, self.mhost = [], [], []
        self.graphs, self.outs, self.host = [], [], []
        self._pending = []            # slots in flight, oldest first
        self._next = 0
        self._sig = None
        # Replayed graphs never use the engine's side stream (kp2d_api.cpp: NetVLAD beside the segmentation head, plain
        # single-frame forwards only), but a stream that merely EXISTS takes one of the runtime's four hardware queues away
        # from the slots' streams (10.3k -> 7.4k frames/s, profiles/r5_ab_side_stream.txt): switched off for this model
        eng = self.net._get_engine(self.dev)
        _lib.check(eng.lib.kp2d_set_option(eng.handle, b"side_overlap", 0))
        self._capture()

    def _step(self, slot):
        """The step on static buffers: everything here is enqueue-only (capturable)."""
        lib = _lib.load()
        # the forward takes its workspace from the engine's cache: give it this slot's own buffer, so that two slots'
        # graphs never share scratch memory
        eng = self.net._get_engine(self.dev)
        if self._slot_ws[slot] is None:
            need = eng.lib.kp2d_workspace_bytes(eng.handle, 1, self.H, self.W)
            self._slot_ws[slot] = torch.empty(max(int(need), 256), dtype=torch.uint8, device=self.dev)
        with eng.using_workspace(self._slot_ws[slot]):
            if _fused_front(self.net) and not self.zero_copy:
                # frame in device memory: the first layer reads it (kp2d_forward_frames)
                fwd = self.net.forward_frames(self.dev_in[slot], (self.H, self.W))
            else:
                # zero-copy slot (pinned host memory): ONE pass over PCIe by the preprocess kernel; the fused first layer
                # would fetch every halo / bilinear tap across the bus again
                x = torch.empty(1, 3, self.H, self.W, device=self.dev)
                stream = torch.cuda.current_stream(self.dev).cuda_stream
                Hs, Ws = self.dev_in[slot].shape[1:3]
                _lib.check(lib.kp2d_preprocess(C.c_void_p(self.dev_in[slot].data_ptr()), 1, Hs, Ws, C.c_void_p(x.data_ptr()),
                                               self.H, self.W, C.c_void_p(stream)))
                fwd = self.net(x)
        out = self.net.post_processing(fwd, self.H, self.W)
        from .selectors import _cap, select_and_gather
        idx, _val, cnt, pts, dsel = select_and_gather(out["score"], out["coord"], out["feat"], _cap(self.top_k, out["score"]),
                                                      self.thr)
        if self.scale is not None:
            pts = pts / self.scale
        cls = None
        if self.semantic:        # class of every selected cell (frontend.py:112-125: seg[mask][top_k])
            cls = torch.gather(out["seg"].reshape(1, -1), 1, idx.clamp(min=0).long()).to(torch.int32)
        return out, pts, dsel, cnt, cls

    def _match_step(self, prev, cur, mo, po):
        """Match slot `prev`'s rows (query) against slot `cur`'s (train) into static buffers: enqueue-only."""
        from .matching import match_descriptors, match_pairs, match_topk_pairs
        rp, rc = self.rows[prev], self.rows[cur]
        if self.lg is not None:
            # visual_odometry.py:237-258: keypoints / (W, H) of the network input; image_size = image_tensor.shape[1:] = (H, W)
            scale = self.scale if self.scale is not None else 1.0      # rows are in the ORIGINAL frame's pixels
            wh, size = self._lg_wh, self._lg_size                       # (device constants: no host copy inside a capture)
            data = {"keypoints0": rp["pts"] * scale / wh, "keypoints1": rc["pts"] * scale / wh,
                    "descriptors0": rp["desc"], "descriptors1": rc["desc"],
                    "view0": {"image_size": size}, "view1": {"image_size": size},
                    "num_keypoints0": rp["cnt"], "num_keypoints1": rc["cnt"]}
            k = rp["pts"].shape[1]
            need = self.lg.workspace_bytes(1, k, k, self.dev)
            if self._lg_ws[cur] is None or self._lg_ws[cur].numel() < need:
                self._lg_ws[cur] = torch.empty(max(need, 256), dtype=torch.uint8, device=self.dev)
            with self.lg.using_workspace(self._lg_ws[cur]):
                mo = self.lg(data)
            po = match_topk_pairs(self.top_k_matches, rp["pts"], rc["pts"], matches0=mo["matches0"],
                                  scores0=mo["matching_scores0"], out=po)
            return mo, po
        mo = match_descriptors(rp["desc"], rp["cnt"], rc["desc"], rc["cnt"], self.ratio, cls0=rp.get("cls"),
                               cls1=rc.get("cls"), out=mo)
        if self.top_k_matches > 0:
            po = match_topk_pairs(self.top_k_matches, rp["pts"], rc["pts"], match=mo, out=po)
        else:
            po = match_pairs(mo, rp["pts"], rc["pts"], out=po)
        return mo, po

    @torch.no_grad()
    def _capture(self):
        self._precision = self.net.__dict__.get("_precision")
        with torch.cuda.stream(self.compute_stream):
            for _ in range(2):                      # warm-up outside capture: engine handle, workspace, lane objects
                out, pts, dsel, cnt, cls = self._step(0)
        self.compute_stream.synchronize()
        self.graphs, self.outs, self.host = [], [], []
        self.match_graphs, self.rows, self.mhost = [], [], []
        for s in range(self.slots):
            k, cdim = pts.shape[1], dsel.shape[2]
            host = None
            if not self.match:
                host = (torch.empty(1, dtype=torch.int32).pin_memory(), torch.empty(1, k, 2).pin_memory(),
                        torch.empty(1, k, cdim).pin_memory())
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=self.compute_streams[s]):
                out, pts, dsel, cnt, cls = self._step(s)
                if host is not None:
                    host[0].copy_(cnt, non_blocking=True)
                    host[1].copy_(pts, non_blocking=True)
                    host[2].copy_(dsel, non_blocking=True)
            self.graphs.append(g)
            self.outs.append(out)
            self.host.append(host)
            row = {"pts": pts, "desc": dsel, "cnt": cnt}
            if cls is not None:
                row["cls"] = cls
            self.rows.append(row)
        if self.match:
            for row in self.rows:         # (captured, never run: the row counts are uninitialised memory until the first replay)
                row["cnt"].zero_()
            torch.cuda.synchronize(self.dev)
            for s in range(self.slots):
                prev = (s - 1) % self.slots
                cs = self.compute_streams[s]
                with torch.cuda.stream(cs):
                    mo, po = self._match_step(prev, s, None, None)      # allocates the static outputs (outside capture)
                cs.synchronize()
                k = po["pairs"].shape[1]                 # the selection's capacity, or the top_k_matches cap
                vkey = "val" if "val" in po else "dist"
                mh = (torch.empty(1, dtype=torch.int32).pin_memory(), torch.empty(1, k, 4).pin_memory(),
                      torch.empty(1, k).pin_memory())
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=cs):
                    mo2, _ = self._match_step(prev, s, None if self.lg is not None else mo, po)
                    mh[0].copy_(po["count"], non_blocking=True)
                    mh[1].copy_(po["pairs"], non_blocking=True)
                    mh[2].copy_(po[vkey], non_blocking=True)
                if self.lg is not None:
                    mo = mo2                                 # (LightGlue's outputs are the tensors of the captured call)
                self.match_graphs.append(g)
                self.mhost.append(mh)
                self.outs[s] = dict(self.outs[s], match=mo, pairs=po, rows=self.rows[s])
        self._prev_slot = None
        self.ev_match = [None] * self.slots
        self._sig = self.net._weights_signature()

    def submit(self, frame):
        """Queue one uint8 [Hs,Ws,3] frame (numpy or torch, host memory).  At most ``slots`` frames may be in flight."""
        if len(self._pending) >= self.slots:
            raise RuntimeError("FrameStream: every slot is in flight; call result() first")
        if self.net.__dict__.get("_precision") != self._precision:
            self._drain()
            self._capture()                         # another arithmetic mode launches other kernels
        elif self.net._weights_signature() != self._sig:
            self._drain()
            self.net._get_engine(self.dev)          # re-upload in place: the packed blob keeps its address, graphs stay valid
            self._sig = self.net._weights_signature()
        s = self._next
        self._next = (s + 1) % self.slots
        f = frame.numpy() if torch.is_tensor(frame) else np.asarray(frame)
        if f.dtype != np.uint8 or tuple(f.shape) != tuple(self.pin_in[s].shape[1:]):
            raise ValueError(f"expected a uint8 frame of shape {tuple(self.pin_in[s].shape[1:])}")
        # plain memcpy into the pinned slot.  NOT torch's copy_: for a 230 KB tensor it fans out over the intra-op
        # thread pool, and on a box whose cgroup grants fewer CPUs than it shows that burns the CPU quota — measured
        # as a 90 ms stall every ~20 frames (mean 4.6 ms per frame against a 0.42 ms median)
        np.copyto(self._pin_np[s][0], f)
        if not self.zero_copy:
            with torch.cuda.stream(self.copy_stream):
                self.dev_in[s].copy_(self.pin_in[s], non_blocking=True)
                self.ev_up[s].record(self.copy_stream)
        cs = self.compute_streams[s]
        # The replay overwrites this slot's static outputs.  Whatever the caller has enqueued on ITS stream to read the
        # tensors of the frame that last used the slot (a clone, a matcher) must run first: order the slot's stream
        # behind the caller's.  (Without it the reuse raced with such reads — seen once in four runs of the test suite
        # with seven slots in flight.)
        cs.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(cs):
            if not self.zero_copy:
                cs.wait_event(self.ev_up[s])
            if self.match:
                # this slot's rows are the QUERY side of the next slot's match of the previous round: it must be done
                nxt = (s + 1) % self.slots
                if self.ev_match[nxt] is not None:
                    cs.wait_event(self.ev_match[nxt])
            self.graphs[s].replay()
            if self.match:
                self.ev_extract[s].record(cs)
                have_prev = self._prev_slot is not None
                if have_prev:
                    cs.wait_event(self.ev_extract[self._prev_slot])      # only the match waits for the previous frame
                    self.match_graphs[s].replay()
                    if self.ev_match[s] is None:
                        self.ev_match[s] = torch.cuda.Event()
                    self.ev_match[s].record(cs)
                self._has_match[s] = have_prev
                self._prev_slot = s
            self.ev_done[s].record(cs)
        self._pending.append(s)

    def result(self):
        """(pts [n,2], feat [n,C], out) of the oldest frame in flight — numpy copies, like ``inference()``; with
        ``match=True``: (kps0 [m,2], kps1 [m,2], dist [m], out) — the matched keypoints of the previous and of this frame."""
        if not self._pending:
            raise RuntimeError("FrameStream: nothing in flight")
        s = self._pending.pop(0)
        self.ev_done[s].synchronize()
        if self.match:
            if not self._has_match.get(s, False):
                return np.zeros((0, 2), np.float32), np.zeros((0, 2), np.float32), np.zeros((0,), np.float32), self.outs[s]
            m = int(self.mhost[s][0][0])
            pr = self.mhost[s][1][0, :m].numpy()
            return pr[:, :2].copy(), pr[:, 2:].copy(), self.mhost[s][2][0, :m].numpy().copy(), self.outs[s]
        n = int(self.host[s][0][0])
        return self.host[s][1][0, :n].numpy().copy(), self.host[s][2][0, :n].numpy().copy(), self.outs[s]

    def _drain(self):
        for s in self._pending:
            self.ev_done[s].synchronize()
        self._pending = []
        self._prev_slot = None        # a re-capture or weight upload breaks the chain: the next frame has no predecessor

    def map(self, frames):
        """Run an iterable of frames through the stream, one frame of look-ahead; yields in order."""
        for f in frames:
            if len(self._pending) >= self.slots:
                yield self.result()
            self.submit(f)
        while self._pending:
            yield self.result()


class BatchStream:
    """Whole batches through the network with ``slots`` batches in flight (throughput path; the batched counterpart of
    ``FrameStream``).  The reference evaluates batches one after the other (``model(x)`` per batch of the data loader,
    src/evaluation/keypoints.py:102-108, src/evaluation/segmentation.py:38-40); on one stream every layer of batch n + 1 waits for
    the last workgroup of batch n's last kernel, and inside a batch every launch has a tail in which most of the chip
    idles.  Here consecutive batches alternate over ``slots`` HIP streams, each with its own engine workspace, and the
    engine runs each forward as ONE stream lane (``kp2d_set_option("lanes", 1)``): the two batches in flight are at
    different layers at any moment, so one's launch tails and its small post-processing kernels lie under the other's
    big launches.  Two stream lanes INSIDE one forward (the engine's default) run the same layer at the same time and
    cannot do that: 64 x 240 x 320: 22.9k frames/s one batch at a time, 23.4k with two in flight (same box; three or
    four in flight: no further gain).  Results are those of ``net(x)`` + ``post_processing`` + ``select_and_gather``
    on one stream, bit for bit (same kernels; the lane count changes no arithmetic —
    tests/test_gpu_parity.py::test_batch_stream_equals_the_plain_loop).

        bs = BatchStream(net, top_k=1000)
        for out, pts, desc, cnt in bs.map(batches): ...     # or h = bs.submit(x) ... bs.result(h)

    ``submit`` orders the slot's work behind everything already queued on the caller's current stream (so ``x`` may
    come from it); ``result`` makes the caller's current stream wait for the slot (no host synchronisation) and
    returns the step's tensors, which stay valid until the slot is submitted again ``slots`` batches later: work the
    caller enqueues on its current stream before that submit sees them intact (``submit`` / ``submit_frames`` order the
    slot's stream behind the caller's); reads from any other stream need their own event.
    Every slot's forward runs on the slot's OWN workspace (``_Engine.using_workspace``); the engine's cached workspace is
    never touched, so a plain ``net(x)`` / ``inference()`` while batches are in flight is safe (it runs as one lane until
    ``close()``: same results, the lane count changes no arithmetic).  Use it as a context manager, or call ``close()``,
    to give the engine its default lane count back; ``__del__`` does so as a last resort.
    """

    def __init__(self, net, slots=2, top_k=1000, nn_thresh=0.7, device="cuda", select=True):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("BatchStream needs a HIP device (there is no CPU path)")
        self.dev = dev if dev.index is not None else torch.device("cuda", torch.cuda.current_device())
        self.net, self.slots, self.top_k, self.thr, self.select = net, max(1, int(slots)), top_k, nn_thresh, select
        # High-priority streams: the runtime keeps a separate pool of hardware queues per priority, so the slots' streams do
        # not share queues with the process's other (normal-priority) streams — which queue a normal stream lands on depends
        # on what the process created before (64 frames of 120 x 160: 62.4k frames/s or 77k by that accident alone; 76.2k
        # every time this way; 240 x 320 batches: no difference; profiles/r5_hw_queues.txt).  KP2D_BS_PRIORITY=0: normal.
        prio = int(os.environ.get("KP2D_BS_PRIORITY", "-1"))
        self.streams = [torch.cuda.Stream(self.dev, priority=prio) for _ in range(self.slots)]
        self.events = [torch.cuda.Event() for _ in range(self.slots)]
        self._ws = [None] * self.slots
        self._res = [None] * self.slots
        self._pin_out, self._held = {}, {}
        self._n = 0
        self._eng = net._get_engine(self.dev)
        self._open = True
        if self.slots > 1:       # several forwards side by side: each as one lane (see the class comment)
            if getattr(self._eng, "_batch_stream_open", False):
                raise RuntimeError("another BatchStream is open on this model: close() it first")
            self._eng._batch_stream_open = True
            _lib.check(self._eng.lib.kp2d_set_option(self._eng.handle, b"lanes", 1))

    def close(self):
        """Wait for the slots and give the engine its default lane count back."""
        if not self._open:
            return
        self._open = False
        torch.cuda.synchronize(self.dev)
        if self.slots > 1:
            self._eng._batch_stream_open = False
            _lib.check(self._eng.lib.kp2d_set_option(self._eng.handle, b"lanes", 0))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check_open(self):
        if not self._open:
            raise RuntimeError("BatchStream is closed")

    def submit(self, x: torch.Tensor) -> int:
        from .selectors import _cap, select_and_gather
        self._check_open()
        slot = self._n % self.slots
        self._n += 1
        st = self.streams[slot]
        st.wait_stream(torch.cuda.current_stream(self.dev))
        B, _, H, W = x.shape
        need = int(self._eng.lib.kp2d_workspace_bytes(self._eng.handle, B, H, W))
        if self._ws[slot] is None or self._ws[slot].numel() < need:
            self._ws[slot] = torch.empty(max(need, 256), dtype=torch.uint8, device=self.dev)
        with torch.cuda.stream(st), torch.no_grad(), self._eng.using_workspace(self._ws[slot]):
            x.record_stream(st)
            out = self.net.post_processing(self.net(x), H, W)
            if self.select:
                _idx, _val, cnt, pts, desc = select_and_gather(out["score"], out["coord"], out["feat"],
                                                               _cap(self.top_k, out["score"]), self.thr)
                self._res[slot] = (out, pts, desc, cnt)
            else:
                self._res[slot] = (out, None, None, None)
            self.events[slot].record(st)
        return slot

    def result(self, slot: int):
        torch.cuda.current_stream(self.dev).wait_event(self.events[slot])
        return self._res[slot]

    # ---- the same with the frames in HOST memory (the PCIe-inclusive path of ``inference()`` for whole batches) ----
    def submit_frames(self, frames, new_size=None) -> int:
        """uint8 frames [B,Hs,Ws,3] in host memory (numpy or torch) -> a slot.  The slot's stream reads them across
        PCIe — pinned, contiguous frames in place with the preprocess kernel (device-visible memory: one pass over the
        bus at the kernel's own rate, no copy engine; the buffer must stay untouched until ``result_host``), pageable
        ones by torch's staged copy and then straight into the first layer, as ``inference()`` does — then runs forward,
        post_processing and selection, and copies the selected rows (counts, points, descriptors) into the slot's
        pinned output buffers, all behind the other slots' compute.  ``result_host(slot)`` waits for that one slot."""
        from .selectors import _cap, select_and_gather
        t = _as_frames(frames)
        if t.device.type != "cpu":
            raise ValueError("submit_frames takes host frames (device tensors: forward_frames / submit)")
        self._check_open()
        slot = self._n % self.slots
        self._n += 1
        st = self.streams[slot]
        # the slot's previous tensors may still be read by work on the caller's stream (see the class comment)
        st.wait_stream(torch.cuda.current_stream(self.dev))
        B, Hs, Ws, _ = t.shape
        H, W = (Hs, Ws) if new_size is None else (int(new_size[0]), int(new_size[1]))
        zero_copy = _pinned_zero_copy(t)
        self._held[slot] = t if zero_copy else None
        need = int(self._eng.lib.kp2d_workspace_bytes(self._eng.handle, B, H, W))
        if self._ws[slot] is None or self._ws[slot].numel() < need:
            self._ws[slot] = torch.empty(max(need, 256), dtype=torch.uint8, device=self.dev)
        with torch.cuda.stream(st), torch.no_grad(), self._eng.using_workspace(self._ws[slot]):
            if zero_copy or not _fused_front(self.net):
                if not zero_copy:
                    t = t.to(self.dev, non_blocking=True)
                x = torch.empty(B, 3, H, W, device=self.dev)
                _lib.check(_lib.load().kp2d_preprocess(C.c_void_p(t.data_ptr()), B, Hs, Ws, C.c_void_p(x.data_ptr()), H, W,
                                                       C.c_void_p(st.cuda_stream)))
                fwd = self.net(x)
            else:
                # pageable frames: torch's staged upload on this slot's stream, then straight into the first layer
                # (kp2d_forward_frames) — what inference() does, with the other slots' kernels running meanwhile
                fwd = self.net.forward_frames(t.to(self.dev, non_blocking=True), (H, W))
            out = self.net.post_processing(fwd, H, W)
            _idx, _val, cnt, pts, desc = select_and_gather(out["score"], out["coord"], out["feat"],
                                                           _cap(self.top_k, out["score"]), self.thr)
            if new_size is not None and (H, W) != (Hs, Ws):     # pts / scale: visual_odometry.py:81-83, 119-121
                pts = pts / torch.tensor([W / float(Ws), H / float(Hs)], device=pts.device, dtype=pts.dtype)
            ho = self._pin_out.get(slot)
            if ho is None or ho[1].shape != pts.shape or ho[2].shape != desc.shape:
                ho = self._pin_out[slot] = (torch.empty(cnt.shape, dtype=cnt.dtype).pin_memory(),
                                            torch.empty(pts.shape, dtype=pts.dtype).pin_memory(),
                                            torch.empty(desc.shape, dtype=desc.dtype).pin_memory())
            ho[0].copy_(cnt, non_blocking=True)
            ho[1].copy_(pts, non_blocking=True)
            ho[2].copy_(desc, non_blocking=True)
            self._res[slot] = (out, pts, desc, cnt)
            self.events[slot].record(st)
        return slot

    def result_host(self, slot: int):
        """(pts, feat, out) of a ``submit_frames`` slot as ``inference()`` returns them for a batch: per-frame numpy
        arrays [n,2] / [n,C] (views of the slot's pinned buffers: valid until the slot is submitted again) and the
        post-processed dict (device tensors)."""
        self.events[slot].synchronize()
        hc, hp, hd = self._pin_out[slot]
        counts = hc.tolist()
        hp, hd = hp.numpy(), hd.numpy()
        self._held[slot] = None
        return [hp[b, :n] for b, n in enumerate(counts)], [hd[b, :n] for b, n in enumerate(counts)], self._res[slot][0]

    def map_frames(self, batches, new_size=None):
        pending = []
        for f in batches:
            pending.append(self.submit_frames(f, new_size))
            if len(pending) == self.slots:
                yield self.result_host(pending.pop(0))
        while pending:
            yield self.result_host(pending.pop(0))

    def map(self, batches):
        """Yield every batch's (out, pts, desc, cnt) in order, ``slots - 1`` batches behind the submissions."""
        pending = []
        for x in batches:
            pending.append(self.submit(x))
            if len(pending) == self.slots:
                yield self.result(pending.pop(0))
        while pending:
            yield self.result(pending.pop(0))


@torch.no_grad()
def two_view_match(net, matcher, image0: torch.Tensor, image1: torch.Tensor, max_num_keypoints: int = 1024):
    """Extractor + LightGlue on batches of image pairs, everything on the device — the model-facing sequence of
    gluefactory's ``two_view_pipeline`` with the ``kp2dtiny`` extractor and the ``lightglue`` matcher
    (gluefactory/models/extractors/kp2dtiny.py:24-58, gluefactory/configs/kp2dtiny_S+lightglue_homography.yaml).

    image0 / image1: float32 [B,3,H,W] in [0,1] (the extractor applies ``.sub(0.5).mul(2)`` itself, :25), H and W are
    cropped to multiples of 8 (:30-33).  Returns (pred0, pred1, matches) with the extractor's and the matcher's dicts.
    """
    from .selectors import extract_topk
    if image0.shape != image1.shape:
        raise ValueError("two_view_match expects both views at the same size")
    B = image0.shape[0]
    H, W = image0.shape[-2] - image0.shape[-2] % 8, image0.shape[-1] - image0.shape[-1] % 8
    # both views go through the extractor as ONE batch (frames are independent): twice the work per launch
    x = (torch.cat([image0[:, :, :H, :W], image1[:, :, :H, :W]], 0) - 0.5) * 2.0
    out = net.post_processing(net(x.contiguous()), H, W)
    both = extract_topk(out, max_num_keypoints)
    size = torch.tensor([float(W), float(H)], device=x.device).expand(B, 2)
    preds = []
    for v in range(2):
        p = {k: t[v * B:(v + 1) * B] for k, t in both.items()}
        p["image_size"] = size
        preds.append(p)
    data = {"keypoints0": preds[0]["keypoints"], "keypoints1": preds[1]["keypoints"],
            "descriptors0": preds[0]["descriptors"], "descriptors1": preds[1]["descriptors"],
            "view0": {"image_size": preds[0]["image_size"]}, "view1": {"image_size": preds[1]["image_size"]}}
    return preds[0], preds[1], matcher(data)
