"""Host mirror of the reference's VO front-end wrapper ``KP2DtinyFrontend`` (src/visual_odometry/frontend.py:11-129).

Same constructor arguments, ``get_info()`` and ``run(img) -> (pts, feat, seg)`` contract; the network, the
post-processing, the threshold / semantic filter / top-k selection all run on the device (the reference copies every
cell to the host and selects with numpy).  The OpenCV debug windows (``debug=True``) are not reproduced.
"""
from __future__ import annotations

import numpy as np
import torch

from ..kp2dtiny.models.kp2dtiny import tiny_factory
from ..selectors import select_and_gather


class KP2DtinyFrontend(object):
    """Wrapper around the net to help with pre and post image processing."""

    def __init__(self, new_size, weights_path, nn_thresh=0.7, device="cuda", semantic_filter=False,
                 classes_to_filter=[21], debug=True, method="kp2dtiny", config="A", top_k=4000, v3=False, nClasses=28):
        if method != "kp2dtiny":
            raise NotImplementedError(f"method={method!r}: only the kp2dtiny front-end is built")
        self.name = method
        self.device = device
        self.nn_thresh = nn_thresh
        self.border_remove = 4
        self.classes_to_filter = list(classes_to_filter)
        self.apply_semantic_filer = semantic_filter          # (sic) attribute name of the reference
        self.weights_path = weights_path
        self.plot = debug
        self.top_k = top_k
        self.new_size = new_size
        self.net = tiny_factory(config, nClasses, to_export=False, to_mcu=False, v3=v3)
        self.net.sample_segmentation = semantic_filter
        if self.weights_path is not None:
            self.net.load_state_dict(torch.load(weights_path, map_location=torch.device("cpu"), weights_only=True)["state_dict"])
        self.net.eval()
        self.net.training = False
        self.net = self.net.to(self.device)
        self.net.device = self.device

    def get_info(self):
        return {
            "nn_thresh": self.nn_thresh, "border_remove": self.border_remove, "weights_path": self.weights_path,
            "device": self.device, "apply_semantic_filer": self.apply_semantic_filer,
            "classes_to_filter": self.classes_to_filter, "plot": self.plot, "top_k": self.top_k,
            "new_size": self.new_size, "name": self.name, "model": self.net.gather_info(),
        }

    @torch.no_grad()
    def run(self, img):
        """img: float [3,H,W] in [0,1].  Returns (pts [n,2], feat [n,C], seg) numpy arrays like the reference
        (frontend.py:78-129): cells with score > nn_thresh (and, with the semantic filter, a class outside
        ``classes_to_filter``), at most ``top_k`` of them; ``seg`` holds the kept cells' classes with the semantic
        filter, the whole flattened class map without it."""
        img = img.unsqueeze(0).sub(0.5).mul(2.0)
        _, _, H, W = img.shape
        img = img.to(self.device).contiguous()
        out = self.net.post_processing(self.net.forward(img), H, W)
        score, seg = out["score"], out["seg"]
        if self.apply_semantic_filer:
            # sample_segmentation=True: seg is the class at each cell.  Cells of a filtered class are EXCLUDED
            # (frontend.py:112-116 ANDs the masks): -inf never passes "score > nn_thresh", whatever the threshold.
            banned = torch.isin(seg.view(1, -1), torch.as_tensor(self.classes_to_filter, device=seg.device))
            score = torch.where(banned.view_as(score), torch.full_like(score, float("-inf")), score)
        # top_k <= 0 is "no cap" in the reference (frontend.py:122): every cell above the threshold is kept
        k = self.top_k if self.top_k > 0 else score[0].numel()
        idx, _val, cnt, pts, dsel = select_and_gather(score, out["coord"], out["feat"], k, self.nn_thresh)
        n = int(cnt[0])
        sel = idx[0, :n].long()
        seg_out = seg.view(-1)[sel] if self.apply_semantic_filer else seg.view(-1)
        return pts[0, :n].cpu().numpy().copy(), dsel[0, :n].cpu().numpy().copy(), seg_out.cpu().numpy().copy()
