"""Time the RoI cropper / target builder (N3) at bs=64: kernels only (prepared task table) and end to end (host prepare +
launches).  Usage: python tools/roibench.py [B]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdrnet_amd import roi_data, synth  # noqa: E402
from gdrnet_amd.cfg import lm13_cfg  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
DEV = "cuda:0"
d = synth.make_roi_frames(B, seed=9)
frames = [torch.from_numpy(f).to(DEV) for f in d["frames"]]
rois = []
for r in d["rois"]:
    q = dict(r)
    q.update(image=frames[r["frame"]], xyz_crop=torch.from_numpy(r["xyz_crop"]).to(DEV), segmentation=torch.from_numpy(r["segmentation"]).to(DEV),
             mask_trunc=None if r["mask_trunc"] is None else torch.from_numpy(r["mask_trunc"]).to(DEV))
    rois.append(q)
crop = roi_data.RoiCropper(lm13_cfg(device=DEV), extents=d["extents"], fps_points=d["fps_points"], device=DEV)
for train in (False, True):
    prep = crop.prepare(rois, train)
    for _ in range(5):
        out = crop.launch(prep)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n):
        out = crop.launch(prep)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    out_bytes = sum(v.numel() * v.element_size() for k, v in out.items() if k.startswith("roi_") and v.dim() >= 3)
    src_bytes = sum(min(r["scale"], 720) ** 2 * (3 + 8) for r in d["rois"]) + (sum(r["xyz_crop"].size * 4 + r["scale"] ** 2 for r in d["rois"]) if train else 0)
    t0 = time.perf_counter()
    for _ in range(10):
        crop(rois, train)
    torch.cuda.synchronize()
    e2e = (time.perf_counter() - t0) / 10 * 1e3
    print(f"train={train}: launches {ms*1e3:.1f} us/batch of {B} ({B/ms*1e3:.0f} RoI/s), ~{(out_bytes+src_bytes)/ms/1e6:.0f} GB/s algorithmic "
          f"({out_bytes/1e6:.1f} MB out + {src_bytes/1e6:.1f} MB in); with host prepare {e2e:.2f} ms/batch ({B/e2e*1e3:.0f} RoI/s)")
