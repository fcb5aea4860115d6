"""What a halo conv launch costs when its weights (and activations) are not cache resident, as inside the training step where every layer
has its own operands: the same launch cycling over NW weight tensors / NX activation buffers (1 = hot, 16 = L2-cold / Infinity-Cache-hot,
256 = HBM-cold).  Usage: python tools/halocold.py"""
import ctypes as C
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import hiputil as H  # noqa: E402
from gdrnet_amd import cabi  # noqa: E402
from gdrnet_amd.cabi import BF16, ConvParams, check, ptr  # noqa: E402

lib = cabi.load()
B = 64
for (C_, Hh) in ((256, 16), (512, 8), (128, 32), (256, 32)):
    line = f"C={C_:3d} H={Hh:2d}:"
    for NW, NX in ((1, 1), (16, 1), (256, 1), (1, 16), (16, 16), (256, 64)):
        NWe = min(NW, max(1, int(600e6 // (C_ * C_ * 18))))
        xs = [torch.randn(B, Hh, Hh, C_, device="cuda").to(torch.bfloat16) for _ in range(NX)]
        y = torch.empty(B, Hh, Hh, C_, device="cuda", dtype=torch.bfloat16)
        w = (torch.randn(C_, 9, C_, device="cuda") * 0.05).to(torch.bfloat16)
        wfs = []
        for _ in range(NWe):
            wf = torch.empty_like(w)
            check(lib.gdrn_pack_wfrag(ptr(w), ptr(wf), C_, C_, BF16, H.stream()), "pack")
            wfs.append(wf)
        cps = []
        n = max(NWe, NX, 32)
        stats = None
        for i in range(n):
            cp = ConvParams()
            cp.x, cp.w, cp.y = ptr(xs[i % NX]), ptr(wfs[i % NWe]), ptr(y)
            cp.Hi = cp.Wi = cp.Ho = cp.Wo = Hh
            cp.Cin = cp.x_cs = cp.Cout = cp.y_cs = C_
            cp.KH = cp.KW = 3
            cp.stride = 1
            cp.pad = 1
            cp.M = B * Hh * Hh
            cp.w_rows = C_
            cp.dtype = BF16
            if stats is None:
                stats = torch.zeros(lib.gdrn_conv3x3_stats_rows(C.byref(cp)), 2, C_, device="cuda")
            cp.stats = ptr(stats)
            cps.append(cp)
        for cp in cps[:3]:
            check(lib.gdrn_conv3x3_halo(C.byref(cp), H.stream()), "halo")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 2 if n > 64 else 4
        e0.record()
        for _ in range(reps):
            for cp in cps:
                lib.gdrn_conv3x3_halo(C.byref(cp), H.stream())
        e1.record()
        torch.cuda.synchronize()
        line += f"  w{NWe}/x{NX} {e0.elapsed_time(e1) / (reps * n) * 1e3:6.1f}"
        del xs, wfs
    print(line, flush=True)
