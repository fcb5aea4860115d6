"""Event timing of the halo forward conv (+ statistics) on four layer shapes of the bs=64 step.  Usage: python tools/halotime.py"""
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
for (C_, Hh) in ((256, 64), (256, 32), (256, 16), (128, 32), (64, 64)):
    x = torch.randn(B, Hh, Hh, C_, device="cuda").to(torch.bfloat16)
    w = (torch.randn(C_, 9, C_, device="cuda") * 0.05).to(torch.bfloat16)
    wf = torch.empty_like(w)
    check(lib.gdrn_pack_wfrag(ptr(w), ptr(wf), C_, C_, BF16, H.stream()), "pack")
    y = torch.empty(B, Hh, Hh, C_, device="cuda", dtype=torch.bfloat16)
    cp = ConvParams()
    cp.x, cp.w, cp.y = ptr(x), ptr(wf), ptr(y)
    cp.Hi = cp.Wi = cp.Ho = cp.Wo = Hh
    cp.Cin = cp.x_cs = cp.Cout = cp.y_cs = C_
    cp.KH = cp.KW = 3
    cp.stride = 1
    cp.pad = 1
    cp.M = B * Hh * Hh
    cp.w_rows = C_
    cp.dtype = BF16
    rows = lib.gdrn_conv3x3_stats_rows(C.byref(cp))
    stats = torch.zeros(rows, 2, C_, device="cuda")
    cp.stats = ptr(stats)
    for _ in range(3):
        check(lib.gdrn_conv3x3_halo(C.byref(cp), H.stream()), "halo")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        lib.gdrn_conv3x3_halo(C.byref(cp), H.stream())
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print(f"C={C_} H={Hh}: {us:7.1f} us {2.0 * B * Hh * Hh * C_ * C_ * 9 / us / 1e6:7.0f} TF")
