"""Timing aid (not product): halo forward conv at 16x16 / bs=64 / 256 output channels for growing Cin -- separates the fixed
per-launch cost (prologue, epilogue, launch) from the per-chunk MFMA time."""
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import hiputil as H  # noqa: E402
from gdrnet_amd.cabi import BF16  # noqa: E402

B, Hh, O = 64, 16, 256
for C_ in (64, 128, 256, 512):
    x = torch.randn(B, Hh, Hh, C_, device="cuda").to(torch.bfloat16)
    w = (torch.randn(O, 9, C_, device="cuda") * 0.05).to(torch.bfloat16)
    ts = []
    for _ in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        H.conv_gemm(x, w, B, Hh, Hh, C_, C_, Hh, Hh, O, 3, 3, 1, 1, BF16, halo=True, want_stats=True)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    print("Cin=%d" % C_, "us per call incl. pack/alloc:", ["%.0f" % t for t in ts])
