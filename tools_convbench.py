"""Micro-benchmark of the conv kernels on the dominant layer shape (bring-up / profiling aid, not product)."""
import sys
import time

import torch

import os
R = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import hiputil as H  # noqa: E402
from gdrnet_amd.cabi import BF16  # noqa: E402

B, C_, Hh = 64, 256, 64
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
x = torch.randn(B, Hh, Hh, C_, device="cuda").to(torch.bfloat16)
w = torch.randn(C_, C_, 3, 3) / 48
wp = H.pack_fwd(w, BF16)
for halo in (True, False):
    for _ in range(reps):
        y, _ = H.conv_gemm(x, wp, B, Hh, Hh, C_, C_, Hh, Hh, C_, 3, 3, 1, 1, BF16, halo=halo)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        y, _ = H.conv_gemm(x, wp, B, Hh, Hh, C_, C_, Hh, Hh, C_, 3, 3, 1, 1, BF16, halo=halo)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("halo" if halo else "generic", "%.1f us  %.1f TF" % (dt * 1e6, 2 * B * Hh * Hh * C_ * C_ * 9 / dt / 1e12), flush=True)
dy = torch.randn(B, Hh, Hh, C_, device="cuda").to(torch.bfloat16)
for _ in range(reps):
    dw = H.conv_wgrad(x, dy, B, Hh, Hh, C_, C_, Hh, Hh, C_, C_, 3, 3, 1, 1, BF16)
