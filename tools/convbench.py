"""Micro-benchmarks of the conv kernels on selected layer shapes (bring-up / profiling aid, not product)."""
import os
import sys
import time

import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import hiputil as H  # noqa: E402
from gdrnet_amd.cabi import BF16  # noqa: E402

reps = 5
B = 64


def timeit(fn):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for (C_, Hh) in ((256, 16), (512, 8), (128, 32), (64, 64)):
    x = torch.randn(B, Hh, Hh, C_, device="cuda").to(torch.bfloat16)
    dy = torch.randn(B, Hh, Hh, C_, device="cuda").to(torch.bfloat16)
    fl = 2 * B * Hh * Hh * C_ * C_ * 9
    for sp in (2, 4, 8, 16, 32, 64, 0):
        dt = timeit(lambda: H.conv_wgrad(x, dy, B, Hh, Hh, C_, C_, Hh, Hh, C_, C_, 3, 3, 1, 1, BF16, splits=sp, halo=True))
        print("wgrad C=%d H=%d splits=%2d : %7.1f us %7.1f TF" % (C_, Hh, sp, dt * 1e6, fl / dt / 1e12), flush=True)
