"""PMC / timing aid (not product): the halo forward conv on four layer shapes of the bs=64 step."""
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import hiputil as H  # noqa: E402
from gdrnet_amd.cabi import BF16  # noqa: E402

B = 64
for (C_, Hh) in ((256, 64), (256, 16), (512, 8), (64, 64)):
    x = torch.randn(B, Hh, Hh, C_, device="cuda").to(torch.bfloat16)
    w = (torch.randn(C_, 9, C_, device="cuda") * 0.05).to(torch.bfloat16)
    for _ in range(3):
        H.conv_gemm(x, w, B, Hh, Hh, C_, C_, Hh, Hh, C_, 3, 3, 1, 1, BF16, halo=True, want_stats=True)
print("done")
