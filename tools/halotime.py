"""Event timing of the halo conv on layer shapes of the bs=64 step: plain (xf 0, with statistics) and with the operand transforms
(xf modes 1-4, with / without the xf_out copy) -- what each fused BatchNorm apply costs inside the conv.
Usage: python tools/halotime.py [modes, e.g. 0,1,3]"""
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
modes = [int(m) for m in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,1,2,3,4".split(","))]
WAVES = int(os.environ.get("HALO_WAVES", "0"))   # gdrn_conv_params.halo_waves: 0 library's choice, 4 / 8 forced
for (C_, Hh) in ((256, 64), (256, 32), (256, 16), (512, 8), (128, 32), (64, 64)):
    x = torch.randn(B, Hh, Hh, C_, device="cuda").to(torch.bfloat16)
    x2 = torch.randn(B, Hh, Hh, C_, device="cuda").to(torch.bfloat16)
    xo = torch.empty_like(x)
    w = (torch.randn(C_, 9, C_, device="cuda") * 0.05).to(torch.bfloat16)
    wf = torch.empty_like(w)
    check(lib.gdrn_pack_wfrag(ptr(w), ptr(wf), C_, C_, BF16, H.stream()), "pack")
    y = torch.empty(B, Hh, Hh, C_, device="cuda", dtype=torch.bfloat16)
    vec = [torch.rand(C_, device="cuda") + 0.5 for _ in range(5)]
    line = f"C={C_:3d} H={Hh:2d}:"
    base = None
    for mode in modes:
        for with_out in ((False,) if mode == 0 else (False, True)):
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
            cp.halo_waves = WAVES
            rows = lib.gdrn_conv3x3_stats_rows(C.byref(cp))
            stats = torch.zeros(rows, 2, C_, device="cuda")
            cp.stats = ptr(stats)
            if mode:
                cp.xf_mode, cp.xf_relu = mode, 1 if mode <= 2 else 0
                cp.xf_a, cp.xf_c = ptr(vec[0]), ptr(vec[1])
                if mode >= 2:
                    cp.xf_x2, cp.xf_b = ptr(x2), ptr(vec[2])
                if mode == 4:
                    cp.xf_msc, cp.xf_msh = ptr(vec[3]), ptr(vec[4])
                if with_out:
                    cp.xf_out = ptr(xo)
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
            if base is None:
                base = us
            line += f"  xf{mode}{'+out' if with_out else ''} {us:6.1f}"
    print(line + f"   (plain: {2.0 * B * Hh * Hh * C_ * C_ * 9 / base / 1e6:5.0f} TF)")
