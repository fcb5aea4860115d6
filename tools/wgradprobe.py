"""probe: halo weight gradient with all patch loads hitting one cached patch (variant 77) vs normal -- is the kernel bound by memory latency?"""
import ctypes as C, os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import hiputil as H
from gdrnet_amd import cabi
from gdrnet_amd.cabi import BF16, WgradParams, check, ptr
lib = cabi.load()
B = 64
for (C_, Hh) in ((256, 64), (256, 32), (256, 16), (128, 32)):
    x = torch.randn(B, Hh, Hh, C_, device="cuda").to(torch.bfloat16)
    dy = torch.randn(B, Hh, Hh, C_, device="cuda").to(torch.bfloat16)
    line = f"C={C_} H={Hh}:"
    for variant in (0, 77):
        for splits in (0, 16 * 1024 // ((C_ // 64) ** 2)):
            wp = WgradParams()
            wp.x, wp.dy = ptr(x), ptr(dy)
            wp.Hi = wp.Wi = wp.Ho = wp.Wo = Hh
            wp.Cin = wp.x_cs = wp.Cout = wp.dy_cs = C_
            wp.KH = wp.KW = 3; wp.stride = 1; wp.pad = 1
            wp.M, wp.dtype, wp.splits, wp.variant = B * Hh * Hh, BF16, splits, variant
            dummy = torch.zeros(4, device="cuda")
            wp.ws = ptr(dummy)
            ns = lib.gdrn_conv3x3_wgrad_splits(C.byref(wp))
            ws = torch.empty(ns * C_ * C_ * 9, device="cuda")
            wp.ws = ptr(ws)
            for _ in range(2):
                check(lib.gdrn_conv3x3_wgrad(C.byref(wp), H.stream()), "wgrad")
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                lib.gdrn_conv3x3_wgrad(C.byref(wp), H.stream())
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 10 * 1e3
            line += f"  v{variant}/s{ns}: {us:7.1f} us {2.0*B*Hh*Hh*C_*C_*9/us/1e6:6.0f} TF"
    print(line, flush=True)
