"""Bring-up aid (not product): cycle stamps of the FIRST halo conv kernel (conv3x3_halo.hip built with -DHALO_DBG together with
conv3x3_v3.hip into tools/_dbg/libhalodbg.so): entry / prologue done / first barrier passed / loop end / exit / stores drained per wave."""
import ctypes as C
import os
import sys

import numpy as np
import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from gdrnet_amd.cabi import BF16, ConvParams, ptr  # noqa: E402

lib = C.CDLL(os.path.join(R, "tools", "_dbg", "libhalodbg.so"))
lib.gdrn_conv3x3_halo.argtypes = [C.POINTER(ConvParams), C.c_void_p]
lib.gdrn_conv3x3_tile.argtypes = [C.POINTER(ConvParams)] + [C.POINTER(C.c_int)] * 3
lib.gdrn_pack_wfrag.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
lib.gdrn_halo_set_dbg.argtypes = [C.c_void_p]


def run(B, C_, Hh, mode):
    x = torch.randn(B, Hh, Hh, C_, device="cuda").to(torch.bfloat16)
    x2 = torch.randn(B, Hh, Hh, C_, device="cuda").to(torch.bfloat16)
    w = (torch.randn(max(C_, 64), 9, C_, device="cuda") * 0.05).to(torch.bfloat16)
    wf = torch.empty_like(w)
    assert lib.gdrn_pack_wfrag(ptr(w), ptr(wf), w.shape[0], C_, BF16, None) == 0
    y = torch.empty(B, Hh, Hh, C_, device="cuda", dtype=torch.bfloat16)
    out = torch.empty_like(x)
    a, b, c = (torch.rand(C_, device="cuda") + 0.5 for _ in range(3))
    cp = ConvParams()
    cp.x, cp.w, cp.y = ptr(x), ptr(wf), ptr(y)
    cp.Hi = cp.Wi = cp.Ho = cp.Wo = Hh
    cp.Cin = cp.Cout = cp.x_cs = cp.y_cs = C_
    cp.KH = cp.KW = 3
    cp.stride = cp.pad = 1
    cp.M, cp.w_rows, cp.dtype = B * Hh * Hh, w.shape[0], BF16
    th, tw, bn = C.c_int(), C.c_int(), C.c_int()
    lib.gdrn_conv3x3_tile(C.byref(cp), C.byref(th), C.byref(tw), C.byref(bn))
    ntile = B * (Hh // th.value) * (Hh // tw.value)
    st = torch.zeros(ntile, 2, C_, device="cuda")
    if mode == "stats":
        cp.stats = ptr(st)
    elif mode == "xf1":
        cp.stats = ptr(st)
        cp.xf_mode, cp.xf_relu, cp.xf_a, cp.xf_c, cp.xf_out = 1, 1, ptr(a), ptr(c), ptr(out)
    elif mode == "xf3_bnb":
        cp.xf_mode, cp.xf_x2, cp.xf_a, cp.xf_b, cp.xf_c, cp.xf_out = 3, ptr(x2), ptr(a), ptr(b), ptr(c), ptr(out)
        cp.bnb_x, cp.bnb_cs, cp.bnb_mean, cp.bnb_invstd, cp.bnb_scale, cp.bnb_shift, cp.bnb_rows = ptr(x2), C_, ptr(a), ptr(b), ptr(a), ptr(c), ptr(st)
    nwg = ntile * max(C_ // bn.value, 1)
    dbg = torch.zeros(nwg * 4 * 8, dtype=torch.int64, device="cuda")
    assert lib.gdrn_halo_set_dbg(ptr(dbg)) == 0
    for _ in range(2):
        assert lib.gdrn_conv3x3_halo(C.byref(cp), None) == 0
    torch.cuda.synchronize()
    dbg.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    assert lib.gdrn_conv3x3_halo(C.byref(cp), None) == 0
    e1.record()
    torch.cuda.synchronize()
    d = dbg.cpu().numpy().reshape(nwg, 4, 8).astype(np.float64)
    t0 = d[:, :, 0].min()
    ent, pro, bar, le, end, dr = (d[:, :, i] for i in range(6))
    stages = (C_ // 64) * 9
    print("%-8s C=%d H=%d tile %dx%dx%d: %d workgroups, launch %.1f us | per wave cycles: prologue %.0f (+barrier %.0f) | loop %.0f (%.0f per tap stage; ideal %d) | epilogue %.0f (+drain %.0f) | "
          "first workgroup starts at 0, last at %.0f, all done at %.0f" % (
              mode, C_, Hh, th.value, tw.value, bn.value, nwg, e0.elapsed_time(e1) * 1e3, (pro - ent).mean(), (bar - pro).mean(), (le - bar).mean(),
              (le - bar).mean() / stages, 512 * bn.value // 128, (end - le).mean(), (dr - end).mean(), ent.min(axis=1).max() - t0, dr.max() - t0))


for (C_, Hh) in ((256, 16), (128, 32), (512, 8), (64, 64), (256, 64)):
    for mode in ("stats", "xf1", "xf3_bnb"):
        run(64, C_, Hh, mode)
