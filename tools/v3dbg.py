"""Bring-up aid (not product): cycle stamps of the second-generation halo conv.  The kernel source is built alone with -DV3_DBG
(hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -DV3_DBG -shared gdr-net_amd/csrc/conv3x3_v3.hip -o tools/_dbg/libv3dbg.so);
every wave writes entry / prologue-done / loop-start / loop-end / exit stamps (s_memtime) and the cycles it spent in the per-unit
s_waitcnt and in s_barrier."""
import ctypes as C
import os
import sys

import numpy as np
import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from gdrnet_amd.cabi import BF16, ConvParams, ptr  # noqa: E402

lib = None


def load(name):
    global lib
    lib = C.CDLL(os.path.join(R, "tools", "_dbg", name))
    lib.gdrn_v3_launch_c.argtypes = [C.POINTER(ConvParams), C.c_void_p]
    lib.gdrn_v3_tile_c.argtypes = [C.POINTER(ConvParams)] + [C.POINTER(C.c_int)] * 3
    lib.gdrn_pack_wfrag32.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.gdrn_v3_set_dbg.argtypes = [C.c_void_p]


def run(B, C_, Hh, stats=True):
    x = torch.randn(B, Hh, Hh, C_, device="cuda").to(torch.bfloat16)
    w = (torch.randn(C_, 9, C_, device="cuda") * 0.05).to(torch.bfloat16)
    wf = torch.empty_like(w)
    assert lib.gdrn_pack_wfrag32(ptr(w), ptr(wf), C_, C_, BF16, None) == 0
    y = torch.empty(B, Hh, Hh, C_, device="cuda", dtype=torch.bfloat16)
    cp = ConvParams()
    cp.x, cp.w, cp.y = ptr(x), ptr(wf), ptr(y)
    cp.Hi = cp.Wi = cp.Ho = cp.Wo = Hh
    cp.Cin = cp.Cout = cp.x_cs = cp.y_cs = C_
    cp.KH = cp.KW = 3
    cp.stride = cp.pad = 1
    cp.M, cp.w_rows, cp.dtype, cp.w_frag = B * Hh * Hh, C_, BF16, 2
    th, tw, bn = C.c_int(), C.c_int(), C.c_int()
    cfg = lib.gdrn_v3_tile_c(C.byref(cp), C.byref(th), C.byref(tw), C.byref(bn))
    ntile = B * (Hh // th.value) * (Hh // 16)
    st = torch.zeros(ntile, 2, C_, device="cuda")
    if stats:
        cp.stats = ptr(st)
    nwg = ntile * (C_ // bn.value)
    nw = 8 if cfg in (1, 2) else 4
    dbg = torch.zeros(nwg * nw * 12, dtype=torch.int64, device="cuda")
    assert lib.gdrn_v3_set_dbg(ptr(dbg)) == 0
    for _ in range(2):
        assert lib.gdrn_v3_launch_c(C.byref(cp), None) == 0
    torch.cuda.synchronize()
    dbg.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    assert lib.gdrn_v3_launch_c(C.byref(cp), None) == 0
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    d = dbg.cpu().numpy().reshape(nwg, nw, 12).astype(np.float64)
    t0 = d[:, :, 0].min()
    ent, pro, l0, l1, end, wm, wb = (d[:, :, i] for i in range(7))
    units = (C_ // 64) * (18 if cfg in (1, 3) else 9)
    print("C=%d H=%d cfg %d tile %dx%dx%d: %d workgroups, launch %.1f us, span of stamps %.0f cycles (%.1f us at 2.4 GHz)" % (
        C_, Hh, cfg, th.value, tw.value, bn.value, nwg, us, d[:, :, 4].max() - t0, (d[:, :, 4].max() - t0) / 2400))
    print("   per wave, mean cycles: prologue %.0f | first barrier %.0f | loop %.0f (%.0f per unit, %d units; ideal %d per unit) | epilogue %.0f" % (
        (pro - ent).mean(), (l0 - pro).mean(), (l1 - l0).mean(), (l1 - l0).mean() / units, units, {1: 1024, 2: 512, 3: 1024, 4: 512}[cfg], (end - l1).mean()))
    e1, e2, dr = d[:, :, 8], d[:, :, 9], d[:, :, 10]
    print("   epilogue: lane reduction %.0f | rows (barrier + sum) %.0f | bias/convert/stores issued %.0f | store drain %.0f" % (
        (e1 - l1).mean(), (e2 - e1).mean(), (end - e2).mean(), (dr - end).mean()))
    print("   inside the loop: s_waitcnt vmcnt/lgkmcnt %.0f per unit, s_barrier %.0f per unit" % (wm.mean() / units, wb.mean() / units))
    first = ent.min(axis=1)
    order = np.argsort(first)
    print("   workgroup start spread: first %.0f, median %.0f, last %.0f cycles after the earliest; duration of a workgroup: mean %.0f" % (
        0, np.median(first) - t0, first.max() - t0, (end.max(axis=1) - first).mean()))


for name in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["libv3dbg.so"]):
    load(name)
    for cfgs in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["12"]):
        os.environ["GDRN_V3_CFG"] = cfgs
        print("==== %s GDRN_V3_CFG=%s" % (name, cfgs))
        for (C_, Hh) in ((256, 64), (256, 16), (128, 32)):
            run(64, C_, Hh)
