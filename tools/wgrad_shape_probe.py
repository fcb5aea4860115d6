"""round 6: which layers make the layer4 + layer3 bucket's grouped weight gradient slow (546 TFLOP/s against 870-990 for the other buckets)?
Grouped launches of one shape class at a time, one / two workgroups per CU, several pixel-range lengths per workgroup.
    python tools/wgrad_shape_probe.py"""
import ctypes as C, os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import hiputil as H
from gdrnet_amd import cabi
from gdrnet_amd.cabi import BF16, WgradParams, check, ptr, to_device_table
lib = cabi.load()
B = 64

def params(C_, Hh, splits, cs=None):
    cs = cs or C_
    x = torch.randn(B, Hh, Hh, cs, device="cuda").to(torch.bfloat16)
    dy = torch.randn(B, Hh, Hh, cs, device="cuda").to(torch.bfloat16)
    wp = WgradParams()
    wp.x, wp.dy = ptr(x), ptr(dy)
    wp.Hi = wp.Wi = wp.Ho = wp.Wo = Hh
    wp.Cin = wp.Cout = C_
    wp.x_cs = wp.dy_cs = cs
    wp.KH = wp.KW = 3; wp.stride = 1; wp.pad = 1
    wp.M, wp.dtype, wp.splits, wp.variant = B * Hh * Hh, BF16, splits, 0
    dummy = torch.zeros(4, device="cuda")
    wp.ws = ptr(dummy)
    ns = lib.gdrn_conv3x3_wgrad_splits(C.byref(wp))
    ws = torch.empty(ns * C_ * C_ * 9, device="cuda")
    wp.ws = ptr(ws)
    wp.splits = ns
    return wp, (x, dy, ws)

def timeit(fn, n=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

cases = (("layer4 (5x 512@8)", [(512, 8, None)] * 5), ("layer3 (11x 256@16)", [(256, 16, None)] * 11),
         ("layer4 + layer3", [(512, 8, None)] * 5 + [(256, 16, None)] * 11),
         ("256@8 with a 512-channel pixel stride (5x)", [(256, 8, 512)] * 5), ("256@8 (5x)", [(256, 8, None)] * 5), ("512@16 (5x)", [(512, 16, None)] * 5),
         ("256@16 with a 512-channel pixel stride (11x)", [(256, 16, 512)] * 11))
for name, shapes in cases:
    for per in (32, 64, 128, 256):
        units = [B * (h // 8) * (h // 8) * 2 for _, h, _ in shapes]
        tiles = [(c // 64) ** 2 for c, _, _ in shapes]
        wps, keep, starts, fl = [], [], [0], 0.0
        for (C_, Hh, cs), u, t in zip(shapes, units, tiles):
            wp, k = params(C_, Hh, max(1, u // per), cs)
            wps.append(wp); keep.append(k)
            starts.append(starts[-1] + t * wp.splits)
            fl += 2.0 * B * Hh * Hh * C_ * C_ * 9
        order = sorted(range(len(wps)), key=lambda i: -(units[i] // wps[i].splits))
        wps = [wps[i] for i in order]
        starts = [0]
        for i in order:
            starts.append(starts[-1] + tiles[i] * wps[len(starts) - 1].splits)
        tab = to_device_table(wps, "cuda")
        stt = torch.tensor(starts, dtype=torch.int32, device="cuda")
        line = f"{name}, {per} k-steps per workgroup, {starts[-1]} workgroups:"
        for lds, tag in ((0, "2/CU"), (84 * 1024, "1/CU")):
            us = timeit(lambda: check(lib.gdrn_conv3x3_wgrad_multi_lds(ptr(tab), ptr(stt), len(wps), starts[-1], lds, H.stream()), "wgrad"))
            line += f"  [{tag}] {us:7.1f} us {fl/us/1e6:5.0f} TF"
        print(line, flush=True)
        del keep
