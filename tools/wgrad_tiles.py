"""round 4: the halo weight gradient on the step's stride-1 layer shapes, 64 x 64 tile (two workgroups per CU) against the 128 x 64 tile
(GDRN_WGRAD_W128: one wave per SIMD, accumulators in the AGPRs), isolated launches at bs = 64; per tile the automatic split count and a
one-workgroup-per-CU count.  Also the wide tile's grouped launch with fewer resident workgroups than CUs (what runs beside a gradient chain)."""
import ctypes as C, os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import hiputil as H
from gdrnet_amd import cabi
from gdrnet_amd.cabi import BF16, WgradParams, check, ptr, to_device_table
lib = cabi.load()
B = 64

def params(C_, Hh, variant, splits):
    x = torch.randn(B, Hh, Hh, C_, device="cuda").to(torch.bfloat16)
    dy = torch.randn(B, Hh, Hh, C_, device="cuda").to(torch.bfloat16)
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
    wp.splits = ns
    return wp, (x, dy, ws)

def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for (C_, Hh) in ((256, 64), (256, 32), (256, 16), (128, 32), (512, 8)):
    line = f"C={C_} H={Hh}:"
    fl = 2.0 * B * Hh * Hh * C_ * C_ * 9
    for variant in (0, 1):
        tiles = (C_ // (128 if variant else 64)) * (C_ // 64)
        for splits in (0, max(1, 256 // tiles), max(1, 512 // tiles)):
            wp, keep = params(C_, Hh, variant, splits)
            us = timeit(lambda: check(lib.gdrn_conv3x3_wgrad(C.byref(wp), H.stream()), "wgrad"))
            line += f"  v{variant}/s{wp.splits}: {us:7.1f} us {fl/us/1e6:6.0f} TF"
    print(line, flush=True)

# grouped launch of the head bucket's six 256-channel layers on the wide tile: all logical workgroups resident vs a subset
for nlog in (256, 384, 512):
    shapes = [(256, 64), (256, 64), (256, 32), (256, 32), (256, 16), (256, 16)]
    units = [B * (h // 8) * (h // 8) * 2 for _, h in shapes]
    per = max(16, sum(u * 8 for u in units) // nlog)
    wps, keep, starts, fl = [], [], [0], 0.0
    for (C_, Hh), u in zip(shapes, units):
        wp, k = params(C_, Hh, 1, max(1, u // per))
        wps.append(wp); keep.append(k)
        starts.append(starts[-1] + 8 * wp.splits)
        fl += 2.0 * B * Hh * Hh * C_ * C_ * 9
    tab = to_device_table(wps, "cuda")
    stt = torch.tensor(starts, dtype=torch.int32, device="cuda")
    line = f"head bucket x6 on the wide tile, {starts[-1]} logical workgroups:"
    for grid in (0, 256, 192, 128, 96, 64):
        us = timeit(lambda: check(lib.gdrn_conv3x3_wgrad_multi_w128(ptr(tab), ptr(stt), len(wps), starts[-1], grid, H.stream()), "w128"), 5)
        line += f"  grid {grid or starts[-1]}: {us:7.1f} us {fl/us/1e6:6.0f} TF"
    print(line, flush=True)
