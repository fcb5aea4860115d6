"""round 6: is the grouped halo weight gradient at ONE workgroup per CU (the side-stream configuration of the step: 84 KiB LDS request) waiting for
its global loads?  The head bucket's six 256-channel layers as one grouped launch, two workgroups per CU vs one, with the shipped kernel and with a
probe build (-DWGRAD_PROBE_SAMEPATCH: every stage re-fetches the first patch of its range -> cache hits; same instruction stream, same MFMAs).
    python tools/wgrad_lat_probe.py          (builds lib/libwgrad_probe.so from csrc/conv3x3_wgrad.hip on the fly)"""
import ctypes as C, os, subprocess, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import hiputil as H
from gdrnet_amd import cabi
from gdrnet_amd.cabi import BF16, WgradParams, check, ptr, to_device_table
lib = cabi.load()
probe_so = os.path.join(R, "gdr-net_amd", "lib", "libwgrad_probe.so")
if not os.path.exists(probe_so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DWGRAD_PROBE_SAMEPATCH", "-I", os.path.join(R, "include"),
                           os.path.join(R, "gdr-net_amd", "csrc", "conv3x3_wgrad.hip"), "-o", probe_so])
C.CDLL(cabi.lib_path(), mode=C.RTLD_GLOBAL)   # (the probe object refers to the library's thread-local error slot)
plib = C.CDLL(probe_so)
for l in (plib,):
    l.gdrn_conv3x3_wgrad_multi_lds.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    l.gdrn_conv3x3_wgrad_multi_lds.restype = C.c_int
B = 64

def params(C_, Hh, splits):
    x = torch.randn(B, Hh, Hh, C_, device="cuda").to(torch.bfloat16)
    dy = torch.randn(B, Hh, Hh, C_, device="cuda").to(torch.bfloat16)
    wp = WgradParams()
    wp.x, wp.dy = ptr(x), ptr(dy)
    wp.Hi = wp.Wi = wp.Ho = wp.Wo = Hh
    wp.Cin = wp.x_cs = wp.Cout = wp.dy_cs = C_
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

for name, shapes in (("head bucket (2x 256@64, 2x 256@32, 2x 256@16)", [(256, 64), (256, 64), (256, 32), (256, 32), (256, 16), (256, 16)]),
                     ("layer3 (12x 256@16)", [(256, 16)] * 12), ("layer2+layer1 (7x 128@32, 6x 64@64)", [(128, 32)] * 7 + [(64, 64)] * 6)):
    for nblocks_target in (768, 1536):
        units = [B * (h // 8) * (h // 8) * 2 for _, h in shapes]
        tiles = [(c // 64) ** 2 for c, _ in shapes]
        per = max(16, sum(u * t for u, t in zip(units, tiles)) // nblocks_target)
        wps, keep, starts, fl = [], [], [0], 0.0
        for (C_, Hh), u, t in zip(shapes, units, tiles):
            wp, k = params(C_, Hh, max(1, u // per))
            wps.append(wp); keep.append(k)
            starts.append(starts[-1] + t * wp.splits)
            fl += 2.0 * B * Hh * Hh * C_ * C_ * 9
        tab = to_device_table(wps, "cuda")
        stt = torch.tensor(starts, dtype=torch.int32, device="cuda")
        line = f"{name}, {starts[-1]} workgroups:"
        for lds, tag in ((0, "2/CU"), (84 * 1024, "1/CU")):
            for l, lt in ((lib, "shipped"), (plib, "same-patch probe")):
                us = timeit(lambda: check(l.gdrn_conv3x3_wgrad_multi_lds(ptr(tab), ptr(stt), len(wps), starts[-1], lds, H.stream()), "wgrad"))
                line += f"  [{tag} {lt}] {us:7.1f} us {fl/us/1e6:5.0f} TF"
        print(line, flush=True)
