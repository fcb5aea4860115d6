"""Thin test-side wrappers that call the C-ABI entry points of libgdrn_hip.so on torch tensors."""
import ctypes as C

import numpy as np
import torch

from gdrnet_amd import cabi
from gdrnet_amd.cabi import BF16, F16, F32, ConvParams, WgradParams, check, ptr

DEV = "cuda:0"


def tdt(dt):
    return {BF16: torch.bfloat16, F16: torch.float16}.get(dt, torch.float32)


def stream():
    return torch.cuda.current_stream().cuda_stream


def ru(a, b):
    return (a + b - 1) // b * b


def rel(a, b):
    a = a.detach().double().cpu().flatten()
    b = b.detach().double().cpu().flatten()
    return float((a - b).norm() / max(b.norm().item(), 1e-30))


def nhwc(x, dt, cpad=None):
    """NCHW fp32 cpu -> NHWC device tensor of dtype dt, channels zero-padded to cpad."""
    x = x.permute(0, 2, 3, 1).contiguous()
    if cpad is not None and cpad > x.shape[-1]:
        x = torch.nn.functional.pad(x, (0, cpad - x.shape[-1]))
    return x.to(DEV).to(tdt(dt)).contiguous()


def nchw(y, C_=None):
    y = y.float().cpu()
    if C_ is not None:
        y = y[..., :C_]
    return y.permute(0, 3, 1, 2).contiguous()


def rounded(x, dt):
    """value a tensor has after storage in dt (for building references of bf16 runs)."""
    return x.to(tdt(dt)).float()


def pack(w_src, A1, A2, T, B, A1v, A2v, Bv, s1, s2, st, sb, flip, dt):
    lib = cabi.load(dt)
    dst = torch.zeros(A1, A2, T, B, dtype=tdt(dt), device=DEV)
    w = w_src.to(DEV).float().contiguous()
    check(lib.gdrn_pack4(ptr(w), ptr(dst), A1, A2, T, B, A1v, A2v, Bv, s1, s2, st, sb, flip, dt, stream()), "pack4")
    return dst


def bn_rows(c):
    return 64 if c <= 64 else ru(c, 128)


def pack_fwd(w, dt, cin_p=None):
    """OIHW -> [rows][KK][cin_p]"""
    O, I, KH, KW = w.shape
    KK = KH * KW
    cin_p = cin_p or ru(I, 64)
    return pack(w, bn_rows(O), 1, KK, cin_p, O, 1, I, I * KK, 0, 1, KK, 0, dt).view(bn_rows(O), KK, cin_p)


def pack_dgrad(w, dt, flip, rows_valid_pad=None, cout_p=None):
    """OIHW -> [rows = in channels][KK][cout_p] (optionally tap-flipped)"""
    O, I, KH, KW = w.shape
    KK = KH * KW
    cout_p = cout_p or ru(O, 64)
    rows = bn_rows(rows_valid_pad or ru(I, 64))
    return pack(w, rows, 1, KK, cout_p, I, 1, O, KK, 0, 1, I * KK, flip, dt).view(rows, KK, cout_p)


def conv_gemm(x, w, B, Hi, Wi, Cin, x_cs, Ho, Wo, Cout, KH, KW, stride, pad, dt, mode=0, bias=None, addend=None, act=0,
              out_f32=0, y_cs=None, want_stats=False, halo=False, bnb=None, xf=None, v3=False, reps=0):
    """bnb (halo and generic kernel, bf16): dict(x, mask|None, mean, invstd, scale|None, shift|None) -> fused BatchNorm-backward statistics;
    the second return value is then the [tiles][2][Cout] rows buffer.
    xf (halo only): dict(mode, relu, x2, a, b, c, c2, msc, msh, out) device tensors -> operand transform while staging.
    v3 (with halo): True = the second-generation kernel (operand of gdrn_pack_wfrag32, w_frag = 2); 4 / 8 = the first kernel with
    gdrn_conv_params.halo_waves forced (the default, 0, lets the library pick by grid size: the small test grids get the eight-wave form).
    reps > 0: also time `reps` back-to-back launches with HIP events; the average (ms) is returned as a third value."""
    lib = cabi.load(dt)
    waves = v3 if (v3 is not True and v3 in (4, 8)) else 0   # v3 = 4 / 8: first halo kernel, forced four- / eight-wave form (halo_waves)
    v3 = v3 is True
    if halo:  # the halo kernels take a fragment-major permutation of the same operand
        wf = torch.empty_like(w)
        check((lib.gdrn_pack_wfrag32 if v3 else lib.gdrn_pack_wfrag)(ptr(w), ptr(wf), w.shape[0], Cin, dt, stream()), "pack_wfrag")
        w = wf
    y_cs = y_cs or ru(Cout, 4)
    ydt = torch.float32 if (out_f32 or dt == F32) else tdt(dt)
    y = torch.full((B, Ho, Wo, y_cs), float("nan"), dtype=ydt, device=DEV)
    cp = ConvParams()
    cp.x, cp.w, cp.y = ptr(x), ptr(w), ptr(y)
    cp.bias, cp.addend = ptr(bias), ptr(addend)
    cp.Hi, cp.Wi, cp.Cin, cp.x_cs = Hi, Wi, Cin, x_cs
    cp.Ho, cp.Wo, cp.Cout, cp.y_cs = Ho, Wo, Cout, y_cs
    cp.add_cs = addend.shape[-1] if addend is not None else 0
    cp.KH, cp.KW, cp.stride, cp.pad = KH, KW, stride, pad
    cp.mode, cp.act, cp.out_f32 = mode, act, out_f32
    cp.M = B * (Ho // 2) * (Wo // 2) if mode == 1 else B * Ho * Wo
    cp.w_rows, cp.dtype = w.shape[0], dt
    cp.w_frag = 2 if (halo and v3) else 0
    cp.halo_waves = waves if halo else 0
    cp.v3_min_wg = 1   # the 256-channel tile of the second-generation kernel also on the small grids of these tests
    # (every operand goes in before the tile / statistics-row queries: the second-generation kernel's tile depends on them)
    if bnb is not None:
        cp.bnb_mask = ptr(bnb.get("mask"))
    if xf is not None:
        cp.xf_mode, cp.xf_relu = xf["mode"], 1 if xf.get("relu") else 0
        cp.xf_x2, cp.xf_a, cp.xf_b, cp.xf_c, cp.xf_c2 = ptr(xf.get("x2")), ptr(xf.get("a")), ptr(xf.get("b")), ptr(xf.get("c")), ptr(xf.get("c2"))
        cp.xf_msc, cp.xf_msh, cp.xf_out = ptr(xf.get("msc")), ptr(xf.get("msh")), ptr(xf.get("out"))
    stats = None
    if want_stats:
        rows = (lib.gdrn_conv3x3_stats_rows if halo else lib.gdrn_conv_stats_rows)(C.byref(cp))
        stats = torch.zeros(rows, 2, Cout, dtype=torch.float32, device=DEV)
        cp.stats = ptr(stats)
    if bnb is not None:
        nrows = (lib.gdrn_conv3x3_stats_rows if halo else lib.gdrn_conv_stats_rows)(C.byref(cp))
        rows_t = torch.full((nrows, 2, Cout), float("nan"), dtype=torch.float32, device=DEV)
        cp.bnb_x, cp.bnb_mask, cp.bnb_cs = ptr(bnb["x"]), ptr(bnb.get("mask")), bnb["x"].shape[-1]
        cp.bnb_mean, cp.bnb_invstd, cp.bnb_scale, cp.bnb_shift = ptr(bnb["mean"]), ptr(bnb["invstd"]), ptr(bnb.get("scale")), ptr(bnb.get("shift"))
        cp.bnb_rows = ptr(rows_t)
        stats = rows_t  # per-tile rows [nrows][2][Cout]: callers sum over dim 0
    fn = lib.gdrn_conv3x3_halo if halo else lib.gdrn_conv_gemm
    check(fn(C.byref(cp), stream()), "conv")
    torch.cuda.synchronize()
    if reps > 0:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn(C.byref(cp), stream())
        e1.record()
        torch.cuda.synchronize()
        return y, stats, e0.elapsed_time(e1) / reps
    return y, stats


def conv_wgrad(x, dy, B, Hi, Wi, Cin, x_cs, Ho, Wo, Cout, dy_cs, KH, KW, stride, pad, dt, variant=0, splits=0, halo=False, ws=False):
    """ws: halo kernel with workspace partials + gdrn_wgrad_reduce_multi; returns the OIHW gradient in that case."""
    lib = cabi.load(dt)
    dw = torch.zeros(Cout, KH * KW, Cin, dtype=torch.float32, device=DEV)
    wp = WgradParams()
    wp.x, wp.dy, wp.dw = ptr(x), ptr(dy), ptr(dw)
    wp.Hi, wp.Wi, wp.Cin, wp.x_cs = Hi, Wi, Cin, x_cs
    wp.Ho, wp.Wo, wp.Cout, wp.dy_cs = Ho, Wo, Cout, dy_cs
    wp.KH, wp.KW, wp.stride, wp.pad = KH, KW, stride, pad
    wp.M, wp.dtype, wp.splits, wp.variant = B * Ho * Wo, dt, splits, variant
    if halo:
        assert lib.gdrn_conv3x3_wgrad_ok(C.byref(wp)) == 1
    if ws:
        from gdrnet_amd.cabi import WreduceTask, to_device_table

        wp.ws = ptr(dw)  # non-null: query the split count of the workspace mode
        ns = lib.gdrn_conv3x3_wgrad_splits(C.byref(wp))
        assert ns >= 1
        wsb = torch.full((ns * Cout * Cin * 9,), float("nan"), dtype=torch.float32, device=DEV)  # every slab must be written
        wp.ws, wp.dw = ptr(wsb), None
        check(lib.gdrn_conv3x3_wgrad(C.byref(wp), stream()), "conv3x3_wgrad(ws)")
        grad = torch.full((Cout, Cin, 3, 3), float("nan"), dtype=torch.float32, device=DEV)
        tab = to_device_table([WreduceTask(ws=ptr(wsb), dst=ptr(grad), nsplit=ns, Cout=Cout, Cin=Cin, cin_valid=0, s_co=Cin * 9, s_ci=9, s_t=1)], DEV)
        stt = torch.tensor([0, Cout * Cin // 256], dtype=torch.int32, device=DEV)
        check(lib.gdrn_wgrad_reduce_multi(ptr(tab), ptr(stt), 1, Cout * Cin // 256, stream()), "wgrad_reduce_multi")
        torch.cuda.synchronize()
        return grad
    check((lib.gdrn_conv3x3_wgrad if halo else lib.gdrn_conv_wgrad)(C.byref(wp), stream()), "conv_wgrad")
    torch.cuda.synchronize()
    return dw


def randn(seed, *shape):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g)
