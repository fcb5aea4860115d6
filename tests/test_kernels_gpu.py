"""Kernel-level parity tests (-m gpu): every HIP kernel, called through the C-ABI, against the same op
evaluated with plain PyTorch on the CPU (fp32).  fp32 kernels: tolerance 2e-5 relative L2 (different
summation order only).  16-bit kernels: inputs are rounded to the storage format first so the reference sees the same
operands; tolerance 6e-3 for bf16 (output rounding is 2^-9 ~ 2e-3 relative per element), 8e-4 for fp16 (2^-12).

The 16-bit format is a property of the library build (csrc/common.h): this module tests libgdrn_hip.so (bf16);
tests/test_kernels_fp16_gpu.py executes the same source a second time with `BF16` bound to the fp16 code and libgdrn_hip_f16.so."""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gdrnet_amd import cabi
from gdrnet_amd.cabi import F32, check, ptr

pytestmark = pytest.mark.gpu

BF16 = globals().get("__HALF__", cabi.BF16)       # the 16-bit dtype code under test ("BF16" reads "the build's half" below)
IS_F16 = BF16 == cabi.F16
HT = torch.float16 if IS_F16 else torch.bfloat16   # its torch dtype
DTS = [BF16] if IS_F16 else [F32, BF16]            # (the fp32 kernels are tested once, out of the bf16 library)
TOL = {F32: 2e-5, BF16: 8e-4 if IS_F16 else 6e-3}


@pytest.fixture(scope="module")
def H():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import hiputil

    cabi.load(BF16)
    return hiputil


# ---------------------------------------------------------------------------------------------- conv forward
CONV_CASES = [
    # B, I, O, H, k, stride, pad
    (2, 64, 64, 16, 3, 1, 1),
    (3, 64, 128, 16, 3, 2, 1),
    (2, 128, 256, 8, 1, 2, 0),
    (1, 256, 256, 16, 3, 1, 1),
    (5, 512, 512, 8, 3, 1, 1),
    (2, 256, 69, 16, 1, 1, 0),
    (2, 69, 128, 16, 3, 2, 1),
    (3, 64, 64, 7, 3, 1, 1),   # M not a multiple of the tile
]


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_forward(H, dt, case):
    B, I, O, Hh, k, s, p = case
    x = H.rounded(H.randn(1, B, I, Hh, Hh), dt)
    w = H.rounded(H.randn(2, O, I, k, k) / math.sqrt(I * k * k), dt)
    Ho = (Hh + 2 * p - k) // s + 1
    cin_p = H.ru(I, 64)
    xd = H.nhwc(x, dt, cin_p)
    wp = H.pack_fwd(w, dt, cin_p)
    y, stats = H.conv_gemm(xd, wp, B, Hh, Hh, cin_p, cin_p, Ho, Ho, O, k, k, s, p, dt, want_stats=True)
    ref = F.conv2d(x, w, None, s, p)
    assert H.rel(H.nchw(y, O), ref) < TOL[dt]
    st = stats.sum(0).cpu()
    assert H.rel(st[0], ref.sum((0, 2, 3))) < 1e-3 + TOL[dt]
    assert H.rel(st[1], (ref * ref).sum((0, 2, 3))) < 1e-3


HALO_CASES = [(2, 64, 64, 16), (1, 128, 128, 32), (3, 256, 256, 16), (2, 512, 512, 8), (1, 256, 256, 64), (2, 64, 128, 24)]
# second-generation kernel (conv3x3_v3.hip, v3=True): Cout % 128 == 0, W % 16 == 0.  The library gives the 16x16x256 tile only grids of
# >= 256 workgroups; gdrn_conv_params.v3_min_wg = 1 (tests/hiputil.py) lets the 256-channel cases of these lists run it on small grids, the
# other cases run the 8x16x128 K-split tile
V3_CASES = [(1, 128, 128, 32), (3, 256, 256, 16), (1, 128, 256, 32), (2, 512, 128, 16), (2, 256, 256, 32), (1, 512, 256, 48)]


# 4: the four-wave form of the 128-channel tile forced (halo_waves; False = the library picks: eight waves on these small grids)
W4_CASES = [c for c in HALO_CASES if c[2] >= 128]


@pytest.mark.parametrize("dt,v3,case", [(BF16, False, c) for c in HALO_CASES] + [(BF16, 4, c) for c in W4_CASES] + [(BF16, True, c) for c in V3_CASES] +
                         ([] if IS_F16 else [(F32, False, c) for c in HALO_CASES]))
def test_conv3x3_halo_forward_and_dgrad(H, dt, case, v3):
    """halo-tiled 3x3 s1 kernels: forward (+BN partial statistics, + addend epilogue) and data gradient (flipped weights).  fp32 (parity
    mode, r4): the 64-channel tile on v_mfma_f32_16x16x4_f32 with per-stage partial accumulators, against fp32 torch at 2e-5."""
    B, I, O, Hh = case
    x = H.rounded(H.randn(1, B, I, Hh, Hh), dt).requires_grad_(True)
    w = H.rounded(H.randn(2, O, I, 3, 3) / math.sqrt(I * 9), dt)
    add = H.rounded(H.randn(3, B, O, Hh, Hh), dt)
    ref = F.conv2d(x, w, None, 1, 1)
    dy = H.rounded(H.randn(4, B, O, Hh, Hh), dt)
    ref.backward(dy)
    xd, wp = H.nhwc(x.detach(), dt), H.pack_fwd(w, dt)
    y, stats = H.conv_gemm(xd, wp, B, Hh, Hh, I, I, Hh, Hh, O, 3, 3, 1, 1, dt, want_stats=True, halo=True, v3=v3)
    assert H.rel(H.nchw(y, O), ref) < TOL[dt]
    st = stats.sum(0).cpu()
    assert H.rel(st[0], ref.detach().sum((0, 2, 3))) < 1e-3 + TOL[dt]
    assert H.rel(st[1], (ref.detach() ** 2).sum((0, 2, 3))) < 1e-3
    y2, _ = H.conv_gemm(xd, wp, B, Hh, Hh, I, I, Hh, Hh, O, 3, 3, 1, 1, dt, addend=H.nhwc(add, dt), act=1, halo=True, v3=v3)
    assert H.rel(H.nchw(y2, O), F.relu(ref.detach() + add)) < TOL[dt]
    bias = H.randn(5, O)
    y3, _ = H.conv_gemm(xd, wp, B, Hh, Hh, I, I, Hh, Hh, O, 3, 3, 1, 1, dt, bias=bias.to(H.DEV), act=1, halo=True, v3=v3)
    assert H.rel(H.nchw(y3, O), F.relu(ref.detach() + bias.view(1, -1, 1, 1))) < TOL[dt]
    if I % 128 == 0 or v3 is not True:
        wd = H.pack_dgrad(w, dt, flip=1)
        dx, _ = H.conv_gemm(H.nhwc(dy, dt), wd, B, Hh, Hh, O, O, Hh, Hh, I, 3, 3, 1, 1, dt, halo=True, v3=v3)
        assert H.rel(H.nchw(dx, I), x.grad) < TOL[dt]


@pytest.mark.parametrize("dt", DTS)
def test_conv_epilogue_bias_act_addend_f32out(H, dt):
    B, I, O, Hh = 2, 256, 69, 16
    x = H.rounded(H.randn(3, B, I, Hh, Hh), dt)
    w = H.rounded(H.randn(4, O, I, 1, 1) / 16, dt)
    bias = H.randn(5, O)
    add = H.rounded(H.randn(6, B, O, Hh, Hh), dt)
    xd, wp = H.nhwc(x, dt), H.pack_fwd(w, dt)
    addd = H.nhwc(add, dt, 72)
    for act in (0, 1, 2):
        y, _ = H.conv_gemm(xd, wp, B, Hh, Hh, I, I, Hh, Hh, O, 1, 1, 1, 0, dt, bias=bias.to(H.DEV), addend=addd, act=act, out_f32=1, y_cs=72)
        ref = F.conv2d(x, w, bias) + add
        ref = ref if act == 0 else (F.relu(ref) if act == 1 else F.leaky_relu(ref, 0.1))
        assert y.dtype == torch.float32
        assert H.rel(H.nchw(y, O), ref) < (2e-5 if dt == F32 else 1e-3)


@pytest.mark.parametrize("dt", DTS)
def test_stem_conv(H, dt):
    """7x7 s2 p3 conv on a zero-padded NHWC4 image as a 7x1-tap gather of 16 px * 4 ch (resnet_backbone.py:23)."""
    lib = cabi.load(BF16)
    B = 2
    img = H.rounded(torch.rand(B, 3, 256, 256, generator=torch.Generator().manual_seed(7)), dt)
    w = H.rounded(H.randn(8, 64, 3, 7, 7) / 12, dt)
    imgp = torch.zeros(B, 262, 272, 4, dtype=H.tdt(dt), device=H.DEV)
    check(lib.gdrn_pack_image(ptr(img.to(H.DEV)), ptr(imgp), B, 256, 256, 262, 272, dt, H.stream()), "pack_image")
    wp = torch.zeros(64, 7, 64, dtype=H.tdt(dt), device=H.DEV)
    check(lib.gdrn_pack_stem_w(ptr(w.to(H.DEV)), ptr(wp), dt, H.stream()), "pack_stem_w")
    y, stats = H.conv_gemm(imgp, wp, B, 262, 272, 64, 4, 128, 128, 64, 7, 1, 2, 0, dt, want_stats=True)
    ref = F.conv2d(img, w, None, 2, 3)
    assert H.rel(H.nchw(y), ref) < TOL[dt]
    # weight gradient through the same gather + unpack
    dy = H.rounded(H.randn(9, B, 64, 128, 128) * 0.1, dt)
    dw = H.conv_wgrad(imgp, H.nhwc(dy, dt), B, 262, 272, 64, 4, 128, 128, 64, 64, 7, 1, 2, 0, dt)
    g = torch.zeros(64, 3, 7, 7, device=H.DEV)
    check(lib.gdrn_unpack_stem_w(ptr(dw), ptr(g), H.stream()), "unpack_stem_w")
    wr = w.clone().requires_grad_(True)
    F.conv2d(img, wr, None, 2, 3).backward(dy)
    assert H.rel(g, wr.grad) < (1e-4 if dt == F32 else 1e-2)


@pytest.mark.parametrize("dt", DTS)
def test_fc_as_conv(H, dt):
    """fc1 = 8x8 'valid' conv over the [B,8,8,128] map with the NCHW flatten order c*64+h*8+w (conv_pnp_net.py:140-152)."""
    B = 4
    x = H.rounded(H.randn(10, B, 128, 8, 8), dt)
    w = H.rounded(H.randn(11, 1024, 8192) / 90, dt)
    b = H.randn(12, 1024)
    xd = H.nhwc(x, dt)
    wp = H.pack_fwd(w.view(1024, 128, 8, 8), dt)
    y, _ = H.conv_gemm(xd, wp, B, 8, 8, 128, 128, 1, 1, 1024, 8, 8, 1, 0, dt, bias=b.to(H.DEV), act=2)
    ref = F.leaky_relu(F.linear(x.reshape(B, -1), w, b), 0.1)
    assert H.rel(y.view(B, 1024), ref) < TOL[dt]


# ---------------------------------------------------------------------------------------------- data gradient
@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("case", [(2, 64, 64, 16, 3, 1, 1), (2, 64, 128, 16, 3, 2, 1), (3, 128, 256, 8, 1, 2, 0), (2, 256, 69, 8, 1, 1, 0),
                                  (2, 69, 128, 16, 3, 2, 1)])
def test_conv_dgrad(H, dt, case):
    B, I, O, Hh, k, s, p = case
    x = H.randn(20, B, I, Hh, Hh).requires_grad_(True)
    w = H.rounded(H.randn(21, O, I, k, k) / math.sqrt(O * k * k), dt)
    Ho = (Hh + 2 * p - k) // s + 1
    dy = H.rounded(H.randn(22, B, O, Ho, Ho), dt)
    F.conv2d(x, w, None, s, p).backward(dy)
    op, ip = H.ru(O, 64), H.ru(I, 64)
    dyd = H.nhwc(dy, dt, op)
    if s == 1:
        wd = H.pack_dgrad(w, dt, flip=1 if k == 3 else 0, cout_p=op)
        dx, _ = H.conv_gemm(dyd, wd, B, Ho, Ho, op, op, Hh, Hh, ip, k, k, 1, p, dt)
    else:
        wd = H.pack_dgrad(w, dt, flip=0, cout_p=op)
        dx, _ = H.conv_gemm(dyd, wd, B, Ho, Ho, op, op, Hh, Hh, ip, k, k, 2, p, dt, mode=1)
    assert H.rel(H.nchw(dx, I), x.grad) < TOL[dt]
    if ip > I:
        assert float(dx[..., I:].float().abs().max()) == 0.0


@pytest.mark.parametrize("dt", DTS)
def test_conv_transpose_fwd_bwd(H, dt):
    """ConvTranspose2d(512,256,3,s2,p1,op1) (cdpn_rot_head_region.py:81-91): forward = transposed gather, data
    gradient = stride-2 conv, weight gradient = conv wgrad with the roles of input / output grad swapped."""
    B = 2
    x = H.rounded(H.randn(30, B, 512, 8, 8), dt).requires_grad_(True)
    w = H.rounded(H.randn(31, 512, 256, 3, 3) / 34, dt).requires_grad_(True)
    ref = F.conv_transpose2d(x, w, None, 2, 1, 1)
    dy = H.rounded(H.randn(32, B, 256, 16, 16), dt)
    ref.backward(dy)
    KK = 9
    wf = H.pack(w.detach(), 256, 1, KK, 512, 256, 1, 512, KK, 0, 1, 256 * KK, 0, dt).view(256, KK, 512)
    wd = H.pack(w.detach(), 512, 1, KK, 256, 512, 1, 256, 256 * KK, 0, 1, KK, 0, dt).view(512, KK, 256)
    xd, dyd = H.nhwc(x.detach(), dt), H.nhwc(dy, dt)
    y, stats = H.conv_gemm(xd, wf, B, 8, 8, 512, 512, 16, 16, 256, 3, 3, 2, 1, dt, mode=1, want_stats=True)
    assert H.rel(H.nchw(y), ref) < TOL[dt]
    assert H.rel(stats.sum(0)[0].cpu(), ref.sum((0, 2, 3))) < 1e-3 + TOL[dt]
    dx, _ = H.conv_gemm(dyd, wd, B, 16, 16, 256, 256, 8, 8, 512, 3, 3, 2, 1, dt, mode=0)
    assert H.rel(H.nchw(dx), x.grad) < TOL[dt]
    dw = H.conv_wgrad(dyd, xd, B, 16, 16, 256, 256, 8, 8, 512, 512, 3, 3, 2, 1, dt)  # [ci=512][tap][co=256]
    assert H.rel(dw.view(512, 3, 3, 256).permute(0, 3, 1, 2), w.grad) < (1e-4 if dt == F32 else 1e-2)


# ---------------------------------------------------------------------------------------------- weight gradient
@pytest.mark.parametrize("variant", [0])
@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("case", [(2, 64, 64, 16, 3, 1, 1), (3, 64, 128, 16, 3, 2, 1), (2, 128, 128, 9, 3, 1, 1), (2, 128, 256, 8, 1, 2, 0),
                                  (2, 256, 69, 8, 1, 1, 0), (4, 1024, 256, 1, 1, 1, 0)])
def test_conv_wgrad(H, dt, case, variant):
    if dt == F32 and variant == 1:
        pytest.skip("variant only exists for bf16")
    B, I, O, Hh, k, s, p = case
    x = H.rounded(H.randn(40, B, I, Hh, Hh), dt)
    w = H.randn(41, O, I, k, k).requires_grad_(True)
    Ho = (Hh + 2 * p - k) // s + 1
    dy = H.rounded(H.randn(42, B, O, Ho, Ho), dt)
    F.conv2d(x, w, None, s, p).backward(dy)
    op = H.bn_rows(O)
    dw = H.conv_wgrad(H.nhwc(x, dt), H.nhwc(dy, dt, op), B, Hh, Hh, I, I, Ho, Ho, O, op, k, k, s, p, dt, variant=variant)
    got = dw.view(O, k, k, I).permute(0, 3, 1, 2)
    assert H.rel(got, w.grad) < (5e-5 if dt == F32 else 1e-2)


W128 = 1  # GDRN_WGRAD_W128: the 128 x 64 workgroup tile of rounds 4-5 -- removed in round 6, the library must reject it


@pytest.mark.parametrize("variant", [0])
@pytest.mark.parametrize("splits", [0, 1, 3])
@pytest.mark.parametrize("case", [(2, 64, 64, 16), (1, 128, 256, 32), (3, 256, 128, 8), (2, 64, 128, 24), (5, 512, 512, 8)])
def test_conv3x3_wgrad_halo(H, case, splits, variant):
    """halo-tiled weight gradient (nine taps from one staged pixel patch) against autograd; both workgroup tiles."""
    B, I, O, Hh = case
    dt = BF16
    wpq = cabi.WgradParams()
    wpq.Hi = wpq.Wi = wpq.Ho = wpq.Wo = Hh
    wpq.Cin, wpq.Cout, wpq.x_cs, wpq.dy_cs, wpq.KH, wpq.KW, wpq.stride, wpq.pad, wpq.M, wpq.dtype, wpq.variant = I, O, I, O, 3, 3, 1, 1, B * Hh * Hh, dt, W128
    assert cabi.load(BF16).gdrn_conv3x3_wgrad_ok(C.byref(wpq)) == 0   # (the wide-tile variant is gone)
    x = H.rounded(H.randn(43, B, I, Hh, Hh), dt)
    w = H.randn(44, O, I, 3, 3).requires_grad_(True)
    dy = H.rounded(H.randn(45, B, O, Hh, Hh), dt)
    F.conv2d(x, w, None, 1, 1).backward(dy)
    dw = H.conv_wgrad(H.nhwc(x, dt), H.nhwc(dy, dt), B, Hh, Hh, I, I, Hh, Hh, O, O, 3, 3, 1, 1, dt, splits=splits, halo=True, variant=variant)
    assert H.rel(dw.view(O, 3, 3, I).permute(0, 3, 1, 2), w.grad) < 1e-2


@pytest.mark.parametrize("variant", [0])
@pytest.mark.parametrize("splits", [0, 1, 5])
@pytest.mark.parametrize("case", [(2, 64, 64, 16), (1, 128, 256, 32), (3, 256, 128, 8), (5, 512, 512, 8), (8, 64, 64, 64)])
def test_conv3x3_wgrad_halo_workspace(H, case, splits, variant):
    """halo weight gradient with per-split workspace partials + multi-layer reduce (writes the OIHW gradient directly)."""
    B, I, O, Hh = case
    dt = BF16
    x = H.rounded(H.randn(46, B, I, Hh, Hh), dt)
    w = H.randn(47, O, I, 3, 3).requires_grad_(True)
    dy = H.rounded(H.randn(48, B, O, Hh, Hh), dt)
    F.conv2d(x, w, None, 1, 1).backward(dy)
    grad = H.conv_wgrad(H.nhwc(x, dt), H.nhwc(dy, dt), B, Hh, Hh, I, I, Hh, Hh, O, O, 3, 3, 1, 1, dt, splits=splits, halo=True, ws=True, variant=variant)
    assert torch.isfinite(grad).all()
    assert H.rel(grad, w.grad) < 1e-2


@pytest.mark.parametrize("variant", [0])
@pytest.mark.parametrize("splits", [0, 1, 3])
@pytest.mark.parametrize("ws", [False, True])
@pytest.mark.parametrize("case", [(2, 64, 128, 16), (3, 128, 256, 8), (2, 256, 512, 4), (1, 128, 128, 32), (5, 64, 64, 12)])
def test_conv3x3_wgrad_halo_stride2(H, case, splits, ws, variant):
    """stride-2 3x3 weight gradient on the halo kernel (4x8 output-pixel stages against a 9x17 input patch): the ResNet stage-entry
    convs / Patch-PnP convs, against autograd; atomics path and workspace + reduce path.  case = (B, Cin, Cout, Hout)."""
    B, I, O, Ho = case
    Hi = 2 * Ho
    dt = BF16
    x = H.rounded(H.randn(143, B, I, Hi, Hi), dt)
    w = H.randn(144, O, I, 3, 3).requires_grad_(True)
    dy = H.rounded(H.randn(145, B, O, Ho, Ho), dt)
    F.conv2d(x, w, None, 2, 1).backward(dy)
    if Ho % 4 or Ho % 8:  # Wo must be a multiple of 8, Ho of 4: other shapes stay on the generic kernel
        wp = cabi.WgradParams()
        wp.Hi = wp.Wi = Hi
        wp.Ho = wp.Wo = Ho
        wp.Cin, wp.Cout, wp.x_cs, wp.dy_cs, wp.KH, wp.KW, wp.stride, wp.pad, wp.M, wp.dtype = I, O, I, O, 3, 3, 2, 1, B * Ho * Ho, dt
        assert cabi.load(BF16).gdrn_conv3x3_wgrad_ok(C.byref(wp)) == 0
        return
    got = H.conv_wgrad(H.nhwc(x, dt), H.nhwc(dy, dt), B, Hi, Hi, I, I, Ho, Ho, O, O, 3, 3, 2, 1, dt, splits=splits, halo=True, ws=ws, variant=variant)
    if not ws:
        got = got.view(O, 3, 3, I).permute(0, 3, 1, 2)
    assert torch.isfinite(got).all()
    assert H.rel(got, w.grad) < 1e-2


def test_conv3x3_wgrad_halo_conv_transpose_and_padded_channels(H):
    """(a) ConvTranspose2d(512 -> 256, 3, stride 2, pad 1, output_padding 1) weight gradient as a stride-2 conv weight gradient with
    the roles swapped (x = gradient of the output, dy = the input), written straight into the (Cin, Cout, 3, 3) parameter layout;
    (b) Patch-PnP's first conv: 69 real input channels in a 128-channel operand -- the reduce skips the padded channels."""
    from gdrnet_amd.cabi import WgradParams, WreduceTask, to_device_table

    lib, dev, dt = cabi.load(BF16), H.DEV, BF16
    # (a)
    B, Ci, Co, Hin = 2, 128, 64, 8
    xin = H.rounded(H.randn(150, B, Ci, Hin, Hin), dt)
    w = H.randn(151, Ci, Co, 3, 3).requires_grad_(True)
    gout = H.rounded(H.randn(152, B, Co, 2 * Hin, 2 * Hin), dt)
    F.conv_transpose2d(xin, w, None, 2, 1, 1).backward(gout)
    # (b)
    B2, I2, O2, Ho2 = 2, 69, 128, 16
    x2 = H.rounded(H.randn(153, B2, I2, 2 * Ho2, 2 * Ho2), dt)
    w2 = H.randn(154, O2, I2, 3, 3).requires_grad_(True)
    dy2 = H.rounded(H.randn(155, B2, O2, Ho2, Ho2), dt)
    F.conv2d(x2, w2, None, 2, 1).backward(dy2)
    jobs = [  # x (kernel input), dy, Hi, Ho, kernel Cin, kernel Cout, cin_valid, destination
        (H.nhwc(gout, dt), H.nhwc(xin, dt), 2 * Hin, Hin, Co, Ci, Co, torch.full((Ci, Co, 3, 3), float("nan"), device=dev), B),
        (H.nhwc(x2, dt, 128), H.nhwc(dy2, dt), 2 * Ho2, Ho2, 128, O2, I2, torch.full((O2, I2, 3, 3), float("nan"), device=dev), B2),
    ]
    keep = []
    for xk, dyk, Hi, Ho, cin, cout, civ, dst, Bk in jobs:
        wp = WgradParams()
        wp.x, wp.dy = ptr(xk), ptr(dyk)
        wp.Hi = wp.Wi = Hi
        wp.Ho = wp.Wo = Ho
        wp.Cin, wp.Cout, wp.x_cs, wp.dy_cs, wp.KH, wp.KW, wp.stride, wp.pad, wp.M, wp.dtype, wp.splits = cin, cout, cin, cout, 3, 3, 2, 1, Bk * Ho * Ho, dt, 2
        assert lib.gdrn_conv3x3_wgrad_ok(C.byref(wp)) == 1
        wp.ws = ptr(dst)
        ns = lib.gdrn_conv3x3_wgrad_splits(C.byref(wp))
        wsb = torch.full((ns * cout * cin * 9,), float("nan"), device=dev)
        wp.ws, wp.dw = ptr(wsb), None
        check(lib.gdrn_conv3x3_wgrad(C.byref(wp), H.stream()), "wgrad s2")
        tab = to_device_table([WreduceTask(ws=ptr(wsb), dst=ptr(dst), nsplit=ns, Cout=cout, Cin=cin, cin_valid=civ, s_co=civ * 9, s_ci=9, s_t=1)], dev)
        stt = torch.tensor([0, cout * cin // 256], dtype=torch.int32, device=dev)
        check(lib.gdrn_wgrad_reduce_multi(ptr(tab), ptr(stt), 1, cout * cin // 256, H.stream()), "reduce")
        keep += [wsb, tab, stt]
    torch.cuda.synchronize()
    assert H.rel(jobs[0][7], w.grad) < 1e-2
    assert torch.isfinite(jobs[1][7]).all() and H.rel(jobs[1][7], w2.grad) < 1e-2


@pytest.mark.parametrize("mask_kind", ["stored", "affine", "none"])
@pytest.mark.parametrize("v3,case", [(False, (2, 64, 64, 16)), (False, (1, 128, 256, 32)), (False, (3, 256, 128, 8)), (4, (1, 128, 256, 32)), (4, (3, 256, 128, 8)),
                                     (True, (1, 128, 256, 32)), (True, (3, 256, 128, 16)), (True, (2, 256, 256, 32))])
def test_conv3x3_halo_fused_bn_backward_stats(H, case, mask_kind, v3):
    """data-gradient launch with the fused BatchNorm(+ReLU) backward prologue: masked gradient + the two per-channel
    sums == conv -> mask -> gdrn_bn_bwd_reduce semantics, computed with torch."""
    B, I, O, Hh = case
    dt, dev = BF16, H.DEV
    x = H.rounded(H.randn(90, B, I, Hh, Hh), dt)
    w = H.rounded(H.randn(91, O, I, 3, 3) / math.sqrt(I * 9), dt)
    add = H.rounded(H.randn(92, B, O, Hh, Hh), dt)
    bx = H.rounded(H.randn(93, B, O, Hh, Hh) * 1.5 + 0.3, dt)       # raw input of the BatchNorm
    mean, invstd = H.randn(94, O) * 0.2, torch.rand(O, generator=torch.Generator().manual_seed(95)) + 0.5
    scale, shift = torch.rand(O, generator=torch.Generator().manual_seed(96)) + 0.5, H.randn(97, O) * 0.3
    ystored = H.rounded(F.relu(bx * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) + H.randn(98, B, O, Hh, Hh)), dt)  # with a residual
    g = F.conv2d(x, w, None, 1, 1) + add
    if mask_kind == "stored":
        m = ystored > 0
    elif mask_kind == "affine":
        m = (bx * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)) > 0
    else:
        m = torch.ones_like(g, dtype=torch.bool)
    gm = g * m
    xhat = (bx - mean.view(1, -1, 1, 1)) * invstd.view(1, -1, 1, 1)
    ref1, ref2 = gm.sum((0, 2, 3)), (gm * xhat).sum((0, 2, 3))
    d = lambda t: t.to(dev)
    bnb = dict(x=H.nhwc(bx, dt), mean=d(mean), invstd=d(invstd))
    if mask_kind == "stored":
        bnb["mask"] = H.nhwc(ystored, dt)
    elif mask_kind == "affine":
        bnb["scale"], bnb["shift"] = d(scale), d(shift)
    y, sums = H.conv_gemm(H.nhwc(x, dt), H.pack_fwd(w, dt), B, Hh, Hh, I, I, Hh, Hh, O, 3, 3, 1, 1, dt, addend=H.nhwc(add, dt), halo=True, bnb=bnb, v3=v3)
    assert H.rel(H.nchw(y, O), gm) < 1e-2
    got = sums.sum(0).cpu()
    assert H.rel(got[0], ref1) < 1e-2 and H.rel(got[1], ref2) < 1e-2
    if v3 is True and mask_kind != "stored":
        # without addend and stored mask the 256-channel tile takes the launch when the grid is large enough (the lean epilogue)
        y, sums = H.conv_gemm(H.nhwc(x, dt), H.pack_fwd(w, dt), B, Hh, Hh, I, I, Hh, Hh, O, 3, 3, 1, 1, dt, halo=True, bnb=bnb, v3=True)
        g0 = (g - add) * m
        assert H.rel(H.nchw(y, O), g0) < 1e-2
        got = sums.sum(0).cpu()
        assert H.rel(got[0], g0.sum((0, 2, 3))) < 1e-2 and H.rel(got[1], (g0 * xhat).sum((0, 2, 3))) < 1e-2


@pytest.mark.parametrize("kind", ["out1x1", "s2_dgrad", "convT_dgrad"])
def test_conv_gemm_fused_bn_backward_stats(H, kind):
    """the same fused BatchNorm(+ReLU)-backward epilogue on the generic kernel, for the data gradients that do not run on the halo
    kernel: the 1x1 output conv (affine mask), a stride-2 conv's transposed gather with an addend (stored mask, all four parity
    classes), the ConvTranspose's stride-2 gather (stored mask).  Reference: autograd data gradient -> mask -> the two sums."""
    dt, dev = BF16, H.DEV
    d = lambda t: t.to(dev)
    if kind == "out1x1":      # y = conv1x1(x): dgrad = dy (N) x W^T, C = 256 channels of the BatchNorm in front
        B, Cb, O, Hh, k, stride, pad = 2, 256, 128, 32, 1, 1, 0
    elif kind == "s2_dgrad":  # y = conv3x3 s2 (x), C = 64
        B, Cb, O, Hh, k, stride, pad = 2, 64, 128, 32, 3, 2, 1
    else:                     # y = conv3x3 s2 seen from the ConvTranspose side: its data gradient is a plain stride-2 conv of d_y
        B, Cb, O, Hh, k, stride, pad = 2, 128, 64, 16, 3, 2, 1
    gen = lambda sd, *sh: H.rounded(H.randn(sd, *sh), dt)
    bx = H.rounded(H.randn(193, B, Cb, Hh, Hh) * 1.5 + 0.3, dt)
    mean, invstd = H.randn(194, Cb) * 0.2, torch.rand(Cb, generator=torch.Generator().manual_seed(195)) + 0.5
    scale, shift = torch.rand(Cb, generator=torch.Generator().manual_seed(196)) + 0.5, H.randn(197, Cb) * 0.3
    ystored = H.rounded(F.relu(bx * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) + H.randn(198, B, Cb, Hh, Hh)), dt)
    bnb = dict(x=H.nhwc(bx, dt), mean=d(mean), invstd=d(invstd))
    if kind == "convT_dgrad":
        # forward y = convT(x_in [Cb ch, Hh]) -> [O, 2Hh]; data gradient g = conv3x3 s2 (dy) with the ConvTranspose weight [Cb][O][3][3]
        w = H.rounded(H.randn(191, Cb, O, 3, 3) / math.sqrt(O * 9), dt)
        dy = gen(190, B, O, 2 * Hh, 2 * Hh)
        g = F.conv2d(dy, w, None, 2, 1)
        wd = H.pack(w, Cb, 1, 9, O, Cb, 1, O, O * 9, 0, 1, 9, 0, dt).view(Cb, 9, O)   # rows = Cb (ConvT in channels), b = O
        m = ystored > 0
        bnb["mask"] = H.nhwc(ystored, dt)
        y, sums = H.conv_gemm(H.nhwc(dy, dt), wd, B, 2 * Hh, 2 * Hh, O, O, Hh, Hh, Cb, 3, 3, 2, 1, dt, mode=0, bnb=bnb)
    else:
        w = H.rounded(H.randn(191, O, Cb, k, k) / math.sqrt(Cb * k * k), dt)
        Ho = Hh // stride
        dy = gen(190, B, O, Ho, Ho)
        xin = torch.zeros(B, Cb, Hh, Hh, requires_grad=True)
        F.conv2d(xin, w, None, stride, pad).backward(dy)
        g = xin.grad
        if kind == "out1x1":
            m = (bx * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)) > 0
            bnb["scale"], bnb["shift"] = d(scale), d(shift)
            y, sums = H.conv_gemm(H.nhwc(dy, dt), H.pack_dgrad(w, dt, 0), B, Ho, Ho, O, O, Hh, Hh, Cb, 1, 1, 1, 0, dt, mode=0, bnb=bnb)
        else:
            add = gen(192, B, Cb, Hh, Hh)
            g = g + add
            m = ystored > 0
            bnb["mask"] = H.nhwc(ystored, dt)
            y, sums = H.conv_gemm(H.nhwc(dy, dt), H.pack_dgrad(w, dt, 0), B, Ho, Ho, O, O, Hh, Hh, Cb, 3, 3, 2, 1, dt, mode=1, addend=H.nhwc(add, dt), bnb=bnb)
    gm = g * m
    xhat = (bx - mean.view(1, -1, 1, 1)) * invstd.view(1, -1, 1, 1)
    ref1, ref2 = gm.sum((0, 2, 3)), (gm * xhat).sum((0, 2, 3))
    assert H.rel(H.nchw(y, Cb), gm) < 1e-2
    got = sums.sum(0).cpu()
    assert torch.isfinite(got).all()
    assert H.rel(got[0], ref1) < 1e-2 and H.rel(got[1], ref2) < 1e-2


@pytest.mark.parametrize("mode", [1, 2, 3, 4])
@pytest.mark.parametrize("v3,case", [(False, c) for c in [(2, 64, 64, 16), (1, 128, 256, 32), (3, 256, 128, 8), (2, 64, 128, 24), (1, 256, 256, 64), (2, 512, 512, 8)]]
                         + [(4, c) for c in [(1, 128, 256, 32), (3, 256, 128, 8), (2, 512, 512, 8)]]
                         + [(True, c) for c in [(1, 128, 256, 32), (3, 256, 128, 16), (2, 64, 128, 48), (2, 256, 256, 32), (1, 512, 256, 16)]])
def test_conv3x3_halo_operand_transform(H, case, mode, v3):
    """xf_mode 1-4 (gdrn_hip.h): the conv consumes v(x, x2) evaluated while the patch is staged -- the BatchNorm forward apply
    (+residual, +ReLU) of the producer / the BatchNorm backward apply in front of a data gradient -- and xf_out receives v.
    Reference: v in fp32 with torch, rounded to bf16, zero padding applied to v, F.conv2d."""
    B, I, O, Hh = case
    dt, dev = BF16, H.DEV
    gen = lambda sd, *sh: H.randn(sd, *sh)
    x = H.rounded(gen(70, B, I, Hh, Hh) * 1.3 + 0.2, dt)
    x2 = H.rounded(gen(71, B, I, Hh, Hh), dt)
    w = H.rounded(gen(72, O, I, 3, 3) / math.sqrt(I * 9), dt)
    a = torch.rand(I, generator=torch.Generator().manual_seed(73)) + 0.5
    b = gen(74, I) * 0.7
    c, c2 = gen(75, I) * 0.3, gen(76, I) * 0.2
    msc, msh = torch.rand(I, generator=torch.Generator().manual_seed(77)) + 0.5, gen(78, I) * 0.4
    V = lambda t: t.view(1, -1, 1, 1)
    relu = mode in (1, 2)
    if mode == 1:
        v = x * V(a) + V(c)
    elif mode == 2:
        v = (x * V(a) + V(c)) + H.rounded(x2 * V(b) + V(c2), dt)
    elif mode == 3:
        v = V(a) * x + (V(b) * x2 + V(c))
    else:
        v = V(a) * (x * ((x2 * V(msc) + V(msh)) > 0)) + (V(b) * x2 + V(c))
    if relu:
        v = F.relu(v)
    vb = H.rounded(v, dt)
    ref = F.conv2d(vb, w, None, 1, 1)
    d = lambda t: t.to(dev).float().contiguous()
    out = torch.full((B, Hh, Hh, I), float("nan"), dtype=HT, device=dev)
    xf = dict(mode=mode, relu=relu, a=d(a), c=d(c), out=out)
    if mode == 2:
        xf["c2"] = d(c2)
    if mode >= 2:
        xf.update(x2=H.nhwc(x2, dt), b=d(b))
    if mode == 4:
        xf.update(msc=d(msc), msh=d(msh))
    y, stats = H.conv_gemm(H.nhwc(x, dt), H.pack_fwd(w, dt), B, Hh, Hh, I, I, Hh, Hh, O, 3, 3, 1, 1, dt, want_stats=True, halo=True, xf=xf, v3=v3)
    # v itself: identical up to fp32 association (fma) -> at most rare one-ulp bf16 flips
    got_v = H.nchw(out, I)
    assert torch.isfinite(got_v).all()
    assert H.rel(got_v, vb) < 2e-3, H.rel(got_v, vb)
    assert float((got_v != vb).float().mean()) < 0.02
    assert H.rel(H.nchw(y, O), ref) < TOL[dt]
    st = stats.sum(0).cpu()
    assert H.rel(st[1], (ref ** 2).sum((0, 2, 3))) < 2e-3
    # without xf_out and with NULL a / b (= 1): same conv result
    if mode == 2:
        xf2 = dict(mode=2, relu=True, x2=H.nhwc(x2, dt), c=d(c))
        y2, _ = H.conv_gemm(H.nhwc(x, dt), H.pack_fwd(w, dt), B, Hh, Hh, I, I, Hh, Hh, O, 3, 3, 1, 1, dt, halo=True, xf=xf2, v3=v3)
        ref2 = F.conv2d(H.rounded(F.relu(x2 + (x + V(c))), dt), w, None, 1, 1)
        assert H.rel(H.nchw(y2, O), ref2) < TOL[dt]


@pytest.mark.parametrize("v3", [False, 4, True])
def test_conv3x3_halo_transform_with_fused_bn_backward_epilogue(H, v3):
    """the launch the engine uses for a BasicBlock conv1 data gradient: xf_mode 3 prologue (bn1 backward apply) + addend
    (residual-path gradient) + the fused mask / BatchNorm-backward sums epilogue of the previous block's bn2."""
    B, I, O, Hh = 2, 128, 128, 16
    dt, dev = BF16, H.DEV
    g = H.rounded(H.randn(50, B, I, Hh, Hh), dt)          # masked gradient w.r.t. bn1's output
    raw = H.rounded(H.randn(51, B, I, Hh, Hh) * 1.2, dt)  # bn1's raw input
    w = H.rounded(H.randn(52, O, I, 3, 3) / math.sqrt(I * 9), dt)
    add = H.rounded(H.randn(53, B, O, Hh, Hh), dt)
    ka, kb, kc = torch.rand(I) + 0.5, H.randn(54, I) * 0.1, H.randn(55, I) * 0.05
    bx = H.rounded(H.randn(56, B, O, Hh, Hh) * 1.5 + 0.3, dt)
    ystored = H.rounded(F.relu(bx + H.randn(57, B, O, Hh, Hh)), dt)
    mean, invstd = H.randn(58, O) * 0.2, torch.rand(O, generator=torch.Generator().manual_seed(59)) + 0.5
    V = lambda t: t.view(1, -1, 1, 1)
    v = H.rounded(V(ka) * g + (V(kb) * raw + V(kc)), dt)
    gg = (F.conv2d(v, w, None, 1, 1) + add) * (ystored > 0)
    xhat = (bx - V(mean)) * V(invstd)
    d = lambda t: t.to(dev).float().contiguous()
    out = torch.full((B, Hh, Hh, I), float("nan"), dtype=HT, device=dev)
    y, sums = H.conv_gemm(H.nhwc(g, dt), H.pack_fwd(w, dt), B, Hh, Hh, I, I, Hh, Hh, O, 3, 3, 1, 1, dt, addend=H.nhwc(add, dt), halo=True,
                          bnb=dict(x=H.nhwc(bx, dt), mask=H.nhwc(ystored, dt), mean=d(mean), invstd=d(invstd)),
                          xf=dict(mode=3, relu=False, x2=H.nhwc(raw, dt), a=d(ka), b=d(kb), c=d(kc), out=out), v3=v3)
    assert H.rel(H.nchw(out, I), v) < 2e-3
    assert H.rel(H.nchw(y, O), gg) < 1e-2
    got = sums.sum(0).cpu()
    assert H.rel(got[0], gg.sum((0, 2, 3))) < 1e-2 and H.rel(got[1], (gg * xhat).sum((0, 2, 3))) < 1e-2


@pytest.mark.parametrize("nrows,C_", [(16, 64), (128, 256), (2048, 256), (37, 512)])
def test_bn_bwd_coef(H, nrows, C_):
    """rows of BatchNorm-backward sums -> (a, b, c) of dx = a*g + b*x + c, dgamma, dbeta (gdrn_bn_bwd_apply's coefficients)."""
    lib, dev = cabi.load(BF16), H.DEV
    rows = H.randn(60, nrows, 2, C_)
    gamma, mean = torch.rand(C_) + 0.5, H.randn(61, C_) * 0.3
    invstd = torch.rand(C_, generator=torch.Generator().manual_seed(62)) + 0.5
    npix = 4096
    rows_d, gamma_d, mean_d, invstd_d = (t.to(dev).float().contiguous() for t in (rows, gamma, mean, invstd))  # alive across the launch
    o = [torch.full((C_,), float("nan"), device=dev) for _ in range(5)]
    check(lib.gdrn_bn_bwd_coef(ptr(rows_d), nrows, C_, npix, ptr(gamma_d), ptr(mean_d), ptr(invstd_d), ptr(o[0]), ptr(o[1]), ptr(o[2]), ptr(o[3]),
                               ptr(o[4]), H.stream()), "bn_bwd_coef")
    torch.cuda.synchronize()
    s = rows.double().sum(0)
    m1, m2 = s[0] / npix, s[1] / npix
    a = gamma.double() * invstd.double()
    b = -a * invstd.double() * m2
    c = -a * m1 - b * mean.double()
    assert H.rel(o[0], a) < 1e-6 and H.rel(o[1], b) < 1e-5 and H.rel(o[2], c) < 1e-5
    assert H.rel(o[3], s[1]) < 1e-6 and H.rel(o[4], s[0]) < 1e-6


@pytest.mark.parametrize("variant,grid", [(0, 0), (0, "one-per-cu")])
def test_conv3x3_wgrad_grouped(H, variant, grid):
    """three layers of different geometry in ONE grouped launch + one reduce launch == autograd, layer by layer; "one-per-cu": the launch the
    training step makes on its side stream (84 KiB LDS request = one workgroup per CU)."""
    from gdrnet_amd.cabi import WgradParams, WreduceTask, to_device_table

    lib = cabi.load(BF16)
    dt, dev = BF16, H.DEV
    cases = [(2, 64, 64, 32, 5), (3, 128, 256, 16, 2), (4, 512, 128, 8, 1)]  # B, I, O, H, requested splits
    keep, wps, tasks, refs, grads = [], [], [], [], []
    for i, (B, I, O, Hh, sp) in enumerate(cases):
        x = H.rounded(H.randn(60 + i, B, I, Hh, Hh), dt)
        w = H.randn(70 + i, O, I, 3, 3).requires_grad_(True)
        dy = H.rounded(H.randn(80 + i, B, O, Hh, Hh), dt)
        F.conv2d(x, w, None, 1, 1).backward(dy)
        refs.append(w.grad)
        xd, dyd = H.nhwc(x, dt), H.nhwc(dy, dt)
        wp = WgradParams()
        wp.x, wp.dy, wp.dw = ptr(xd), ptr(dyd), None
        wp.Hi, wp.Wi, wp.Cin, wp.x_cs, wp.Ho, wp.Wo, wp.Cout, wp.dy_cs = Hh, Hh, I, I, Hh, Hh, O, O
        wp.KH, wp.KW, wp.stride, wp.pad, wp.M, wp.dtype, wp.splits, wp.variant = 3, 3, 1, 1, B * Hh * Hh, dt, sp, variant
        wp.ws = ptr(xd)  # placeholder for the query
        wp.splits = lib.gdrn_conv3x3_wgrad_splits(C.byref(wp))
        assert 1 <= wp.splits <= sp
        ws = torch.full((wp.splits * O * I * 9,), float("nan"), dtype=torch.float32, device=dev)
        wp.ws = ptr(ws)
        g = torch.full((O, I, 3, 3), float("nan"), dtype=torch.float32, device=dev)
        keep += [xd, dyd, ws]
        grads.append(g)
        wps.append(wp)
        tasks.append(WreduceTask(ws=ptr(ws), dst=ptr(g), nsplit=wp.splits, Cout=O, Cin=I, cin_valid=0, s_co=I * 9, s_ci=9, s_t=1))
    st1, st2 = [0], [0]
    for wp in wps:
        st1.append(st1[-1] + (wp.Cout // 64) * (wp.Cin // 64) * wp.splits)
        st2.append(st2[-1] + wp.Cout * wp.Cin // 256)
    tab1, tab2 = to_device_table(wps, dev), to_device_table(tasks, dev)
    s1, s2 = torch.tensor(st1, dtype=torch.int32, device=dev), torch.tensor(st2, dtype=torch.int32, device=dev)
    if grid == "one-per-cu":
        check(lib.gdrn_conv3x3_wgrad_multi_lds(ptr(tab1), ptr(s1), len(wps), st1[-1], 84 * 1024, H.stream()), "conv3x3_wgrad_multi_lds")
    else:
        check(lib.gdrn_conv3x3_wgrad_multi(ptr(tab1), ptr(s1), len(wps), st1[-1], H.stream()), "conv3x3_wgrad_multi")
    check(lib.gdrn_wgrad_reduce_multi(ptr(tab2), ptr(s2), len(tasks), st2[-1], H.stream()), "wgrad_reduce_multi")
    torch.cuda.synchronize()
    for g, r in zip(grads, refs):
        assert torch.isfinite(g).all() and H.rel(g, r) < 1e-2



@pytest.mark.parametrize("shape", [(3, 8, 256), (4, 16, 256), (2, 32, 256), (2, 8, 64)])
@pytest.mark.parametrize("dt", DTS)
def test_bn_relu_upsample_fused_equals_the_two_launch_path_bit_for_bit(H, dt, shape):
    """r6: gdrn_bn_relu_upsample2x_fwd == gdrn_bn_apply (ReLU) followed by gdrn_upsample2x_fwd, bit for bit (the fused launch rounds every
    source value to the storage format exactly where the stored activation would have been rounded); and gdrn_upsample2x_bwd_bnsums ==
    gdrn_upsample2x_bwd followed by gdrn_bn_bwd_reduce (affine ReLU mask): dx bit for bit, the partial rows bit for bit (same workgroup
    geometry, same order of additions) -- cdpn_rot_head_region.py:103-123 (BatchNorm -> ReLU -> UpsamplingBilinear2d) and its backward."""
    lib = cabi.load(BF16)
    dev = H.DEV
    N, Hh, C_ = shape
    raw = H.nhwc(H.rounded(H.randn(300, N, C_, Hh, Hh) * 1.5, dt), dt)
    g = torch.Generator().manual_seed(7)
    scale = (0.5 + torch.rand(C_, generator=g)).to(dev)
    shift = (torch.rand(C_, generator=g) - 0.5).to(dev)
    mean = (torch.rand(C_, generator=g) - 0.5).to(dev)
    invstd = (0.5 + torch.rand(C_, generator=g)).to(dev)
    npix = N * Hh * Hh
    st = H.stream()
    act = torch.empty_like(raw)
    check(lib.gdrn_bn_apply(ptr(raw), ptr(scale), ptr(shift), None, ptr(act), npix, C_, 1, dt, st), "bn_apply")
    u_ref = torch.empty(N, 2 * Hh, 2 * Hh, C_, dtype=raw.dtype, device=dev)
    check(lib.gdrn_upsample2x_fwd(ptr(act), ptr(u_ref), N, Hh, Hh, C_, dt, st), "upsample_fwd")
    u = torch.full_like(u_ref, float("nan"))
    check(lib.gdrn_bn_relu_upsample2x_fwd(ptr(raw), ptr(scale), ptr(shift), ptr(u), N, Hh, Hh, C_, dt, st), "bn_relu_upsample_fwd")
    torch.cuda.synchronize()
    assert torch.equal(u, u_ref)
    # ... against torch on the rounded operands too
    ref = F.interpolate(H.rounded(F.relu(H.nchw(raw) * scale.cpu().view(1, -1, 1, 1) + shift.cpu().view(1, -1, 1, 1)), dt), scale_factor=2, mode="bilinear", align_corners=True)
    assert H.rel(H.nchw(u), ref) < TOL[dt]
    # backward
    d_u = H.nhwc(H.rounded(H.randn(301, N, C_, 2 * Hh, 2 * Hh), dt), dt)
    nrows = lib.gdrn_bn_bwd_reduce_rows(npix, C_, dt)
    dx_ref = torch.empty_like(raw)
    rows_ref = torch.full((nrows, 2, C_), float("nan"), device=dev)
    check(lib.gdrn_upsample2x_bwd(ptr(d_u), ptr(dx_ref), N, Hh, Hh, C_, dt, st), "upsample_bwd")
    check(lib.gdrn_bn_bwd_reduce(ptr(dx_ref), None, ptr(raw), ptr(mean), ptr(invstd), ptr(scale), ptr(shift), npix, C_, ptr(rows_ref), dt, st), "bn_bwd_reduce")
    dx = torch.full_like(raw, float("nan"))
    rows = torch.full((nrows, 2, C_), float("nan"), device=dev)
    check(lib.gdrn_upsample2x_bwd_bnsums(ptr(d_u), ptr(dx), ptr(raw), ptr(mean), ptr(invstd), ptr(scale), ptr(shift), N, Hh, Hh, C_, ptr(rows), dt, st), "upsample_bwd_bnsums")
    torch.cuda.synchronize()
    assert torch.equal(dx, dx_ref)
    assert torch.equal(rows, rows_ref), H.rel(rows, rows_ref)


@pytest.mark.parametrize("case", [(2, 64, 64), (3, 16, 32), (1, 8, 16), (2, 24, 48)])
def test_block64_eval_equals_two_halo_launches_bit_for_bit(H, case):
    """r6: one 64-channel BasicBlock in eval mode as ONE launch (gdrn_block64_eval: conv1 on the 10 x 18 halo'd region, the intermediate in
    LDS, conv2, + identity) == conv1 (bias, ReLU) -> conv2 (bias, + identity, ReLU) on the halo kernel, BIT FOR BIT (same accumulation order,
    same roundings) -- incl. tiles on the image border (the intermediate is zero outside the image), a one-tile image and a non-square tiling;
    and against torch on the rounded operands (resnet_backbone.py:69-80 under model.eval(): torchvision BasicBlock with folded BatchNorms)."""
    lib = cabi.load(BF16)
    dt = BF16
    B, Hh, Ww = case
    x = H.rounded(H.randn(400, B, 64, Hh, Ww), dt)
    w1 = H.rounded(H.randn(401, 64, 64, 3, 3) / 24, dt)
    w2 = H.rounded(H.randn(402, 64, 64, 3, 3) / 24, dt)
    b1, b2 = H.randn(403, 64) * 0.3, H.randn(404, 64) * 0.3
    xd = H.nhwc(x, dt)
    wp1, wp2 = H.pack_fwd(w1, dt), H.pack_fwd(w2, dt)
    b1d, b2d = b1.to(H.DEV), b2.to(H.DEV)
    # two launches: first halo kernel, four waves (the 64-channel tile has no other form)
    def halo(xin, wp, bias, addend):
        cp_y, _ = H.conv_gemm(xin, wp, B, Hh, Ww, 64, 64, Hh, Ww, 64, 3, 3, 1, 1, dt, bias=bias, addend=addend, act=1, halo=True)
        return cp_y
    if Ww == Hh:   # (the helper takes square maps' geometry as Hi, Wi separately: fine for all cases)
        pass
    a1 = halo(xd, wp1, b1d, None)
    y_ref = halo(a1, wp2, b2d, xd)
    wf1, wf2 = torch.empty_like(wp1), torch.empty_like(wp2)
    check(lib.gdrn_pack_wfrag(ptr(wp1), ptr(wf1), 64, 64, dt, H.stream()), "pack_wfrag")
    check(lib.gdrn_pack_wfrag(ptr(wp2), ptr(wf2), 64, 64, dt, H.stream()), "pack_wfrag")
    y = torch.full_like(xd, float("nan"))
    assert lib.gdrn_block64_eval_ok(B, Hh, Ww, dt) == 1
    check(lib.gdrn_block64_eval(ptr(xd), ptr(wf1), ptr(b1d), ptr(wf2), ptr(b2d), ptr(y), B, Hh, Ww, dt, H.stream()), "block64_eval")
    torch.cuda.synchronize()
    assert torch.equal(y, y_ref), H.rel(y, y_ref)
    ref = F.relu(F.conv2d(H.rounded(F.relu(F.conv2d(x, w1, b1, 1, 1)), dt), w2, b2, 1, 1) + x)
    assert H.rel(H.nchw(y, 64), ref) < TOL[dt]
    assert lib.gdrn_block64_eval_ok(B, 12, 16, dt) == 0 and lib.gdrn_block64_eval(ptr(xd), ptr(wf1), ptr(b1d), ptr(wf2), ptr(b2d), ptr(xd), B, Hh, Ww, dt, H.stream()) == -1


@pytest.mark.parametrize("case", [(2, 64, 128, 32, True), (3, 128, 256, 16, True), (2, 128, 128, 32, False), (1, 128, 128, 16, False), (2, 256, 512, 16, True),
                                  (4, 256, 512, 8, True), (2, 128, 128, 8, False), (6, 64, 128, 8, True)])
def test_conv3x3_stride2_parity_plane_kernel(H, case):
    """r6: gdrn_conv3x3s2 -- 3x3 stride-2 pad-1 forward conv on the parity-plane halo kernel, with the block's 1x1 stride-2 shortcut conv in the
    same launch (ds), per-tile BatchNorm-statistics rows for both, and the eval-mode epilogue (bias, ReLU on the main conv only) -- against
    torch on the rounded operands (resnet_backbone.py:69-80 stage-entry blocks; conv_pnp_net.py:76-92).  case = (B, Cin, Cout, Hout, ds).
    Hout = 8: the two-images-per-tile form (layer4.0, Patch-PnP's third conv)."""
    from gdrnet_amd.cabi import S2Params

    lib = cabi.load(BF16)
    dt, dev = BF16, H.DEV
    B, I, O, Ho, ds = case
    Hi = 2 * Ho
    x = H.rounded(H.randn(500, B, I, Hi, Hi), dt)
    w = H.rounded(H.randn(501, O, I, 3, 3) / math.sqrt(I * 9), dt)
    wd = H.rounded(H.randn(502, O, I, 1, 1) / math.sqrt(I), dt)
    xd = H.nhwc(x, dt)
    wp = H.pack_fwd(w, dt)
    wf = torch.empty_like(wp)
    check(lib.gdrn_pack_wfrag(ptr(wp), ptr(wf), wp.shape[0], I, dt, H.stream()), "pack_wfrag")
    wdp = H.pack_fwd(wd, dt)          # [rows][1][Cin] row-major
    ref, refd = F.conv2d(x, w, None, 2, 1), F.conv2d(x, wd, None, 2, 0)
    for mode in ("train", "eval"):
        sp = S2Params()
        y = torch.full((B, Ho, Ho, O), float("nan"), dtype=xd.dtype, device=dev)
        yd = torch.full((B, Ho, Ho, O), float("nan"), dtype=xd.dtype, device=dev)
        sp.x, sp.w, sp.y = ptr(xd), ptr(wf), ptr(y)
        sp.Hi = sp.Wi = Hi
        sp.Ho = sp.Wo = Ho
        sp.Cin, sp.x_cs, sp.Cout, sp.y_cs, sp.yd_cs = I, I, O, O, O
        sp.N, sp.w_rows, sp.wd_rows, sp.dtype = B, wp.shape[0], wdp.shape[0], dt
        if ds:
            sp.wd, sp.yd = ptr(wdp), ptr(yd)
        assert lib.gdrn_conv3x3s2_ok(C.byref(sp)) == 1
        rows = lib.gdrn_conv3x3s2_stats_rows(C.byref(sp))
        assert rows == (B * (Ho // 4) * (Ho // 16) if Ho % 16 == 0 else (B // 2) * (Ho // 4))
        st1 = torch.full((rows, 2, O), float("nan"), device=dev)
        st2 = torch.full((rows, 2, O), float("nan"), device=dev)
        bias, bias_d = H.randn(503, O).to(dev), H.randn(504, O).to(dev)
        if mode == "train":
            sp.stats = ptr(st1)
            if ds:
                sp.stats_d = ptr(st2)
        else:
            sp.bias, sp.act = ptr(bias), 1
            if ds:
                sp.bias_d = ptr(bias_d)
        check(lib.gdrn_conv3x3s2(C.byref(sp), H.stream()), "conv3x3s2")
        torch.cuda.synchronize()
        if mode == "train":
            assert H.rel(H.nchw(y, O), ref) < TOL[dt]
            s_ = st1.sum(0).cpu()
            assert H.rel(s_[0], ref.sum((0, 2, 3))) < 1e-3 + TOL[dt] and H.rel(s_[1], (ref ** 2).sum((0, 2, 3))) < 1e-3
            if ds:
                assert H.rel(H.nchw(yd, O), refd) < TOL[dt]
                s_ = st2.sum(0).cpu()
                assert H.rel(s_[0], refd.sum((0, 2, 3))) < 1e-3 + TOL[dt] and H.rel(s_[1], (refd ** 2).sum((0, 2, 3))) < 1e-3
        else:
            assert H.rel(H.nchw(y, O), F.relu(ref + bias.cpu().view(1, -1, 1, 1))) < TOL[dt]
            if ds:
                assert H.rel(H.nchw(yd, O), refd + bias_d.cpu().view(1, -1, 1, 1)) < TOL[dt]
    sp.Wo = sp.Ho = 8
    sp.Hi = sp.Wi = 16
    sp.N = 3
    assert lib.gdrn_conv3x3s2_ok(C.byref(sp)) == 0   # 8-wide maps go two images to a tile: an odd image count stays on the generic kernel
    sp.N, sp.Wo, sp.Ho, sp.Hi, sp.Wi = 2, 4, 4, 8, 8
    assert lib.gdrn_conv3x3s2_ok(C.byref(sp)) == 0   # ... and so do maps narrower than 8 pixels


@pytest.mark.parametrize("case", [(2, 64, 128, 32, True, True), (3, 128, 256, 16, True, True), (2, 128, 128, 32, False, False), (1, 128, 128, 16, False, False),
                                  (2, 64, 128, 16, False, True), (2, 128, 256, 32, True, False),
                                  (4, 256, 512, 8, True, True), (2, 128, 128, 8, False, False), (6, 64, 128, 8, True, False), (2, 128, 64, 8, False, True)])
def test_conv3x3_stride2_dgrad_parity_class_kernel(H, case):
    """r6: gdrn_conv3x3s2_dgrad -- data gradient of the 3x3 stride-2 conv (four parity classes of input pixels, 1 / 2 / 2 / 4 taps), with the 1x1
    stride-2 shortcut's data gradient added in the same launch (ds) and the ReLU mask + BatchNorm-backward sums of the BatchNorm whose output
    the gradient belongs to in the epilogue (bnb) -- against autograd on the rounded operands.  case = (B, Cin, Cout, Hout, ds, bnb).
    Hout = 8: the two-images-per-tile form (layer4.0, Patch-PnP's third conv)."""
    from gdrnet_amd.cabi import S2dParams

    lib = cabi.load(BF16)
    dt, dev = BF16, H.DEV
    B, I, O, Ho, ds, bnb = case
    Hi = 2 * Ho
    x = H.rounded(H.randn(600, B, I, Hi, Hi), dt).requires_grad_(True)
    w = H.rounded(H.randn(601, O, I, 3, 3) / math.sqrt(O * 9), dt)
    wd = H.rounded(H.randn(602, O, I, 1, 1) / math.sqrt(O), dt)
    dy = H.rounded(H.randn(603, B, O, Ho, Ho), dt)
    dyd = H.rounded(H.randn(604, B, O, Ho, Ho), dt)
    tot = (F.conv2d(x, w, None, 2, 1) * dy).sum() + ((F.conv2d(x, wd, None, 2, 0) * dyd).sum() if ds else 0.0)
    tot.backward()
    ref = x.grad
    wdp = H.pack_dgrad(w, dt, flip=0)            # [rows >= I][9][O]
    wdf = torch.empty_like(wdp)
    check(lib.gdrn_pack_wfrag(ptr(wdp), ptr(wdf), wdp.shape[0], O, dt, H.stream()), "pack_wfrag")
    wddp = H.pack_dgrad(wd, dt, flip=0)          # [rows >= I][1][O] row-major
    sp = S2dParams()
    dyn, dydn = H.nhwc(dy, dt), H.nhwc(dyd, dt)
    dx = torch.full((B, Hi, Hi, I), float("nan"), dtype=dyn.dtype, device=dev)
    sp.dy, sp.w, sp.dx = ptr(dyn), ptr(wdf), ptr(dx)
    sp.Hi = sp.Wi = Hi
    sp.Ho = sp.Wo = Ho
    sp.Cin, sp.dx_cs, sp.Cout, sp.dy_cs, sp.dyd_cs = I, I, O, O, O
    sp.N, sp.w_rows, sp.wdd_rows, sp.dtype = B, wdp.shape[0], wddp.shape[0], dt
    if ds:
        sp.dyd, sp.wdd = ptr(dydn), ptr(wddp)
    if bnb:
        raw = H.rounded(H.randn(605, B, I, Hi, Hi), dt)
        act = H.rounded(H.randn(606, B, I, Hi, Hi), dt)      # the stored activation whose sign is the ReLU mask
        mean, invstd = H.randn(607, I) * 0.2, 0.5 + torch.rand(I, generator=torch.Generator().manual_seed(9))
        rawn, actn, mean_d, invstd_d = H.nhwc(raw, dt), H.nhwc(act, dt), mean.to(dev), invstd.to(dev)
        assert lib.gdrn_conv3x3s2_dgrad_ok(C.byref(sp)) == 1
        nrows = lib.gdrn_conv3x3s2_dgrad_rows(C.byref(sp))
        rows = torch.full((nrows, 2, I), float("nan"), device=dev)
        sp.bnb_x, sp.bnb_mask, sp.bnb_mean, sp.bnb_invstd, sp.bnb_rows, sp.bnb_cs = ptr(rawn), ptr(actn), ptr(mean_d), ptr(invstd_d), ptr(rows), I
    assert lib.gdrn_conv3x3s2_dgrad_ok(C.byref(sp)) == 1
    check(lib.gdrn_conv3x3s2_dgrad(C.byref(sp), H.stream()), "conv3x3s2_dgrad")
    torch.cuda.synchronize()
    got = H.nchw(dx, I)
    if bnb:
        gm = ref * (act > 0)
        assert H.rel(got, gm) < TOL[dt]
        gq = H.rounded(gm, dt) if False else gm
        s_ = rows.sum(0).cpu()
        assert H.rel(s_[0], gm.sum((0, 2, 3))) < 2e-3 + TOL[dt]
        xh = (raw - mean.view(1, -1, 1, 1)) * invstd.view(1, -1, 1, 1)
        assert H.rel(s_[1], (gm * xh).sum((0, 2, 3))) < 2e-3 + TOL[dt]
    else:
        assert H.rel(got, ref) < TOL[dt]
    sp.N, sp.Ho, sp.Wo, sp.Hi, sp.Wi = 3, 8, 8, 16, 16
    assert lib.gdrn_conv3x3s2_dgrad_ok(C.byref(sp)) == 0   # 8-wide maps: two images per tile, an odd image count is not covered


@pytest.mark.parametrize("case", [(4, 256, 512, 8), (2, 64, 128, 16), (2, 128, 128, 8)])
def test_conv3x3_stride2_kernel_with_the_batchnorm_backward_epilogue(H, case):
    """r6 (ABI 5): gdrn_conv3x3s2 as a data gradient -- the head's ConvTranspose2d backward is a stride-2 conv of the output gradient -- w.r.t. a
    BatchNorm(+ReLU)'s output: ReLU mask from the stored activation and the BatchNorm-backward sums in the epilogue (gdrn_s2_params.bnb_*),
    against torch on the rounded operands.  case = (B, Cin, Cout, Hout)."""
    from gdrnet_amd.cabi import S2Params

    lib = cabi.load(BF16)
    dt, dev = BF16, H.DEV
    B, I, O, Ho = case
    Hi = 2 * Ho
    x = H.rounded(H.randn(800, B, I, Hi, Hi), dt)
    w = H.rounded(H.randn(801, O, I, 3, 3) / math.sqrt(I * 9), dt)
    ref = F.conv2d(x, w, None, 2, 1)
    raw = H.rounded(H.randn(802, B, O, Ho, Ho), dt)
    act = H.rounded(H.randn(803, B, O, Ho, Ho), dt)
    mean, invstd = H.randn(804, O) * 0.2, 0.5 + torch.rand(O, generator=torch.Generator().manual_seed(11))
    xd, rawn, actn, mean_d, invstd_d = H.nhwc(x, dt), H.nhwc(raw, dt), H.nhwc(act, dt), mean.to(dev), invstd.to(dev)
    wp = H.pack_fwd(w, dt)
    wf = torch.empty_like(wp)
    check(lib.gdrn_pack_wfrag(ptr(wp), ptr(wf), wp.shape[0], I, dt, H.stream()), "pack_wfrag")
    sp = S2Params()
    y = torch.full((B, Ho, Ho, O), float("nan"), dtype=xd.dtype, device=dev)
    sp.x, sp.w, sp.y = ptr(xd), ptr(wf), ptr(y)
    sp.Hi = sp.Wi = Hi
    sp.Ho = sp.Wo = Ho
    sp.Cin, sp.x_cs, sp.Cout, sp.y_cs = I, I, O, O
    sp.N, sp.w_rows, sp.dtype = B, wp.shape[0], dt
    nrows = lib.gdrn_conv3x3s2_stats_rows(C.byref(sp))
    rows = torch.full((nrows, 2, O), float("nan"), device=dev)
    sp.bnb_x, sp.bnb_mask, sp.bnb_mean, sp.bnb_invstd, sp.bnb_rows, sp.bnb_cs = ptr(rawn), ptr(actn), ptr(mean_d), ptr(invstd_d), ptr(rows), O
    assert lib.gdrn_conv3x3s2_ok(C.byref(sp)) == 1
    check(lib.gdrn_conv3x3s2(C.byref(sp), H.stream()), "conv3x3s2 (bnb)")
    torch.cuda.synchronize()
    gm = ref * (act > 0)
    assert H.rel(H.nchw(y, O), gm) < TOL[dt]
    s_ = rows.sum(0).cpu()
    assert H.rel(s_[0], gm.sum((0, 2, 3))) < 2e-3 + TOL[dt]
    xh = (raw - mean.view(1, -1, 1, 1)) * invstd.view(1, -1, 1, 1)
    assert H.rel(s_[1], (gm * xh).sum((0, 2, 3))) < 2e-3 + TOL[dt]
    sp.act = 1
    assert lib.gdrn_conv3x3s2_ok(C.byref(sp)) == 0   # the BatchNorm-backward epilogue does not combine with the forward one


@pytest.mark.parametrize("case", [(4, 512, 256, 8), (2, 128, 64, 16), (2, 64, 128, 8)])
def test_conv_transpose_forward_on_the_stride2_dgrad_kernel(H, case):
    """r6 (ABI 5): gdrn_conv3x3s2_dgrad with the forward epilogue = the forward pass of nn.ConvTranspose2d(Cin, Cout, 3, stride 2, padding 1,
    output_padding 1) (cdpn_rot_head_region.py:96-101): train mode -- raw output + per-tile BatchNorm statistics rows; eval mode -- bias + ReLU.
    case = (B, Cin, Cout, Hin)."""
    from gdrnet_amd.cabi import S2dParams

    lib = cabi.load(BF16)
    dt, dev = BF16, H.DEV
    B, I, O, Hin = case
    Hout = 2 * Hin
    x = H.rounded(H.randn(700, B, I, Hin, Hin), dt)
    w = H.rounded(H.randn(701, I, O, 3, 3) / math.sqrt(I * 2.25), dt)     # ConvTranspose2d weight [Cin][Cout][3][3]
    ref = F.conv_transpose2d(x, w, None, 2, 1, 1)
    # forward operand [rows = Cout][9 taps, not flipped][Cin] (Engine._pack_args, kind "convT", which "f"), fragment-major
    rows = O if O <= 64 else (O + 127) // 128 * 128
    wp = torch.zeros(rows, 9, I)
    wp[:O] = w.permute(1, 2, 3, 0).reshape(O, 9, I)
    wp = wp.to(dev).to(H.tdt(dt)).contiguous()
    wf = torch.empty_like(wp)
    check(lib.gdrn_pack_wfrag(ptr(wp), ptr(wf), rows, I, dt, H.stream()), "pack_wfrag")
    xn = H.nhwc(x, dt)
    for mode in ("train", "eval"):
        sp = S2dParams()
        y = torch.full((B, Hout, Hout, O), float("nan"), dtype=xn.dtype, device=dev)
        sp.dy, sp.w, sp.dx = ptr(xn), ptr(wf), ptr(y)
        sp.Hi = sp.Wi = Hout
        sp.Ho = sp.Wo = Hin
        sp.Cin, sp.dx_cs, sp.Cout, sp.dy_cs = O, O, I, I
        sp.N, sp.w_rows, sp.dtype = B, rows, dt
        assert lib.gdrn_conv3x3s2_dgrad_ok(C.byref(sp)) == 1
        nrows = lib.gdrn_conv3x3s2_dgrad_rows(C.byref(sp))
        st = torch.full((nrows, 2, O), float("nan"), device=dev)
        bias = H.randn(702, O).to(dev)
        if mode == "train":
            sp.stats = ptr(st)
        else:
            sp.bias, sp.act = ptr(bias), 1
        check(lib.gdrn_conv3x3s2_dgrad(C.byref(sp), H.stream()), "conv3x3s2_dgrad (forward)")
        torch.cuda.synchronize()
        if mode == "train":
            assert H.rel(H.nchw(y, O), ref) < TOL[dt]
            s_ = st.sum(0).cpu()
            assert H.rel(s_[0], ref.sum((0, 2, 3))) < 1e-3 + TOL[dt] and H.rel(s_[1], (ref ** 2).sum((0, 2, 3))) < 1e-3
        else:
            assert H.rel(H.nchw(y, O), F.relu(ref + bias.cpu().view(1, -1, 1, 1))) < TOL[dt]
    sp.bnb_x = ptr(y)
    assert lib.gdrn_conv3x3s2_dgrad_ok(C.byref(sp)) == 0   # the forward epilogue does not combine with the BatchNorm-backward one


# ---------------------------------------------------------------------------------------------- BatchNorm
@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("C_", [64, 256, 512])
def test_batchnorm_train_fwd_bwd(H, dt, C_):
    lib = cabi.load(BF16)
    B, Hh = 3, 12
    x = H.rounded(H.randn(50, B, C_, Hh, Hh) * 2 + 0.5, dt).requires_grad_(True)
    res = H.rounded(H.randn(51, B, C_, Hh, Hh), dt)
    gam = (0.5 + torch.rand(C_, generator=torch.Generator().manual_seed(1))).requires_grad_(True)
    bet = (torch.rand(C_, generator=torch.Generator().manual_seed(2)) - 0.5).requires_grad_(True)
    rm, rv = torch.zeros(C_), torch.ones(C_)
    yref = F.relu(F.batch_norm(x, rm, rv, gam, bet, True, 0.1, 1e-5) + res)
    dy = H.rounded(H.randn(52, B, C_, Hh, Hh), dt)
    yref.backward(dy)
    npix = B * Hh * Hh
    xd, resd, dyd = H.nhwc(x.detach(), dt), H.nhwc(res, dt), H.nhwc(dy, dt)
    # partial stats as the conv epilogue would emit them (2 row tiles)
    xf = xd.float().view(npix, C_)
    half = npix // 2
    part = torch.stack([torch.stack([xf[:half].sum(0), (xf[:half] ** 2).sum(0)]), torch.stack([xf[half:].sum(0), (xf[half:] ** 2).sum(0)])]).contiguous()
    dev = H.DEV
    mk = lambda: torch.zeros(C_, device=dev)
    mean, invstd, scale, shift = mk(), mk(), mk(), mk()
    rmd, rvd, nbt = torch.zeros(C_, device=dev), torch.ones(C_, device=dev), torch.zeros((), dtype=torch.int64, device=dev)
    gam_d, bet_d = gam.detach().to(dev), bet.detach().to(dev)  # keep alive: the launches are asynchronous
    st = H.stream()
    check(lib.gdrn_bn_finalize(ptr(part), 2, C_, float(npix), ptr(gam_d), ptr(bet_d), ptr(rmd), ptr(rvd), ptr(nbt),
                               0.1, 1e-5, ptr(mean), ptr(invstd), ptr(scale), ptr(shift), None, st), "bn_finalize")
    y = torch.empty_like(xd)
    check(lib.gdrn_bn_apply(ptr(xd), ptr(scale), ptr(shift), ptr(resd), ptr(y), npix, C_, 1, dt, st), "bn_apply")
    assert H.rel(H.nchw(y), yref) < TOL[dt]
    assert H.rel(rmd, rm) < 1e-5 and H.rel(rvd, rv) < 1e-5 and int(nbt) == 1
    nrows = lib.gdrn_bn_bwd_reduce_rows(npix, C_, dt)
    assert 1 <= nrows <= 1024
    rows = torch.full((nrows, 2, C_), float("nan"), device=dev)  # every row is written (no pre-zeroing, no atomics)
    dx, gout = torch.empty_like(xd), torch.empty_like(xd)
    dg, db, ka, kb, kc = mk(), mk(), mk(), mk(), mk()
    yd = H.nhwc(yref.detach(), dt)  # mask source: the stored activation
    check(lib.gdrn_bn_bwd_reduce(ptr(dyd), ptr(yd), ptr(xd), ptr(mean), ptr(invstd), None, None, npix, C_, ptr(rows), dt, st), "bn_bwd_reduce")
    check(lib.gdrn_bn_bwd_coef(ptr(rows), nrows, C_, npix, ptr(gam_d), ptr(mean), ptr(invstd), ptr(ka), ptr(kb), ptr(kc), ptr(dg), ptr(db), st), "bn_bwd_coef")
    check(lib.gdrn_bn_bwd_apply(ptr(dyd), ptr(yd), ptr(xd), ptr(ka), ptr(kb), ptr(kc), None, None, npix, C_, ptr(dx), ptr(gout), dt, st), "bn_bwd_apply")
    rows2 = torch.full((nrows, 2, C_), float("nan"), device=dev)
    check(lib.gdrn_bn_bwd_reduce(ptr(dyd), ptr(yd), ptr(xd), ptr(mean), ptr(invstd), None, None, npix, C_, ptr(rows2), dt, st), "bn_bwd_reduce")
    assert torch.equal(rows, rows2)  # deterministic
    tol = 2e-4 if dt == F32 else 1e-2
    assert H.rel(H.nchw(dx), x.grad) < tol
    assert H.rel(dg, gam.grad) < tol and H.rel(db, bet.grad) < tol
    assert H.rel(H.nchw(gout), dy * (yref > 0)) < TOL[dt]
    # eval-mode scale/shift
    check(lib.gdrn_bn_eval_params(ptr(gam_d), ptr(bet_d), ptr(rmd), ptr(rvd), 1e-5, C_, ptr(scale), ptr(shift), st), "bn_eval")
    check(lib.gdrn_bn_apply(ptr(xd), ptr(scale), ptr(shift), None, ptr(y), npix, C_, 0, dt, st), "bn_apply")
    assert H.rel(H.nchw(y), F.batch_norm(x.detach(), rm, rv, gam.detach(), bet.detach(), False, 0.1, 1e-5)) < TOL[dt]


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("C_,B,Hh", [(64, 3, 12), (512, 2, 8), (256, 5, 16)])
def test_batchnorm_bwd_affine_mask(H, dt, C_, B, Hh):
    """BN -> ReLU (no residual) backward with the ReLU mask recomputed from x*scale+shift == autograd."""
    lib = cabi.load(BF16)
    dev = H.DEV
    x = H.rounded(H.randn(150, B, C_, Hh, Hh) * 2 + 0.5, dt).requires_grad_(True)
    gam = (0.5 + torch.rand(C_, generator=torch.Generator().manual_seed(1))).requires_grad_(True)
    bet = (torch.rand(C_, generator=torch.Generator().manual_seed(2)) - 0.5).requires_grad_(True)
    yref = F.relu(F.batch_norm(x, None, None, gam, bet, True, 0.1, 1e-5))
    dy = H.rounded(H.randn(152, B, C_, Hh, Hh), dt)
    yref.backward(dy)
    npix = B * Hh * Hh
    xd, dyd = H.nhwc(x.detach(), dt), H.nhwc(dy, dt)
    xf = xd.float().view(npix, C_)
    part = torch.stack([xf.sum(0), (xf ** 2).sum(0)]).unsqueeze(0).contiguous()
    mk = lambda: torch.zeros(C_, device=dev)
    mean, invstd, scale, shift, dg, db = mk(), mk(), mk(), mk(), mk(), mk()
    gam_d, bet_d = gam.detach().to(dev), bet.detach().to(dev)
    st = H.stream()
    check(lib.gdrn_bn_finalize(ptr(part), 1, C_, float(npix), ptr(gam_d), ptr(bet_d), None, None, None, 0.1, 1e-5, ptr(mean),
                               ptr(invstd), ptr(scale), ptr(shift), None, st), "bn_finalize")
    nrows = lib.gdrn_bn_bwd_reduce_rows(npix, C_, dt)
    rows = torch.full((nrows, 2, C_), float("nan"), device=dev)
    ka, kb, kc = mk(), mk(), mk()
    dx, gout = torch.empty_like(xd), torch.empty_like(xd)
    check(lib.gdrn_bn_bwd_reduce(ptr(dyd), None, ptr(xd), ptr(mean), ptr(invstd), ptr(scale), ptr(shift), npix, C_, ptr(rows), dt, st),
          "bn_bwd_reduce")
    check(lib.gdrn_bn_bwd_coef(ptr(rows), nrows, C_, npix, ptr(gam_d), ptr(mean), ptr(invstd), ptr(ka), ptr(kb), ptr(kc), ptr(dg), ptr(db), st), "bn_bwd_coef")
    check(lib.gdrn_bn_bwd_apply(ptr(dyd), None, ptr(xd), ptr(ka), ptr(kb), ptr(kc), ptr(scale), ptr(shift), npix, C_, ptr(dx), ptr(gout), dt, st),
          "bn_bwd_apply")
    torch.cuda.synchronize()
    tol = 2e-4 if dt == F32 else 1e-2
    assert H.rel(H.nchw(dx), x.grad) < tol
    assert H.rel(dg, gam.grad) < tol and H.rel(db, bet.grad) < tol
    assert H.rel(H.nchw(gout), dy * (yref > 0)) < TOL[dt]


@pytest.mark.parametrize("dt", DTS)
def test_bn_relu_maxpool(H, dt):
    lib = cabi.load(BF16)
    B, C_, Hh = 2, 64, 24
    x = H.rounded(H.randn(60, B, C_, Hh, Hh), dt).requires_grad_(True)
    scale, shift = 0.5 + torch.rand(C_), torch.rand(C_) - 0.5
    ref = F.max_pool2d(F.relu(x * scale[None, :, None, None] + shift[None, :, None, None]), 3, 2, 1)
    dy = H.rounded(H.randn(61, B, C_, Hh // 2, Hh // 2), dt)
    ref.backward(dy)
    xd, dyd = H.nhwc(x.detach(), dt), H.nhwc(dy, dt)
    y = torch.empty(B, Hh // 2, Hh // 2, C_, dtype=H.tdt(dt), device=H.DEV)
    idx = torch.empty(B, Hh // 2, Hh // 2, C_, dtype=torch.uint8, device=H.DEV)
    sc, sh = scale.to(H.DEV), shift.to(H.DEV)
    check(lib.gdrn_bn_relu_maxpool_fwd(ptr(xd), ptr(sc), ptr(sh), ptr(y), ptr(idx), B, Hh, Hh, C_, dt, H.stream()), "pool_fwd")
    assert H.rel(H.nchw(y), ref) < TOL[dt]
    g = torch.empty_like(xd)
    check(lib.gdrn_maxpool_bwd(ptr(dyd), ptr(idx), ptr(xd), ptr(sc), ptr(sh), ptr(g), B, Hh, Hh, C_, None, None, None, dt, H.stream()), "pool_bwd")
    # g = grad wrt the BN output (pre-ReLU) = x.grad / scale
    assert H.rel(H.nchw(g) * scale[None, :, None, None], x.grad) < TOL[dt]
    # with the fused BatchNorm-backward sums: same g, rows[r] sum to (sum g, sum g*xhat) of the stored g
    mean, invstd = (torch.rand(C_) - 0.5).to(H.DEV), (0.5 + torch.rand(C_)).to(H.DEV)
    nrows = lib.gdrn_maxpool_bwd_rows(B, Hh, Hh, C_, dt)
    rows = torch.full((nrows, 2, C_), float("nan"), device=H.DEV)
    g2 = torch.empty_like(xd)
    check(lib.gdrn_maxpool_bwd(ptr(dyd), ptr(idx), ptr(xd), ptr(sc), ptr(sh), ptr(g2), B, Hh, Hh, C_, ptr(mean), ptr(invstd), ptr(rows), dt, H.stream()), "pool_bwd")
    torch.cuda.synchronize()
    assert torch.equal(g2, g)
    gf, xf = g.float().view(-1, C_), xd.float().view(-1, C_)
    tot = rows.sum(0)
    assert H.rel(tot[0], gf.sum(0)) < 1e-4 and H.rel(tot[1], (gf * (xf - mean) * invstd).sum(0)) < 1e-4


@pytest.mark.parametrize("dt", DTS)
def test_upsample2x(H, dt):
    lib = cabi.load(BF16)
    B, C_, Hh = 2, 256, 16
    x = H.rounded(H.randn(70, B, C_, Hh, Hh), dt).requires_grad_(True)
    ref = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    dy = H.rounded(H.randn(71, B, C_, 2 * Hh, 2 * Hh), dt)
    ref.backward(dy)
    xd, dyd = H.nhwc(x.detach(), dt), H.nhwc(dy, dt)
    y = torch.empty(B, 2 * Hh, 2 * Hh, C_, dtype=H.tdt(dt), device=H.DEV)
    check(lib.gdrn_upsample2x_fwd(ptr(xd), ptr(y), B, Hh, Hh, C_, dt, H.stream()), "up_fwd")
    assert H.rel(H.nchw(y), ref) < TOL[dt]
    dx = torch.empty_like(xd)
    check(lib.gdrn_upsample2x_bwd(ptr(dyd), ptr(dx), B, Hh, Hh, C_, dt, H.stream()), "up_bwd")
    assert H.rel(H.nchw(dx), x.grad) < TOL[dt]


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("Hh", [32, 8, 40])   # (40: more rows than a thread keeps in registers between the passes -- the streaming path)
def test_groupnorm_relu(H, dt, Hh):
    lib = cabi.load(BF16)
    B, C_ = 3, 128
    x = H.rounded(H.randn(80, B, C_, Hh, Hh) * 1.5 + 0.3, dt).requires_grad_(True)
    gam = (0.5 + torch.rand(C_)).requires_grad_(True)
    bet = (torch.rand(C_) - 0.5).requires_grad_(True)
    ref = F.relu(F.group_norm(x, 32, gam, bet, 1e-5))
    dy = H.rounded(H.randn(81, B, C_, Hh, Hh), dt)
    ref.backward(dy)
    dev = H.DEV
    xd, dyd = H.nhwc(x.detach(), dt), H.nhwc(dy, dt)
    y = torch.empty_like(xd)
    mr = torch.zeros(B, 32, 2, device=dev)
    g, b = gam.detach().to(dev), bet.detach().to(dev)
    check(lib.gdrn_gn_relu_fwd(ptr(xd), ptr(g), ptr(b), ptr(y), ptr(mr), B, Hh * Hh, C_, 32, 1e-5, dt, H.stream()), "gn_fwd")
    assert H.rel(H.nchw(y), ref) < TOL[dt]
    dx = torch.empty_like(xd)
    dg, db = torch.zeros(C_, device=dev), torch.zeros(C_, device=dev)
    yd = H.nhwc(ref.detach(), dt)
    check(lib.gdrn_gn_relu_bwd(ptr(dyd), ptr(yd), ptr(xd), ptr(g), ptr(mr), ptr(dx), ptr(dg), ptr(db), B, Hh * Hh, C_, 32, dt, H.stream()), "gn_bwd")
    tol = 2e-4 if dt == F32 else 1e-2
    assert H.rel(H.nchw(dx), x.grad) < tol
    assert H.rel(dg, gam.grad) < tol and H.rel(db, bet.grad) < tol


@pytest.mark.parametrize("dt", DTS)
def test_leaky_and_bias_grad(H, dt):
    lib = cabi.load(BF16)
    B, C_ = 4, 1024
    y = H.rounded(H.randn(90, B, C_), dt)
    dy = H.rounded(H.randn(91, B, C_), dt)
    yd, dyd = y.to(H.DEV).to(H.tdt(dt)), dy.to(H.DEV).to(H.tdt(dt))
    dx = torch.empty_like(yd)
    check(lib.gdrn_leaky_bwd(ptr(dyd), ptr(yd), ptr(dx), B * C_, dt, H.stream()), "leaky_bwd")
    assert H.rel(dx, dy * torch.where(y > 0, 1.0, 0.1)) < TOL[dt]
    db = torch.zeros(C_, device=H.DEV)
    check(lib.gdrn_bias_grad(ptr(dyd), C_, B, C_, ptr(db), dt, H.stream()), "bias_grad")
    assert H.rel(db, dy.sum(0)) < 1e-5
    # strided / partially valid columns, many rows (head output conv bias)
    R = 5000
    d2 = H.rounded(H.randn(92, R, 128), dt)
    d2d = d2.to(H.DEV).to(H.tdt(dt))
    db2 = torch.zeros(69, device=H.DEV)
    check(lib.gdrn_bias_grad(ptr(d2d), 128, R, 69, ptr(db2), dt, H.stream()), "bias_grad")
    assert H.rel(db2, d2[:, :69].sum(0)) < 1e-4


# ---------------------------------------------------------------------------------------------- head tail + map losses
@pytest.mark.parametrize("dt", DTS)
def test_head_tail_and_map_losses(H, dt):
    """GDRN.py:156-169 glue + conv_pnp_net.py:121-125 + GDRN.py:345-400, forward and backward, against autograd."""
    from gdrnet_amd import synth
    from oracle import gdrn_oracle as O

    lib = cabi.load(BF16)
    B, HW, nreg, hs = 2, 4096, 64, 72
    M = B * HW
    b = synth.make_batch(B, seed=9)
    b["roi_mask_visib"][1, :8] = 0
    head = (H.randn(100, B, 69, 64, 64) * 1.5).requires_grad_(True)
    mask, cx, cy, cz, region = head[:, :1], head[:, 1:2], head[:, 2:3], head[:, 3:4], head[:, 4:]
    coor_feat = torch.cat([cx, cy, cz, b["roi_coord_2d"]], 1)
    sm = torch.softmax(region[:, 1:], 1)
    xyz = (coor_feat[:, :3] - 0.5) * b["roi_extent"].view(B, 3, 1, 1)
    pnp_ref = torch.cat([xyz, coor_feat[:, 3:], sm], 1)
    d_pnp = H.rounded(H.randn(101, B, 69, 64, 64) * 0.01, dt)
    dummy_rot = torch.eye(3).expand(B, 3, 3)
    L = O.gdrn_loss(mask, cx, cy, cz, region, dummy_rot, torch.zeros(B, 3), dict(b, ego_rot=dummy_rot))
    gw = torch.tensor([1.0, 0.5, 2.0, 1.5, 0.7])
    tot = gw[0] * L["loss_coor_x"] + gw[1] * L["loss_coor_y"] + gw[2] * L["loss_coor_z"] + gw[3] * L["loss_mask"] + gw[4] * L["loss_region"]
    (tot + (pnp_ref * d_pnp).sum()).backward()

    dev, st = H.DEV, H.stream()
    head_d = torch.zeros(M, hs, device=dev)
    head_d[:, :69] = head.detach().permute(0, 2, 3, 1).reshape(M, 69).to(dev)
    f = lambda t: t.to(dev).float().contiguous()
    c2d, ext, gxyz, mv, mt = f(b["roi_coord_2d"]), f(b["roi_extent"]), f(b["roi_xyz"]), f(b["roi_mask_visib"]), f(b["roi_mask_trunc"])
    greg = b["roi_region"].to(dev).contiguous()
    pnp = torch.full((M, 128), float("nan"), dtype=H.tdt(dt), device=dev)
    check(lib.gdrn_head_tail_fwd(ptr(head_d), hs, ptr(c2d), ptr(ext), ptr(pnp), 128, B, HW, nreg, dt, st), "head_tail_fwd")
    got = pnp.float().cpu().view(B, 64, 64, 128)
    assert H.rel(got[..., :69].permute(0, 3, 1, 2), pnp_ref.detach()) < TOL[dt]
    assert float(got[..., 69:].abs().max()) == 0.0
    acc = torch.zeros(8, dtype=torch.float64, device=dev)
    losses = torch.zeros(8, device=dev)
    check(lib.gdrn_map_loss_fwd(ptr(head_d), hs, ptr(gxyz), ptr(mv), ptr(mt), ptr(greg), B, HW, nreg, ptr(acc), st), "map_loss_fwd")
    check(lib.gdrn_map_loss_finalize(ptr(acc), B, HW, ptr(losses), st), "map_loss_finalize")
    ref = torch.stack([L[k] for k in ("loss_coor_x", "loss_coor_y", "loss_coor_z", "loss_mask", "loss_region")]).detach()
    np.testing.assert_allclose(losses[:5].cpu().numpy(), ref.numpy(), rtol=2e-5)
    # train-mode entry point: both of the above in one pass; with GDRN_PREZEROED the pad channels are left to the caller
    pnp2 = torch.full((M, 128), 7.0, dtype=H.tdt(dt), device=dev)
    acc2 = torch.full((8,), float("nan"), dtype=torch.float64, device=dev)
    losses2 = torch.zeros(8, device=dev)
    check(lib.gdrn_head_tail_loss_fwd(ptr(head_d), hs, ptr(c2d), ptr(ext), ptr(pnp2), 128, ptr(gxyz), ptr(mv), ptr(mt), ptr(greg), ptr(acc2), B, HW,
                                      nreg, dt | cabi.PREZEROED, st), "head_tail_loss_fwd")
    check(lib.gdrn_map_loss_finalize(ptr(acc2), B, HW, ptr(losses2), st), "map_loss_finalize")
    assert torch.equal(pnp2[:, :72], pnp[:, :72]) and float((pnp2[:, 72:].float() - 7.0).abs().max()) == 0.0
    np.testing.assert_allclose(losses2[:5].cpu().numpy(), ref.numpy(), rtol=2e-5)
    # GDRN_ACC_ROWS (what the engine calls): per-workgroup partial rows behind the totals instead of a memset + atomics; the rows are added
    # in a fixed order -> the same bits on every run, the same sums as the atomics to fp64 rounding
    nrows = lib.gdrn_head_tail_loss_rows(B, HW, nreg, hs, 128)
    assert nrows == min(M // 16, 4096)
    outs = []
    for rep in range(2):
        pnp3 = torch.full((M, 128), 7.0, dtype=H.tdt(dt), device=dev)
        acc3 = torch.full((8 + 8 * nrows,), float("nan"), dtype=torch.float64, device=dev)   # (never read before it is written)
        losses3 = torch.zeros(8, device=dev)
        check(lib.gdrn_head_tail_loss_fwd(ptr(head_d), hs, ptr(c2d), ptr(ext), ptr(pnp3), 128, ptr(gxyz), ptr(mv), ptr(mt), ptr(greg), ptr(acc3), B, HW,
                                          nreg, dt | cabi.PREZEROED | cabi.ACC_ROWS, st), "head_tail_loss_fwd rows")
        check(lib.gdrn_map_loss_finalize_rows(ptr(acc3), nrows, B, HW, ptr(losses3), st), "map_loss_finalize_rows")
        assert torch.equal(pnp3[:, :72], pnp[:, :72])
        outs.append((acc3[:8].cpu(), losses3.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    # (r6, ABI 5) gdrn_loss_finalize: the same rows + the three pose losses from per-RoI rows (added in RoI order) + the weighted loss vector
    pose_rows = torch.rand(B, 4, generator=torch.Generator().manual_seed(5)).to(dev)
    lw = (0.5 + torch.rand(8, generator=torch.Generator().manual_seed(6))).to(dev)
    losses4, weighted = torch.full((8,), float("nan"), device=dev), torch.full((8,), float("nan"), device=dev)
    check(lib.gdrn_loss_finalize(ptr(acc3), nrows, B, HW, ptr(pose_rows), ptr(losses4), ptr(lw), ptr(weighted), st), "loss_finalize")
    assert torch.equal(losses4[:5].cpu(), outs[0][1][:5])
    want = torch.zeros(3)
    for n_ in range(B):
        want += pose_rows[n_, :3].cpu()           # fp32, RoI order
    assert torch.equal(losses4[5:].cpu(), want)
    assert torch.equal(weighted.cpu(), (losses4 * lw).cpu())
    losses5 = losses4.clone()
    check(lib.gdrn_loss_finalize(ptr(acc3), nrows, B, HW, None, ptr(losses5), None, None, st), "loss_finalize (no pose rows)")
    assert torch.equal(losses5, losses4)          # losses[5..7] are left alone without pose rows
    assert lib.gdrn_loss_finalize(ptr(acc3), nrows, B, HW, None, ptr(losses5), ptr(lw), None, st) == -1   # GDRN_ERR_ARG: weights without an output
    np.testing.assert_allclose(outs[0][0][:6].numpy(), acc2[:6].cpu().numpy(), rtol=1e-12)
    assert float(outs[0][0][6:].abs().max()) == 0.0
    np.testing.assert_allclose(outs[0][1][:5].numpy(), ref.numpy(), rtol=2e-5)
    dpn = torch.zeros(M, 128, dtype=H.tdt(dt), device=dev)
    dpn[:, :69] = d_pnp.permute(0, 2, 3, 1).reshape(M, 69).to(dev).to(H.tdt(dt))
    dh = torch.full((M, 128), float("nan"), dtype=H.tdt(dt), device=dev)
    gw_d = gw.to(dev)
    check(lib.gdrn_head_tail_bwd(ptr(head_d), hs, ptr(pnp), ptr(dpn), 128, ptr(ext), ptr(gxyz), ptr(mv), ptr(mt), ptr(greg), ptr(acc),
                                 ptr(gw_d), ptr(dh), 128, B, HW, nreg, dt, st), "head_tail_bwd")
    gotd = dh.float().cpu().view(B, 64, 64, 128)
    assert H.rel(gotd[..., :69].permute(0, 3, 1, 2), head.grad) < (1e-4 if dt == F32 else 2e-2)
    assert float(gotd[..., 69:].abs().max()) == 0.0
    dh2 = torch.full((M, 128), 7.0, dtype=H.tdt(dt), device=dev)
    check(lib.gdrn_head_tail_bwd(ptr(head_d), hs, ptr(pnp), ptr(dpn), 128, ptr(ext), ptr(gxyz), ptr(mv), ptr(mt), ptr(greg), ptr(acc),
                                 ptr(gw_d), ptr(dh2), 128, B, HW, nreg, dt | cabi.PREZEROED, st), "head_tail_bwd")
    assert torch.equal(dh2[:, :72], dh[:, :72]) and float((dh2[:, 72:].float() - 7.0).abs().max()) == 0.0


def test_head_tail_generic_region_count(H):
    """a region count other than 64 takes the generic kernels (the 64-region fast path needs 72-float rows): same numbers as the fast
    path gives when the 32 surplus classes carry -inf-like logits and are never the target."""
    from gdrnet_amd import synth

    lib = cabi.load(BF16)
    dt = F32
    B, HW, hs = 1, 4096, 72
    M = B * HW
    b = synth.make_batch(B, seed=3)
    dev, st = H.DEV, H.stream()
    f = lambda t: t.to(dev).float().contiguous()
    c2d, ext, gxyz, mv, mt = f(b["roi_coord_2d"]), f(b["roi_extent"]), f(b["roi_xyz"]), f(b["roi_mask_visib"]), f(b["roi_mask_trunc"])
    greg = (b["roi_region"] % 33).to(dev).contiguous()
    head = torch.zeros(M, hs, device=dev)
    head[:, :37] = H.randn(5, M, 37).to(dev)
    head64 = head.clone()
    head64[:, 37:69] = -80.0          # classes 33..64: exp() == 0 next to the live ones
    out = {}
    for nreg, hd in ((32, head), (64, head64)):
        pnp = torch.zeros(M, 128, device=dev)
        acc = torch.zeros(8, dtype=torch.float64, device=dev)
        check(lib.gdrn_head_tail_loss_fwd(ptr(hd), hs, ptr(c2d), ptr(ext), ptr(pnp), 128, ptr(gxyz), ptr(mv), ptr(mt), ptr(greg), ptr(acc), B, HW, nreg,
                                          dt, st), "head_tail_loss_fwd")
        # the same sums through the partial rows (the generic path's loss kernel has 2048 workgroups at most, the 64-region one 4096)
        nrows = lib.gdrn_head_tail_loss_rows(B, HW, nreg, hs, 128)
        assert nrows == min(M // 16, 4096 if nreg == 64 else 2048)
        accr = torch.full((8 + 8 * nrows,), float("nan"), dtype=torch.float64, device=dev)
        lossr = torch.zeros(8, device=dev)
        check(lib.gdrn_head_tail_loss_fwd(ptr(hd), hs, ptr(c2d), ptr(ext), ptr(pnp), 128, ptr(gxyz), ptr(mv), ptr(mt), ptr(greg), ptr(accr), B, HW, nreg,
                                          dt | cabi.ACC_ROWS, st), "head_tail_loss_fwd rows")
        check(lib.gdrn_map_loss_finalize_rows(ptr(accr), nrows, B, HW, ptr(lossr), st), "map_loss_finalize_rows")
        np.testing.assert_allclose(accr[:6].cpu().numpy(), acc[:6].cpu().numpy(), rtol=1e-12)
        dh = torch.zeros(M, 128, device=dev)
        gw = torch.ones(5, device=dev)
        dpn = (H.randn(6, M, 128) * 0.01).to(dev)
        dpn[:, 37:] = 0
        check(lib.gdrn_head_tail_bwd(ptr(hd), hs, ptr(pnp), ptr(dpn), 128, ptr(ext), ptr(gxyz), ptr(mv), ptr(mt), ptr(greg), ptr(acc), ptr(gw), ptr(dh),
                                     128, B, HW, nreg, dt, st), "head_tail_bwd")
        torch.cuda.synchronize()
        out[nreg] = (pnp.cpu(), acc.cpu(), dh.cpu())
    # only where the mask is 1: a masked-out pixel's cross entropy is log(number of classes)
    vis = (mv.view(-1) > 0).cpu()
    assert float((out[32][0][:, :37] - out[64][0][:, :37]).abs().max()) < 1e-6
    assert float(out[64][0][:, 37:69].abs().max()) < 1e-20
    np.testing.assert_allclose(out[32][1][[0, 1, 2, 3, 5]].numpy(), out[64][1][[0, 1, 2, 3, 5]].numpy(), rtol=1e-6)
    assert float((out[32][2][vis][:, :37] - out[64][2][vis][:, :37]).abs().max()) < 1e-6


@pytest.mark.parametrize("rows,C_", [(1000, 64), (4096, 64), (513, 256), (130, 512), (64, 128)])
def test_bn_finalize_workspace(H, rows, C_):
    """two-launch finalize for many partial rows (fold to <= 64 rows, then finalise) == fp64 numpy, also on the second call."""
    lib = cabi.load(BF16)
    dev = H.DEV
    rng = np.random.default_rng(rows + C_)
    part = rng.normal(size=(rows, 2, C_)).astype(np.float32)
    part[:, 1] = np.abs(part[:, 1]) * 3 + 1.0
    count = float(rows * 7)
    pd = torch.from_numpy(part).to(dev)
    gam, bet = torch.rand(C_, device=dev) + 0.5, torch.randn(C_, device=dev)
    ws = torch.full((64 * 2 * C_,), float("nan"), dtype=torch.float64, device=dev)  # 64*2*C doubles, uninitialised
    m = part[:, 0].astype(np.float64).sum(0) / count
    var = np.maximum(part[:, 1].astype(np.float64).sum(0) / count - m * m, 0)
    for it in range(2):
        mk = lambda: torch.zeros(C_, device=dev)
        mean, invstd, scale, shift = mk(), mk(), mk(), mk()
        rm, rv, nbt = torch.zeros(C_, device=dev), torch.ones(C_, device=dev), torch.zeros((), dtype=torch.int64, device=dev)
        check(lib.gdrn_bn_finalize(ptr(pd), rows, C_, count, ptr(gam), ptr(bet), ptr(rm), ptr(rv), ptr(nbt), 0.1, 1e-5, ptr(mean),
                                   ptr(invstd), ptr(scale), ptr(shift), ptr(ws), H.stream()), "bn_finalize")
        torch.cuda.synchronize()
        np.testing.assert_allclose(mean.cpu().numpy(), m, rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(invstd.cpu().numpy(), 1 / np.sqrt(var + 1e-5), rtol=1e-6)
        np.testing.assert_allclose(rm.cpu().numpy(), 0.1 * m, rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(scale.cpu().numpy(), gam.cpu().numpy() / np.sqrt(var + 1e-5), rtol=2e-6)
        assert int(nbt) == 1


@pytest.mark.parametrize("M,K,N,act", [(64, 8192, 1024, 2), (4, 8192, 1024, 2), (37, 1024, 256, 0), (16, 128, 16, 1)])
def test_linear_splitk(H, M, K, N, act):
    """split-K skinny linear (Patch-PnP fc1) == F.linear + bias + activation; workspace needs no initialisation (called twice)."""
    lib = cabi.load(BF16)
    dev = H.DEV
    x = H.rounded(H.randn(200, M, K), BF16)
    w = H.rounded(H.randn(201, N, K) / math.sqrt(K), BF16)
    b = H.randn(202, N) * 0.1
    ref = F.linear(x, w, b)
    ref = F.relu(ref) if act == 1 else (F.leaky_relu(ref, 0.1) if act == 2 else ref)
    xd, wd, bd = x.to(dev).to(HT), w.to(dev).to(HT), b.to(dev)
    ws = torch.full((16 * M * N,), float("nan"), dtype=torch.float32, device=dev)  # GDRN_LINEAR_MAX_SPLITS slabs, uninitialised
    for _ in range(2):
        y = torch.full((M, N), float("nan"), dtype=HT, device=dev)
        check(lib.gdrn_linear_splitk(ptr(xd), ptr(wd), ptr(bd), ptr(y), M, K, N, K, K, N, act, ptr(ws), BF16, H.stream()), "linear_splitk")
        torch.cuda.synchronize()
        assert H.rel(y.float().cpu(), ref) < TOL[BF16]


@pytest.mark.parametrize("O,I", [(128, 64), (64, 256), (256, 128)])
def test_pack_multi_fragment_major(H, O, I):
    """gdrn_pack_multi (brick transpose through LDS) == gdrn_pack4 + gdrn_pack_wfrag for the forward and the flipped
    data-gradient operand of a 3x3 conv, both in ONE launch."""
    from gdrnet_amd.cabi import PackTask, to_device_table

    lib = cabi.load(BF16)
    dev, dt = H.DEV, BF16
    w = H.randn(300, O, I, 3, 3).to(dev).contiguous()
    ru64 = lambda v: (v + 63) // 64 * 64
    rows_f, cin_p, rows_d, cout_p = H.bn_rows(O), ru64(I), H.bn_rows(ru64(I)), ru64(O)
    specs = [(rows_f, cin_p, O, I, I * 9, 9, 0), (rows_d, cout_p, I, O, 9, I * 9, 1)]  # A1, B, A1v, Bv, s1, sb, flip
    refs, dsts, tasks, starts = [], [], [], [0]
    for A1, B, A1v, Bv, s1, sb, flip in specs:
        rowmajor = H.pack(w, A1, 1, 9, B, A1v, 1, Bv, s1, 0, 1, sb, flip, dt)
        ref = torch.empty_like(rowmajor)
        check(lib.gdrn_pack_wfrag(ptr(rowmajor), ptr(ref), A1, B, dt, H.stream()), "pack_wfrag")
        dst = torch.full((A1 * 9 * B,), float("nan"), dtype=HT, device=dev)
        kch = B // 64
        tasks.append(PackTask(src=ptr(w), dst=ptr(dst), A1=A1, A2=1, T=9, B=B, A1v=A1v, A2v=1, Bv=Bv, flip=flip, s1=s1, s2=0, st=1, sb=sb,
                              n=A1 * 9 * B, frag=1, pad_=kch.bit_length() if kch & (kch - 1) == 0 else 0))
        starts.append(starts[-1] + (A1 // 16) * (B // 64))
        refs.append(ref)
        dsts.append(dst)
    tab = to_device_table(tasks, dev)
    stt = torch.tensor(starts, dtype=torch.int32, device=dev)
    check(lib.gdrn_pack_multi(ptr(tab), ptr(stt), len(tasks), starts[-1], dt, H.stream()), "pack_multi")
    torch.cuda.synchronize()
    for ref, dst in zip(refs, dsts):
        assert torch.equal(dst.view(torch.int16), ref.reshape(-1).view(torch.int16))


@pytest.mark.parametrize("case", [(256, 1, 64, 128, 256, 1, 128, 8192, 0, 1, 64),      # fc1's forward operand (1/4 of its rows): u = t
                                  (64, 128, 1, 256, 64, 128, 256, 1, 64, 0, 8192),     # fc1's data-gradient operand (1/4 of its columns): u = a1
                                  (128, 1, 1, 128, 100, 1, 69, 1, 0, 1, 128),          # a 1x1 layer's data-gradient operand with padding on both sides
                                  (96, 1, 1, 72, 96, 1, 72, 1, 0, 1, 96)])             # extents that are no multiples of the 64 x 64 tile (B % 8 == 0)
def test_pack_unpack_multi_tiled_transpose(H, case):
    """r6 (ABI 5): the tiled-transpose path of gdrn_pack_multi / gdrn_unpack_multi (frag = 3: row-major copies whose unit-stride source index is
    a1 / a2 / t rather than b, moved as 64 x 64 tiles through LDS) == gdrn_pack4 / gdrn_unpack4 bit for bit, in bf16 and fp32.
    case = (A1, A2, T, B, A1v, A2v, Bv, s1, s2, st, sb)."""
    from gdrnet_amd.cabi import PackTask, to_device_table, transpose_blocks

    lib = cabi.load(BF16)
    dev = H.DEV
    A1, A2, T, B, A1v, A2v, Bv, s1, s2, st, sb = case
    nsrc = (A1v - 1) * s1 + (A2v - 1) * s2 + (T - 1) * st + (Bv - 1) * sb + 1
    src = H.randn(900, nsrc).to(dev).contiguous()
    for dt, tdt in ((BF16, HT), (F32, torch.float32)):
        ref = torch.full((A1 * A2 * T * B,), float("nan"), dtype=tdt, device=dev)
        check(lib.gdrn_pack4(ptr(src), ptr(ref), A1, A2, T, B, A1v, A2v, Bv, s1, s2, st, sb, 0, dt, H.stream()), "pack4")
        dst = torch.full_like(ref, float("nan"))
        t = PackTask(src=ptr(src), dst=ptr(dst), A1=A1, A2=A2, T=T, B=B, A1v=A1v, A2v=A2v, Bv=Bv, flip=0, s1=s1, s2=s2, st=st, sb=sb, n=A1 * A2 * T * B, frag=0, pad_=0)
        nb = transpose_blocks(lib, t)
        assert nb > 0 and t.frag == 3 and t.pad_ in (1, 3)
        tab = to_device_table([t], dev)
        stt = torch.tensor([0, nb], dtype=torch.int32, device=dev)
        check(lib.gdrn_pack_multi(ptr(tab), ptr(stt), 1, nb, dt, H.stream()), "pack_multi")
        torch.cuda.synchronize()
        assert torch.equal(dst.view(torch.int16 if dt == BF16 else torch.int32), ref.view(torch.int16 if dt == BF16 else torch.int32))
    # ... and back: packed fp32 [A1][A2][T][B] -> the strided layout (valid region only; the rest of the destination untouched)
    packed = H.randn(901, A1 * A2 * T * B).to(dev).contiguous()
    ref = torch.full((nsrc,), 7.0, device=dev)
    check(lib.gdrn_unpack4(ptr(packed), ptr(ref), A1, A2, T, B, A1v, A2v, Bv, s1, s2, st, sb, 0, H.stream()), "unpack4")
    dst = torch.full((nsrc,), 7.0, device=dev)
    t = PackTask(src=ptr(packed), dst=ptr(dst), A1=A1, A2=A2, T=T, B=B, A1v=A1v, A2v=A2v, Bv=Bv, flip=0, s1=s1, s2=s2, st=st, sb=sb, n=A1v * A2v * T * Bv, frag=0, pad_=0)
    nb = transpose_blocks(lib, t)
    assert nb > 0
    tab = to_device_table([t], dev)
    stt = torch.tensor([0, nb], dtype=torch.int32, device=dev)
    check(lib.gdrn_unpack_multi(ptr(tab), ptr(stt), 1, nb, H.stream()), "unpack_multi")
    torch.cuda.synchronize()
    assert torch.equal(dst, ref)
    # a task with a flip (or whose fastest source index IS b) does not qualify
    t2 = PackTask(src=ptr(src), dst=ptr(dst), A1=A1, A2=A2, T=T, B=B, A1v=A1v, A2v=A2v, Bv=Bv, flip=1, s1=s1, s2=s2, st=st, sb=sb, n=1, frag=0, pad_=0)
    assert transpose_blocks(lib, t2) == 0 and t2.frag == 0


@pytest.mark.parametrize("B", [1, 3])
def test_stem_conv_direct(H, B):
    """dedicated stem kernel (one kernel row per MFMA k-step, fragments straight from the NHWC4 canvas) == F.conv2d
    7x7 s2 p3, and its per-wave partial statistics sum to the tensor's sum / sum of squares."""
    lib = cabi.load(BF16)
    dev, dt = H.DEV, BF16
    img = torch.rand(B, 3, 256, 256, generator=torch.Generator().manual_seed(7))
    w = H.rounded(H.randn(400, 64, 3, 7, 7) / math.sqrt(147), dt)
    ref = F.conv2d(H.rounded(img, dt), w, None, 2, 3)
    canvas = torch.zeros(B, 262, 272, 4, dtype=HT, device=dev)
    imgd = img.to(dev).contiguous()
    check(lib.gdrn_pack_image(ptr(imgd), ptr(canvas), B, 256, 256, 262, 272, dt, H.stream()), "pack_image")
    w32 = torch.empty(64 * 7 * 32, dtype=HT, device=dev)
    wd = w.to(dev).contiguous()
    check(lib.gdrn_pack_stem_w32(ptr(wd), ptr(w32), dt, H.stream()), "pack_stem_w32")
    rows = lib.gdrn_stem_stats_rows(B)
    stats = torch.full((rows, 2, 64), float("nan"), dtype=torch.float32, device=dev)
    y = torch.full((B, 128, 128, 64), float("nan"), dtype=HT, device=dev)
    check(lib.gdrn_stem_conv(ptr(canvas), ptr(w32), ptr(y), ptr(stats), B, dt, H.stream()), "stem_conv")
    torch.cuda.synchronize()
    assert H.rel(H.nchw(y), ref) < TOL[dt]
    s = stats.double().sum(0).cpu()
    assert H.rel(s[0].float(), ref.sum((0, 2, 3))) < 2e-3 + 1e-2 and H.rel(s[1].float(), (ref ** 2).sum((0, 2, 3))) < 1e-2


@pytest.mark.parametrize("B,fused", [(1, False), (3, True), (2, True)])
def test_stem_wgrad_fused_bn_backward(H, B, fused):
    """dedicated stem weight gradient == autograd's conv weight gradient, with dy either given (fused=False) or evaluated on the
    fly as the BatchNorm-backward apply dy = a*g + (b*raw + c) (bn_bwd_apply's formula, bf16-rounded), incl. dgamma / dbeta."""
    lib = cabi.load(BF16)
    dev, dt = H.DEV, BF16
    gen = torch.Generator().manual_seed(11 + B)
    img = torch.rand(B, 3, 256, 256, generator=gen)
    g = H.rounded(torch.randn(B, 64, 128, 128, generator=gen) * 0.1, dt)
    raw = H.rounded(torch.randn(B, 64, 128, 128, generator=gen), dt)
    canvas = torch.zeros(B, 262, 272, 4, dtype=HT, device=dev)
    imgd = img.to(dev).contiguous()
    check(lib.gdrn_pack_image(ptr(imgd), ptr(canvas), B, 256, 256, 262, 272, dt, H.stream()), "pack_image")
    gd, rawd = H.nhwc(g, dt), H.nhwc(raw, dt)
    npix = B * 128 * 128
    mean, invstd, gamma = torch.randn(64, generator=gen) * 0.1, 0.5 + torch.rand(64, generator=gen), 0.5 + torch.rand(64, generator=gen)
    if fused:
        s1 = g.sum((0, 2, 3))
        s2 = (g * (raw - mean[None, :, None, None]) * invstd[None, :, None, None]).sum((0, 2, 3))
        a = gamma * invstd
        b = -a * invstd * (s2 / npix)
        c = -a * (s1 / npix) - b * mean
        dy = H.rounded(a[None, :, None, None] * g + (b[None, :, None, None] * raw + c[None, :, None, None]), dt)
        rows = torch.zeros(16, 2, 64)
        rows[3, 0], rows[5, 1] = s1 * 0.25, s2 * 0.5  # spread over partial rows: gdrn_bn_bwd_coef adds them up
        rows[9, 0], rows[0, 1] = s1 * 0.75, s2 * 0.5
    else:
        dy = g
    x = H.rounded(img, dt).requires_grad_(False)
    w = torch.zeros(64, 3, 7, 7, requires_grad=True)
    (F.conv2d(x, w, None, 2, 3) * dy).sum().backward()
    parts = lib.gdrn_stem_wgrad_parts(B)
    ws = torch.full((parts * 64 * 224,), float("nan"), dtype=torch.float32, device=dev)
    grad = torch.full((64, 3, 7, 7), float("nan"), dtype=torch.float32, device=dev)
    dgam = torch.full((64,), float("nan"), dtype=torch.float32, device=dev)
    dbet = torch.full((64,), float("nan"), dtype=torch.float32, device=dev)
    if fused:
        md, isd, gmd, sd = mean.to(dev), invstd.to(dev), gamma.to(dev), rows.to(dev).contiguous()
        ka, kb, kc = (torch.full((64,), float("nan"), dtype=torch.float32, device=dev) for _ in range(3))
        check(lib.gdrn_bn_bwd_coef(ptr(sd), 16, 64, npix, ptr(gmd), ptr(md), ptr(isd), ptr(ka), ptr(kb), ptr(kc), ptr(dgam), ptr(dbet), H.stream()), "bn_bwd_coef")
        check(lib.gdrn_stem_wgrad(ptr(canvas), ptr(gd), ptr(rawd), ptr(ka), ptr(kb), ptr(kc), B, ptr(ws), ptr(grad), dt, H.stream()), "stem_wgrad")
    else:
        check(lib.gdrn_stem_wgrad(ptr(canvas), ptr(gd), None, None, None, None, B, ptr(ws), ptr(grad), dt, H.stream()), "stem_wgrad")
    torch.cuda.synchronize()
    assert H.rel(grad.cpu(), w.grad) < (4e-3 if fused else 1e-4)  # fused: dy is re-rounded to bf16 from fp32 constants
    if fused:
        assert H.rel(dbet.cpu(), s1) < 1e-5 and H.rel(dgam.cpu(), s2) < 1e-5
    assert lib.gdrn_stem_wgrad(ptr(canvas), ptr(gd), None, ptr(dgam), None, None, B, ptr(ws), ptr(grad), dt, H.stream()) == -1


def test_nonfinite_flag(H):
    """gdrn_nonfinite_flag: raises the flag for an inf or a NaN anywhere in the buffer (incl. the last n % 4 elements), leaves it alone otherwise"""
    lib = cabi.load(BF16)
    n = 1_000_003
    x = torch.randn(n + 1, device=H.DEV)[:n]
    flag = torch.zeros(1, dtype=torch.int32, device=H.DEV)
    run = lambda: (check(lib.gdrn_nonfinite_flag(ptr(x), n, ptr(flag), H.stream()), "nonfinite"), torch.cuda.synchronize(), int(flag.item()))[2]
    assert run() == 0
    for pos, val in ((n - 1, float("inf")), (12345, float("nan")), (0, float("-inf"))):
        old = float(x[pos])
        x[pos] = val
        flag.zero_()
        assert run() == 1, pos
        x[pos] = old
    flag.zero_()
    assert run() == 0


@pytest.mark.parametrize("N", [1, 3])
def test_stem_conv_pool_eval_equals_the_two_kernel_path_bit_for_bit(H, N):
    """gdrn_stem_conv_pool (eval mode: conv 7x7 s2 + BatchNorm scale / shift + ReLU + 3x3 s2 max-pool in one kernel, r5) against
    gdrn_stem_conv -> gdrn_bn_relu_maxpool_fwd: the pooled tensor bit-identical (the fused kernel rounds the conv output to the 16-bit format
    where the two-kernel path stores it), and against torch on the rounded operands."""
    lib, dev, dt = cabi.load(BF16), H.DEV, BF16
    img = torch.rand(N, 3, 256, 256, generator=torch.Generator().manual_seed(7)).to(dev)
    w = H.randn(8, 64, 3, 7, 7) / math.sqrt(147.0)
    scale = (torch.rand(64, generator=torch.Generator().manual_seed(9)) + 0.5).to(dev)
    shift = (H.randn(10, 64) * 0.3).to(dev)
    canvas = torch.zeros(N, 262, 272, 4, dtype=HT, device=dev)
    check(lib.gdrn_pack_image(ptr(img), ptr(canvas), N, 256, 256, 262, 272, dt, H.stream()), "pack_image")
    w32 = torch.zeros(64 * 7 * 32, dtype=HT, device=dev)
    wd = w.to(dev).contiguous()
    check(lib.gdrn_pack_stem_w32(ptr(wd), ptr(w32), dt, H.stream()), "pack_stem_w32")
    raw = torch.empty(N, 128, 128, 64, dtype=HT, device=dev)
    check(lib.gdrn_stem_conv(ptr(canvas), ptr(w32), ptr(raw), None, N, dt, H.stream()), "stem_conv")
    want = torch.empty(N, 64, 64, 64, dtype=HT, device=dev)
    idx = torch.empty(N, 64, 64, 64, dtype=torch.uint8, device=dev)
    check(lib.gdrn_bn_relu_maxpool_fwd(ptr(raw), ptr(scale), ptr(shift), ptr(want), ptr(idx), N, 128, 128, 64, dt, H.stream()), "bn_relu_maxpool")
    got = torch.full((N, 64, 64, 64), float("nan"), dtype=HT, device=dev)
    check(lib.gdrn_stem_conv_pool(ptr(canvas), ptr(w32), ptr(scale), ptr(shift), ptr(got), N, dt, H.stream()), "stem_conv_pool")
    torch.cuda.synchronize()
    assert torch.isfinite(got.float()).all()
    assert torch.equal(got.view(torch.int16), want.view(torch.int16)), float((got.float() - want.float()).abs().max())
    ref = F.max_pool2d(F.relu(H.rounded(F.conv2d(H.rounded(img.cpu(), dt), H.rounded(w, dt), None, 2, 3), dt) * scale.cpu().view(1, -1, 1, 1)
                              + shift.cpu().view(1, -1, 1, 1)), 3, 2, 1)
    assert H.rel(H.nchw(got, 64), ref) < TOL[dt]


def test_head_conv_tail_eval_equals_the_two_launch_path(H):
    """gdrn_head_conv_tail_fwd (inference: the head's 1x1 output conv 256 -> 69 + the head tail in one kernel, r5) against the reference's
    arithmetic (F.conv2d on the rounded operands -> slice / softmax / concat / extent scaling, cdpn_rot_head_region.py:127-135, GDRN.py:156-169)
    and against the two-launch path gdrn_conv_gemm -> gdrn_head_tail_fwd on the same operands."""
    from gdrnet_amd import synth

    lib, dev, dt, st = cabi.load(BF16), H.DEV, BF16, H.stream()
    B, HW, nreg, hs = 2, 4096, 64, 72
    M = B * HW
    b = synth.make_batch(B, seed=9)
    x = H.rounded(torch.relu(H.randn(120, B, 256, 64, 64)), dt)
    w = H.rounded(H.randn(121, 69, 256, 1, 1) / 16.0, dt)
    bias = H.randn(122, 69) * 0.2
    logits = F.conv2d(x, w, bias)
    sm = torch.softmax(logits[:, 5:], 1)
    pnp_ref = torch.cat([(logits[:, 1:4] - 0.5) * b["roi_extent"].view(B, 3, 1, 1), b["roi_coord_2d"], sm], 1)
    f = lambda t: t.to(dev).float().contiguous()
    c2d, ext = f(b["roi_coord_2d"]), f(b["roi_extent"])
    xd = H.nhwc(x, dt)
    wrows = torch.zeros(128, 256, dtype=HT, device=dev)
    wrows[:69] = w.view(69, 256).to(dev).to(HT)
    bias_d = f(bias)
    head = torch.full((M, hs), float("nan"), device=dev)
    pnp = torch.zeros(M, 128, dtype=HT, device=dev)
    check(lib.gdrn_head_conv_tail_fwd(ptr(xd), 256, ptr(wrows), 128, ptr(bias_d), ptr(c2d), ptr(ext), ptr(head), hs, ptr(pnp), 128, B, HW, nreg, dt, st),
          "head_conv_tail_fwd")
    torch.cuda.synchronize()
    got_h = head[:, :69].cpu().view(B, 64, 64, 69).permute(0, 3, 1, 2)
    assert H.rel(got_h, logits) < 2e-5, H.rel(got_h, logits)
    assert float(head[:, 69:].abs().max()) == 0.0   # (rows 69..71 of the weight fragments are zero)
    got = pnp.float().cpu().view(B, 64, 64, 128)
    assert H.rel(got[..., :69].permute(0, 3, 1, 2), pnp_ref) < TOL[dt]
    assert float(got[..., 69:].abs().max()) == 0.0
    # the two-launch path on the same operands
    y2, _ = H.conv_gemm(xd, wrows.view(128, 1, 256), B, 64, 64, 256, 256, 64, 64, 69, 1, 1, 1, 0, dt, bias=bias_d, out_f32=1, y_cs=hs)
    pnp2 = torch.zeros(M, 128, dtype=HT, device=dev)
    check(lib.gdrn_head_tail_fwd(ptr(y2), hs, ptr(c2d), ptr(ext), ptr(pnp2), 128, B, HW, nreg, dt | cabi.PREZEROED, st), "head_tail_fwd")
    torch.cuda.synchronize()
    assert H.rel(head[:, :69], y2.view(M, hs)[:, :69]) < 2e-6
    assert float((pnp.float() - pnp2.float()).abs().max()) < 2e-3 and float((pnp != pnp2).float().mean()) < 2e-3   # rare one-ulp flips of the 16-bit rounding
    # head = NULL: the logits are not written, pnp_in unchanged
    pnp3 = torch.zeros(M, 128, dtype=HT, device=dev)
    check(lib.gdrn_head_conv_tail_fwd(ptr(xd), 256, ptr(wrows), 128, ptr(bias_d), ptr(c2d), ptr(ext), None, hs, ptr(pnp3), 128, B, HW, nreg, dt, st), "head_conv_tail_fwd")
    torch.cuda.synchronize()
    assert torch.equal(pnp3.view(torch.int16), pnp.view(torch.int16))


def test_head_conv_tail_loss_fwd_equals_the_two_launch_path(H):
    """gdrn_head_conv_tail_loss_fwd (do_loss=True: 1x1 output conv + head tail + the map-loss sums in one kernel, r5) against gdrn_conv_gemm ->
    gdrn_head_tail_loss_fwd on the same operands: logits, pnp_in and the five map losses (GDRN.py:345-400)."""
    from gdrnet_amd import synth

    lib, dev, dt, st = cabi.load(BF16), H.DEV, BF16, H.stream()
    B, HW, nreg, hs = 2, 4096, 64, 72
    M = B * HW
    b = synth.make_batch(B, seed=9)
    b["roi_mask_visib"][1, :8] = 0
    x = H.rounded(torch.relu(H.randn(130, B, 256, 64, 64)), dt)
    w = H.rounded(H.randn(131, 69, 256, 1, 1) / 16.0, dt)
    bias = H.randn(132, 69) * 0.2
    f = lambda t: t.to(dev).float().contiguous()
    c2d, ext, gxyz, mv, mt = f(b["roi_coord_2d"]), f(b["roi_extent"]), f(b["roi_xyz"]), f(b["roi_mask_visib"]), f(b["roi_mask_trunc"])
    greg = b["roi_region"].to(dev).contiguous()
    xd = H.nhwc(x, dt)
    wrows = torch.zeros(128, 256, dtype=HT, device=dev)
    wrows[:69] = w.view(69, 256).to(dev).to(HT)
    bias_d = f(bias)
    # two launches
    y2, _ = H.conv_gemm(xd, wrows.view(128, 1, 256), B, 64, 64, 256, 256, 64, 64, 69, 1, 1, 1, 0, dt, bias=bias_d, out_f32=1, y_cs=hs)
    rows2 = int(lib.gdrn_head_tail_loss_rows(B, HW, nreg, hs, 128))
    acc2 = torch.zeros(8 + 8 * rows2, dtype=torch.float64, device=dev)
    pnp2 = torch.zeros(M, 128, dtype=HT, device=dev)
    l2 = torch.zeros(8, device=dev)
    check(lib.gdrn_head_tail_loss_fwd(ptr(y2), hs, ptr(c2d), ptr(ext), ptr(pnp2), 128, ptr(gxyz), ptr(mv), ptr(mt), ptr(greg), ptr(acc2), B, HW, nreg,
                                      dt | cabi.PREZEROED | cabi.ACC_ROWS, st), "head_tail_loss_fwd")
    check(lib.gdrn_map_loss_finalize_rows(ptr(acc2), rows2, B, HW, ptr(l2), st), "finalize_rows")
    # one launch
    rows = int(lib.gdrn_head_conv_tail_loss_rows(B, HW))
    assert rows == min(M // 64, 1024)
    acc = torch.full((8 + 8 * rows,), float("nan"), dtype=torch.float64, device=dev)
    head = torch.full((M, hs), float("nan"), device=dev)
    pnp = torch.zeros(M, 128, dtype=HT, device=dev)
    l1 = torch.zeros(8, device=dev)
    check(lib.gdrn_head_conv_tail_loss_fwd(ptr(xd), 256, ptr(wrows), 128, ptr(bias_d), ptr(c2d), ptr(ext), ptr(head), hs, ptr(pnp), 128, ptr(gxyz), ptr(mv), ptr(mt),
                                           ptr(greg), ptr(acc), B, HW, nreg, dt | cabi.PREZEROED, st), "head_conv_tail_loss_fwd")
    check(lib.gdrn_map_loss_finalize_rows(ptr(acc), rows, B, HW, ptr(l1), st), "finalize_rows")
    torch.cuda.synchronize()
    assert H.rel(head[:, :69], y2.view(M, hs)[:, :69]) < 2e-6
    assert float((pnp.float() - pnp2.float()).abs().max()) < 2e-3 and float((pnp != pnp2).float().mean()) < 2e-3
    assert torch.isfinite(l1).all()
    np.testing.assert_allclose(l1[:5].cpu().numpy(), l2[:5].cpu().numpy(), rtol=2e-6)
    # without a logits buffer the loss variant refuses (the backward pass reads it)
    assert lib.gdrn_head_conv_tail_loss_fwd(ptr(xd), 256, ptr(wrows), 128, ptr(bias_d), ptr(c2d), ptr(ext), None, hs, ptr(pnp), 128, ptr(gxyz), ptr(mv), ptr(mt),
                                            ptr(greg), ptr(acc), B, HW, nreg, dt | cabi.PREZEROED, st) == -1


def test_head_out_dgrad_with_bn_backward_sums(H):
    """gdrn_head_out_dgrad (r5): the data gradient of the head's 1x1 output conv with the ReLU mask (affine: scale * raw + shift > 0) and the two
    BatchNorm-backward sums of the BatchNorm in front of it, against torch and against gdrn_conv_gemm's bnb_* epilogue on the same operands."""
    lib, dev, dt, st = cabi.load(BF16), H.DEV, BF16, H.stream()
    B, HW = 2, 4096
    M = B * HW
    dy = H.rounded(H.randn(140, B, 69, 64, 64) * 0.02, dt)
    w = H.rounded(H.randn(141, 69, 256, 1, 1) / 16.0, dt)
    raw = H.rounded(H.randn(142, B, 256, 64, 64) * 1.5 + 0.3, dt)
    mean, invstd = H.randn(143, 256) * 0.2, torch.rand(256, generator=torch.Generator().manual_seed(144)) + 0.5
    scale, shift = torch.rand(256, generator=torch.Generator().manual_seed(145)) + 0.5, H.randn(146, 256) * 0.3
    V = lambda t: t.view(1, -1, 1, 1)
    g = F.conv_transpose2d(dy, w)                      # [B, 256, 64, 64]
    m = (raw * V(scale) + V(shift)) > 0
    gm = g * m
    xhat = (raw - V(mean)) * V(invstd)
    ref1, ref2 = gm.sum((0, 2, 3)), (gm * xhat).sum((0, 2, 3))
    d = lambda t: t.to(dev).float().contiguous()
    dyd = torch.zeros(M, 128, dtype=HT, device=dev)
    dyd[:, :69] = dy.permute(0, 2, 3, 1).reshape(M, 69).to(dev).to(HT)
    wd = torch.zeros(256, 128, dtype=HT, device=dev)
    wd[:, :69] = w.view(69, 256).t().to(dev).to(HT)
    rawd = H.nhwc(raw, dt)
    nrows = int(lib.gdrn_head_out_dgrad_rows(B, HW))
    assert nrows == 128
    rows = torch.full((nrows, 2, 256), float("nan"), device=dev)
    dx = torch.full((M, 256), float("nan"), dtype=HT, device=dev)
    mean_d, invstd_d, scale_d, shift_d = d(mean), d(invstd), d(scale), d(shift)   # (named: a temporary's memory may be handed out again before the launch reads it)
    check(lib.gdrn_head_out_dgrad(ptr(dyd), 128, ptr(wd), 128, ptr(rawd), 256, ptr(mean_d), ptr(invstd_d), ptr(scale_d), ptr(shift_d), ptr(dx), 256, ptr(rows),
                                  B, HW, dt, st), "head_out_dgrad")
    torch.cuda.synchronize()
    got = dx.float().cpu().view(B, 64, 64, 256).permute(0, 3, 1, 2)
    assert torch.isfinite(got).all() and H.rel(got, gm) < TOL[dt]
    sums = rows.sum(0).cpu()
    assert H.rel(sums[0], ref1) < 2e-3 and H.rel(sums[1], ref2) < 2e-3
    # the generic kernel's epilogue on the same operands: the same masked gradient up to the 16-bit rounding, the same sums
    y2, s2 = H.conv_gemm(dyd.view(B, 64, 64, 128), wd.view(256, 1, 128), B, 64, 64, 128, 128, 64, 64, 256, 1, 1, 1, 0, dt,
                         bnb=dict(x=rawd, mean=mean_d, invstd=invstd_d, scale=scale_d, shift=shift_d))
    assert float((dx.float() - y2.view(M, 256).float()).abs().max()) < 1e-3 * float(gm.abs().max()) + 1e-6
    assert H.rel(rows.sum(0), s2.sum(0)) < 1e-4
