"""Bring-up aid (not product): the second-generation halo conv (conv3x3_v3.hip) against the first one (validated against
PyTorch in tests/test_kernels_gpu.py) on the layer shapes of the bs=64 step -- outputs, statistics rows, fused BatchNorm-backward
rows, operand transforms incl. the copy-out -- and the launch times of both.   python tools/v3check.py [quick]"""
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import hiputil as H  # noqa: E402
from gdrnet_amd.cabi import BF16  # noqa: E402

dev = "cuda"
torch.manual_seed(0)


G = [None]


def mk(B, Hh, C_, scale=1.0):
    return (torch.randn(B, Hh, Hh, C_, device=dev, generator=G[0]) * scale).to(torch.bfloat16)


def vec(C_, lo, hi):
    return (torch.rand(C_, device=dev, generator=G[0]) * (hi - lo) + lo).float()


def run(kind, B, C_, Hh, v3, reps):
    """kind: plain | stats | relu_bias | addend | xf1 | xf2 | xf3 | xf4 | bnb | bnb_mask_add | xf3_bnb"""
    g = G[0] = torch.Generator(device=dev).manual_seed(1)   # the same operands for both kernels
    x = (torch.randn(B, Hh, Hh, C_, device=dev, generator=g)).to(torch.bfloat16)
    w = (torch.randn(C_, 9, C_, device=dev, generator=g) * 0.05).to(torch.bfloat16)
    kw = dict(halo=True, v3=v3, reps=reps)
    xf = bnb = None
    extra = {}
    if kind == "stats":
        kw["want_stats"] = True
    elif kind == "relu_bias":
        kw.update(bias=vec(C_, -1, 1), act=1)
    elif kind == "addend":
        kw.update(addend=mk(B, Hh, C_))
    if kind.startswith("xf"):
        mode = int(kind[2])
        x2 = (torch.randn(B, Hh, Hh, C_, device=dev, generator=g)).to(torch.bfloat16)
        out = torch.full_like(x, float("nan"))
        xf = dict(mode=mode, relu=mode in (1, 2), x2=x2 if mode >= 2 else None, a=vec(C_, 0.5, 1.5), b=vec(C_, -1, 1) if mode >= 2 else None,
                  c=vec(C_, -0.5, 0.5), c2=vec(C_, -0.5, 0.5) if mode == 2 else None, msc=vec(C_, 0.5, 1.5) if mode == 4 else None,
                  msh=vec(C_, -0.5, 0.5) if mode == 4 else None, out=out)
        kw["xf"] = xf
        kw["want_stats"] = mode in (1, 2)
        extra["xf_out"] = out
    if "bnb" in kind:
        bx = (torch.randn(B, Hh, Hh, C_, device=dev, generator=g)).to(torch.bfloat16)
        bnb = dict(x=bx, mean=vec(C_, -0.2, 0.2), invstd=vec(C_, 0.5, 2.0))
        if "mask" in kind:
            bnb["mask"] = (torch.randn(B, Hh, Hh, C_, device=dev, generator=g)).clamp_min(0).to(torch.bfloat16)
            kw["addend"] = mk(B, Hh, C_)
        else:
            bnb.update(scale=vec(C_, 0.5, 1.5), shift=vec(C_, -0.5, 0.5))
        kw["bnb"] = bnb
        kw.pop("want_stats", None)
    r = H.conv_gemm(x, w, B, Hh, Hh, C_, C_, Hh, Hh, C_, 3, 3, 1, 1, BF16, **kw)
    y, st = r[0], r[1]
    ms = r[2] if reps else None
    return y, st, extra.get("xf_out"), ms


def rel(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    bad = 0
    for cfgs in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["12"]):
        os.environ["GDRN_V3_CFG"] = cfgs
        print("==== GDRN_V3_CFG=%s (large-map / small-map tile configuration)" % cfgs, flush=True)
        bad += main1(quick)
    return bad


def main1(quick):
    # correctness: small batch, every mode, both tile configurations (the 256-channel tile needs >= 256 workgroups: B*H*W/256 >= 256)
    cases = []
    for kind in ("plain", "stats", "relu_bias", "xf1", "xf2", "xf3", "xf4", "bnb", "xf3_bnb"):
        cases.append((kind, 64, 256, 32))     # 256 tiles of 16x16 -> 256-channel configuration
    for kind in ("plain", "stats", "relu_bias", "addend", "xf1", "xf2", "xf3", "xf4", "bnb", "bnb_mask_add", "xf3_bnb"):
        cases.append((kind, 3, 128, 32))      # 128-channel configuration, K split
        cases.append((kind, 2, 256, 16))
    cases.append(("stats", 2, 512, 16))
    cases.append(("xf3_bnb", 5, 128, 48))     # non-power-of-two map, odd batch
    bad = 0
    for kind, B, C_, Hh in cases:
        y0, s0, o0, _ = run(kind, B, C_, Hh, False, 0)
        y1, s1, o1, _ = run(kind, B, C_, Hh, True, 0)
        e = rel(y1, y0)
        msg = "%-13s B=%d C=%d H=%d  y %.2e" % (kind, B, C_, Hh, e)
        ok = e < 4e-3 and bool(torch.isfinite(y1.float()).all())
        if s0 is not None:
            es = rel(s1.sum(0), s0.sum(0))
            msg += "  rows %.2e" % es
            ok = ok and es < 2e-3
        if o0 is not None:
            same = bool(torch.equal(o1.view(torch.int16), o0.view(torch.int16)))
            msg += "  xf_out bit-equal %s" % same
            ok = ok and same
        print(("ok   " if ok else "FAIL ") + msg, flush=True)
        bad += 0 if ok else 1
    print("correctness: %d failures of %d" % (bad, len(cases)), flush=True)
    if quick:
        return bad
    # timing on the step's layer shapes (bs = 64)
    shapes = [(256, 64), (256, 32), (256, 16), (128, 32), (512, 16)]
    kinds = ("stats", "xf1", "xf3_bnb", "xf2", "bnb")
    for C_, Hh in shapes:
        fl = 2.0 * 64 * Hh * Hh * C_ * C_ * 9
        for kind in kinds:
            reps = 10
            t0 = run(kind, 64, C_, Hh, False, reps)[3]
            t1 = run(kind, 64, C_, Hh, True, reps)[3]
            print("time %-8s C=%d H=%d   old %7.1f us %6.0f TF   v3 %7.1f us %6.0f TF   x%.2f" % (kind, C_, Hh, t0 * 1e3, fl / t0 / 1e9, t1 * 1e3, fl / t1 / 1e9, t0 / t1), flush=True)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
