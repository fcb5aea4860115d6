#!/usr/bin/env python3
"""Stage-by-stage comparison of the bf16 train step with fused (GDRN_FUSE_XF=1) and separate (=0) BatchNorm apply passes, plus a repeat
of the separate run (run-to-run reproducibility).  Prints every plan tensor / BatchNorm coefficient vector that differs, in backward
execution order for the gradients.  Bring-up aid for tests/test_e2e_gpu.py::test_fused_batchnorm_applies_equal_separate_passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gdrnet_amd import GDRN, synth
from gdrnet_amd.cfg import lm13_cfg

DEV = "cuda:0"
B = int(os.environ.get("XF_DIFF_B", "4"))

def run(fx):
    os.environ["GDRN_FUSE_XF"] = fx
    cfg = lm13_cfg(device=DEV)
    cfg.MODEL.CDPN.HIP_DTYPE = "bf16"
    model, _ = GDRN.build_model_optimizer(cfg)
    model.load_state_dict(synth.make_state_dict(0))
    model.train()
    batch = {k: (v.to(DEV) if isinstance(v, torch.Tensor) else v) for k, v in synth.make_batch(B, seed=5).items()}
    kw = synth.model_kwargs(batch, do_loss=True); kw.pop("do_loss")
    model.train_step(batch["roi_img"], optimizer=None, **kw)
    torch.cuda.synchronize()
    eng = model.engine(); plan = eng.plan(B, True, True)
    t = {k: v.float().cpu().clone() for k, v in plan.tensors.items()}
    co = {}
    for k, s in plan.bn.items():
        for nm in ("mean", "invstd", "scale", "shift", "ka", "kb", "kc"):
            v = getattr(s, nm, None)
            if v is not None: co[k + ":" + nm] = v.cpu().clone()
    g = {n: x.cpu().clone() for n, x in eng.grads.items()}
    return t, co, g

def rel(a, b):
    return float((a.double() - b.double()).norm() / max(float(b.double().norm()), 1e-30))

def cmp(A, Bm, title, order=None):
    keys = list(A.keys()) if order is None else order
    nd = 0
    for k in keys:
        if k not in Bm: continue
        if not torch.equal(A[k], Bm[k]):
            nd += 1
            d = (A[k] != Bm[k]).float().mean().item()
            print("  %-48s rel %.3e  frac differing %.4f" % (k, rel(A[k], Bm[k]), d))
    print("== %s: %d of %d differ" % (title, nd, len(keys)))

u1 = run("0"); u2 = run("0"); f1 = run("1")
tk = list(u1[0].keys())
fwd = [k for k in tk if ".d_" not in k]
bwd = [k for k in reversed(tk) if ".d_" in k]
print("##### separate vs separate (reproducibility)")
cmp(u2[0], u1[0], "forward tensors", fwd); cmp(u2[0], u1[0], "backward tensors (backward order)", bwd); cmp(u2[1], u1[1], "BN vectors", list(reversed(list(u1[1].keys()))))
print("##### fused vs separate")
cmp(f1[0], u1[0], "forward tensors", fwd); cmp(f1[0], u1[0], "backward tensors (backward order)", bwd); cmp(f1[1], u1[1], "BN vectors", list(reversed(list(u1[1].keys()))))
cmp(f1[2], u1[2], "parameter gradients")
