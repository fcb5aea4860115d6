#!/usr/bin/env python3
"""Exploration for tests/test_e2e_gpu.py::test_bf16_parity_on_a_conditioned_network: error of the bf16 engine against the fp32
engine (itself within 1e-4 of the reference) on networks of different conditioning: the random-init synth weights, and the same
network after k fp32 training steps; train-mode (batch statistics) and eval-mode (running statistics, folded BatchNorm) forward."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gdrnet_amd import GDRN, synth
from gdrnet_amd.cfg import lm13_cfg

DEV = "cuda:0"
def rel(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return float((a - b).norm() / max(float(b.norm()), 1e-30))

def build(dtype, sd):
    cfg = lm13_cfg(device=DEV); cfg.MODEL.CDPN.HIP_DTYPE = dtype
    m, opt = GDRN.build_model_optimizer(cfg); m.load_state_dict(sd); return m, opt

def fwd(model, batch, B, train):
    kw = synth.model_kwargs(batch, do_loss=train)
    if train:
        model.train()
        with torch.no_grad(): model(batch["roi_img"], **kw)
        plan = model.engine().plan(B, True, True)
    else:
        model.eval(); kw.pop("do_loss")
        kw = {k: v for k, v in kw.items() if not k.startswith("gt_") and k != "sym_infos"}
        with torch.no_grad(): model(batch["roi_img"], do_loss=False, **kw)
        plan = model.engine().plan(B, False, False)
    torch.cuda.synchronize()
    return dict(rot6d=plan.fc_out[:, :6].clone(), t_=plan.fc_out[:, 6:9].clone(), rot=plan.rot.clone(), trans=plan.trans.clone(), maps=plan.head_out[:, :69].clone(),
                losses=(plan.losses.clone() if train else None))

def compare(tag, sd, B, seed):
    batch = {k: (v.to(DEV) if isinstance(v, torch.Tensor) else v) for k, v in synth.make_batch(B, seed=seed).items()}
    out = {}
    for train in (True, False):
        r = {}
        for dt in ("fp32", "bf16"):
            m, _ = build(dt, sd)
            r[dt] = fwd(m, batch, B, train)
        e = {k: rel(r["bf16"][k], r["fp32"][k]) for k in ("rot6d", "t_", "rot", "trans", "maps")}
        # per-RoI rotation angle error in degrees
        Ra, Rb = r["bf16"]["rot"].view(B, 3, 3).double(), r["fp32"]["rot"].view(B, 3, 3).double()
        cos = ((Ra.transpose(1, 2) @ Rb).diagonal(dim1=1, dim2=2).sum(1) - 1) / 2
        ang = torch.rad2deg(torch.acos(cos.clamp(-1, 1)))
        print("%-26s B=%2d %-5s " % (tag, B, "train" if train else "eval") + " ".join("%s %.2e" % kv for kv in e.items()) + "  rot-angle err deg: mean %.3f max %.3f" % (ang.mean(), ang.max()))

sd0 = synth.make_state_dict(0)
for B in (4, 16, 64):
    compare("synth init", sd0, B, 31)
# condition by training in fp32
for k, lr in ((20, 1e-4), (100, 1e-3), (300, 1e-3)):
    m, opt = build("fp32", sd0)
    for g in opt.param_groups: g["lr"] = lr
    m.train()
    Bt = 16
    for it in range(k):
        b = {kk: (v.to(DEV) if isinstance(v, torch.Tensor) else v) for kk, v in synth.make_batch(Bt, seed=100 + it % 8).items()}
        kw = synth.model_kwargs(b, do_loss=True); kw.pop("do_loss")
        L = m.train_step(b["roi_img"], optimizer=opt, **kw)
    torch.cuda.synchronize()
    print("after %d steps lr %g: losses" % (k, lr), [round(float(x), 4) for x in L])
    sd = {kk: v.detach().cpu().clone() for kk, v in m.state_dict().items()}
    for B in (4, 16):
        compare("trained %d steps lr %g" % (k, lr), sd, B, 31)
