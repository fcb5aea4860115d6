"""Amplification-free bf16 parity (-m gpu): the throughput-mode engine at the BASELINE size (bs = 64), checked STAGE BY STAGE.

The end-to-end bf16 figures of tests/test_e2e_gpu.py carry the x80 - x1600 error amplification of a BatchNorm chain with batch
statistics and therefore cannot have tight bounds.  Here every stage of the plan -- each convolution, BatchNorm
(+residual)+ReLU, max-pool, upsampling, GroupNorm, fully connected layer, forward AND the data-gradient / weight-gradient chain -- is
re-evaluated on the CPU from the ENGINE'S OWN input tensors of that stage (teacher forcing) with the reference's arithmetic
(torch fp32 / fp64, operands as stored = bf16, cf. oracle/bf16_emulation.py for the storage points) and compared with the engine's
output tensor of the stage.  Both sides round the stage's fp32 result to bf16 once, so what remains is the summation order in front of
that rounding: measured <= 5.8e-4 relative L2 over the 193 bf16 tensors (bound 1e-3; one bf16 ulp is 3.9e-3), <= 2.8e-5 for the 53 fp32
parameter gradients.  A mis-wired tensor,
a wrong coefficient vector, a wrong tap / stride / mask anywhere in the 263-launch step fails ITS stage by orders of magnitude.

References per stage: torchvision BasicBlock / resnet_backbone.py:69-80, cdpn_rot_head_region.py:182-193, conv_pnp_net.py:111-157 and
their autograd backward (core/gdrn_modeling/engine.py:279)."""
import math
import os

import pytest
import torch
import torch.nn.functional as F

from gdrnet_amd import synth
from gdrnet_amd.cfg import lm13_cfg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
# Bounds = about 1.7x the worst value measured over the five plan variants below (printed by the test):
TOL_BF16 = 1e-3   # bf16 tensors: measured <= 5.8e-4 (a quarter of ONE bf16 ulp, 3.9e-3: engine and reference round the same fp32 value the
                  # same way almost everywhere; what differs is the summation order in front of the rounding)
TOL_GRAD = 5e-5   # fp32 weight gradients (bf16 operand products, fp32 accumulation over up to 262144 pixels): measured <= 2.8e-5
TOL_VEC = 3e-4    # fp32 head / Patch-PnP outputs and the BatchNorm vectors: measured <= 1.6e-4
TOL_SUM = 7e-3    # dgamma / dbeta: the engine sums the fp32 gradient BEFORE its bf16 rounding (data-gradient epilogue), the reference sums the
                  # stored tensor; the rounding noise does not cancel the way the signed terms of the sum do (measured <= 3.9e-3)
RESNET34_LAYERS = (3, 4, 6, 3)
HEAD_CONVS = ((3, 4, False), (6, 7, False), (10, 11, True), (13, 14, False), (17, 18, True), (20, 21, False))


_ST = [torch.bfloat16]   # the 16-bit storage format of the case: bf16, or fp16 for the fp16 library build (HIP_DTYPE="fp16")


def r(x):
    return x.to(_ST[0]).to(torch.float32)


def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def bn_stats_ref(raw, gamma, beta, eps=1e-5):
    """train-mode BatchNorm constants from the stored tensor (fp64 statistics, fp32 vectors: gdrn_bn_finalize's arithmetic)"""
    a = raw.double()
    m = a.mean((0, 2, 3))
    var = ((a * a).mean((0, 2, 3)) - m * m).clamp_min(0)
    inv = (1.0 / torch.sqrt(var + eps)).float()
    sc = gamma.float() * inv
    sh = beta.float() - m.float() * sc
    return m.float(), inv, sc, sh


def fma(x, a, c):
    return (x.double() * a.double().view(1, -1, 1, 1) + c.double().view(1, -1, 1, 1)).float()


def bn_bwd(g, x, gamma, mean, inv):
    """dx, dgamma, dbeta of a train-mode BatchNorm for the (masked) upstream gradient g"""
    n = g.shape[0] * g.shape[2] * g.shape[3]
    V = lambda t: t.double().view(1, -1, 1, 1)
    xh = (x.double() - V(mean)) * V(inv)
    gd = g.double()
    dbeta, dgamma = gd.sum((0, 2, 3)), (gd * xh).sum((0, 2, 3))
    dx = V(gamma) * V(inv) * (gd - V(dbeta) / n - xh * V(dgamma) / n)
    return dx.float(), dgamma.float(), dbeta.float()


# (batch size, environment): the BASELINE size with the default plan; the plan variants (separate BatchNorm passes, no BatchNorm-backward
# epilogue in the generic kernel, first / second generation halo kernel everywhere it applies) at a size the CPU reference finishes quickly
# ... and the fp16 arithmetic mode (the same kernels built with IEEE half, csrc/common.h): the 16-bit tensors are rounded in 2^-12 instead of
# 2^-9 steps, so the bounds of everything that carries a 16-bit rounding are 8x tighter; the data-gradient chain runs under the engine's
# static loss scale (every check is relative and from the engine's own stage inputs: the scale drops out unless something underflows)
CASES = [(64, {}, "bf16"), (8, {"GDRN_GEMM_BNB": "0"}, "bf16"), (8, {"GDRN_FUSE_XF": "0"}, "bf16"), (8, {"GDRN_V3": "2"}, "bf16"),
         (8, {"GDRN_V3": "0"}, "bf16"), (64, {}, "fp16"), (8, {"GDRN_V3": "2"}, "fp16")]
TOLS = {"bf16": (1e-3, 7e-3), "fp16": (1.25e-4, 9e-4)}   # (16-bit tensors, dgamma / dbeta)


@pytest.mark.parametrize("B,env,dtype", CASES, ids=["bs64-default", "bs8-no-gemm-bnb", "bs8-unfused-bn", "bs8-v3-everywhere", "bs8-no-v3",
                                                    "bs64-fp16", "bs8-fp16-v3-everywhere"])
def test_every_stage_of_the_bf16_step_against_its_own_inputs(B, env, dtype, monkeypatch):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    run_case(B, env, dtype)


def run_case(B, env, dtype):
    """the stage-by-stage check of one plan (also called by __graft_entry__.smoke at bs = 4); returns [(stage, error, bound)] after asserting
    every bound"""
    from gdrnet_amd import GDRN as G

    global TOL_BF16, TOL_SUM
    TOL_BF16, TOL_SUM = TOLS[dtype]
    _ST[0] = torch.float16 if dtype == "fp16" else torch.bfloat16   # read by r(); every case sets it
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    cfg = lm13_cfg(device=DEV)
    cfg.MODEL.CDPN.HIP_DTYPE = dtype
    model, _ = G.build_model_optimizer(cfg)
    sd = synth.make_state_dict(0)
    model.load_state_dict(sd)
    model.train()
    batch = {k: (v.to(DEV) if isinstance(v, torch.Tensor) else v) for k, v in synth.make_batch(B, seed=3).items()}
    kw = synth.model_kwargs(batch, do_loss=True)
    kw.pop("do_loss")
    model.train_step(batch["roi_img"], optimizer=None, **kw)
    torch.cuda.synchronize()
    eng = model.engine()
    plan = eng.plan(B, True, True)
    P = {k: v.detach().float().cpu() for k, v in model.named_parameters()}
    W = lambda n: r(P[n])
    # fp16: the gradient chain (plan tensors d_*) runs under the engine's static loss scale; train_step(optimizer=None) has already divided
    # it out of the parameter gradients -- put it back so that every stage is checked against its own (scaled) inputs
    GR = {n: g.detach().float().cpu() * eng.loss_scale for n, g in eng.grads.items()}

    def T(name, C=None):
        """plan tensor (NHWC on the device) -> NCHW fp32 on the host, first C channels"""
        t = plan.tensors[name].detach().float().cpu()
        if t.dim() == 2 and C is not None and name in ("head_out", "pnp_in", "d_head", "d_pnp_in"):
            t = t.view(B, 64, 64, -1)
        if t.dim() == 4:
            t = t[..., :C] if C else t
            return t.permute(0, 3, 1, 2).contiguous()
        return t[..., :C] if C else t

    res = []   # (stage, tensor, error, bound)
    stat_chk = []

    def bn_stats(raw, gamma, beta, key=None):
        """BatchNorm vectors of a stage: the ENGINE's (mean, invstd, scale, shift) -- they are inputs of the apply / mask / backward
        stages -- after checking them against the statistics of the tensor the engine stored (the engine reduces the fp32
        accumulators in the conv epilogue, before the bf16 rounding: the two differ by the rounding noise averaged over >= 4096 pixels)"""
        m_, i_, sc_, sh_ = bn_stats_ref(raw, gamma, beta)
        s_ = plan.bn[key]
        em, ei, esc, esh = s_.mean.float().cpu(), s_.invstd.float().cpu(), s_.scale.float().cpu(), s_.shift.float().cpu()
        stat_chk.append((key + " statistics (mean | invstd)", rel(torch.cat([em, ei]), torch.cat([m_, i_])), TOL_VEC))
        stat_chk.append((key + " scale | shift", rel(torch.cat([esc, esh]), torch.cat([sc_, sh_])), TOL_VEC))
        return em, ei, esc, esh

    def chk(stage, got, ref, tol=TOL_BF16):
        e = rel(got, ref)
        res.append((stage, e, tol))

    img = r(batch["roi_img"].float().cpu())

    # ------------------------------------------------------------------ forward
    raw0 = T("stem.raw")
    chk("stem conv", raw0, r(F.conv2d(img, W("backbone.conv1.weight"), None, 2, 3)))
    m0, i0, sc0, sh0 = bn_stats(raw0, P["backbone.bn1.weight"], P["backbone.bn1.bias"], "backbone.bn1")
    p0 = T("stem.pool")
    chk("stem bn+relu+maxpool", p0, r(F.max_pool2d(F.relu(fma(raw0, sc0, sh0)), 3, 2, 1)))

    x = p0
    bnc = {}   # bn key -> (mean, invstd, scale, shift) recomputed from the stored raw tensor
    blocks = []
    for li, nb in enumerate(RESNET34_LAYERS, start=1):
        for b in range(nb):
            q = f"backbone.layer{li}.{b}"
            s = 2 if (b == 0 and li > 1) else 1
            raw1, a1, raw2, out = T(q + ".raw1"), T(q + ".a1"), T(q + ".raw2"), T(q + ".out")
            chk(q + ".conv1", raw1, r(F.conv2d(x, W(q + ".conv1.weight"), None, s, 1)))
            bnc[q + ".bn1"] = c1 = bn_stats(raw1, P[q + ".bn1.weight"], P[q + ".bn1.bias"], q + ".bn1")
            chk(q + ".bn1+relu", a1, r(F.relu(fma(raw1, c1[2], c1[3]))))
            chk(q + ".conv2", raw2, r(F.conv2d(a1, W(q + ".conv2.weight"), None, 1, 1)))
            bnc[q + ".bn2"] = c2 = bn_stats(raw2, P[q + ".bn2.weight"], P[q + ".bn2.bias"], q + ".bn2")
            if s == 2:
                rawd = T(q + ".rawd")
                chk(q + ".downsample conv", rawd, r(F.conv2d(x, W(q + ".downsample.0.weight"), None, 2, 0)))
                bnc[q + ".downsample.1"] = cd = bn_stats(rawd, P[q + ".downsample.1.weight"], P[q + ".downsample.1.bias"], q + ".downsample.1")
                idn = r(fma(rawd, cd[2], cd[3]))
            else:
                idn = x
            chk(q + ".bn2+add+relu", out, r(F.relu(fma(raw2, c2[2], c2[3]) + idn)))
            blocks.append((q, li, b, s, x))
            x = out
    feat = x
    h = "rot_head_net.features."
    rawt, h0 = T(h + "0.raw"), T(h + "0.act")
    chk("head convT", rawt, r(F.conv_transpose2d(feat, W(h + "0.weight"), None, stride=2, padding=1, output_padding=1)))
    bnc[h + "1"] = ct = bn_stats(rawt, P[h + "1.weight"], P[h + "1.bias"], h + "1")
    chk("head convT bn+relu", h0, r(F.relu(fma(rawt, ct[2], ct[3]))))
    hx = h0
    head = []
    for ci, bi, up in HEAD_CONVS:
        if up:
            u = T(h + f"{ci}.up")
            chk(h + f"{ci} upsample", u, r(F.interpolate(hx, scale_factor=2, mode="bilinear", align_corners=True)))
            xin = u
        else:
            xin = hx
        raw = T(h + f"{ci}.raw")
        chk(h + f"{ci} conv", raw, r(F.conv2d(xin, W(h + f"{ci}.weight"), None, 1, 1)))
        bnc[h + str(bi)] = c = bn_stats(raw, P[h + f"{bi}.weight"], P[h + f"{bi}.bias"], h + str(bi))
        if (h + f"{ci}.act") in plan.tensors:
            act = T(h + f"{ci}.act")
            chk(h + f"{bi} bn+relu", act, r(F.relu(fma(raw, c[2], c[3]))))
        else:   # (r6) in front of a fused upsampling the activation is never stored: the upsampling evaluates it (checked as the next stage's input)
            act = r(F.relu(fma(raw, c[2], c[3])))
        head.append((ci, bi, up, xin, hx))
        hx = act
    maps = T("head_out", 69)
    chk("head 1x1 conv (fp32 out)", maps, F.conv2d(hx, W(h + "23.weight"), P[h + "23.bias"], 1, 0), TOL_VEC)
    ext = batch["roi_extent"].float().cpu().view(B, 3, 1, 1)
    pin_ref = r(torch.cat([(maps[:, 1:4] - 0.5) * ext, batch["roi_coord_2d"].float().cpu(), F.softmax(maps[:, 5:], dim=1)], 1))
    pnp_in = T("pnp_in", 69)
    chk("head tail (slice, softmax, concat, extent)", pnp_in, pin_ref)
    qn = "pnp_net.features."
    px = pnp_in
    pnp = []
    for ci, gi in ((0, 1), (3, 4), (6, 7)):
        raw, act = T(qn + f"{ci}.raw"), T(qn + f"{ci}.act")
        chk(qn + f"{ci} conv", raw, r(F.conv2d(px, W(qn + f"{ci}.weight"), None, 2, 1)))
        chk(qn + f"{gi} groupnorm+relu", act, r(F.relu(F.group_norm(raw, 32, P[qn + f"{gi}.weight"], P[qn + f"{gi}.bias"], 1e-5))))
        pnp.append((ci, gi, px, raw, act))
        px = act
    flat = px.reshape(B, 128 * 8 * 8)
    f1, f2 = T("pnp_net.fc1.act"), T("pnp_net.fc2.act")
    chk("pnp fc1", f1, r(F.leaky_relu(F.linear(flat, W("pnp_net.fc1.weight"), P["pnp_net.fc1.bias"]), 0.1)))
    chk("pnp fc2", f2, r(F.leaky_relu(F.linear(f1, W("pnp_net.fc2.weight"), P["pnp_net.fc2.bias"]), 0.1)))
    fc_out = T("fc_out", 9)
    fc_ref = torch.cat([F.linear(f2, W("pnp_net.fc_r.weight"), P["pnp_net.fc_r.bias"]), F.linear(f2, W("pnp_net.fc_t.weight"), P["pnp_net.fc_t.bias"])], 1)
    chk("pnp fc_r | fc_t (fp32 out)", fc_out, fc_ref, TOL_VEC)

    res += stat_chk

    # ------------------------------------------------------------------ backward: Patch-PnP
    d_fc = T("d_fc", 9)
    Wrt = torch.cat([W("pnp_net.fc_r.weight"), W("pnp_net.fc_t.weight")], 0)
    d_f2 = T("pnp_net.fc2.d_act")
    chk("bwd fc_rt dgrad", d_f2, r(d_fc @ Wrt))
    chk("bwd fc_r wgrad", GR["pnp_net.fc_r.weight"], d_fc[:, :6].t() @ f2, TOL_GRAD)
    chk("bwd fc_t wgrad", GR["pnp_net.fc_t.weight"], d_fc[:, 6:9].t() @ f2, TOL_GRAD)
    d_f2p = T("pnp_net.fc2.d_pre")
    chk("bwd fc2 leaky", d_f2p, r(d_f2 * torch.where(f2 > 0, 1.0, 0.1)))
    d_f1 = T("pnp_net.fc1.d_act")
    chk("bwd fc2 dgrad", d_f1, r(d_f2p @ W("pnp_net.fc2.weight")))
    chk("bwd fc2 wgrad", GR["pnp_net.fc2.weight"], d_f2p.t() @ f1, TOL_GRAD)
    chk("bwd fc2 bias", GR["pnp_net.fc2.bias"], d_f2p.sum(0), TOL_GRAD)
    d_f1p = T("pnp_net.fc1.d_pre")
    chk("bwd fc1 leaky", d_f1p, r(d_f1 * torch.where(f1 > 0, 1.0, 0.1)))
    chk("bwd fc1 wgrad", GR["pnp_net.fc1.weight"], d_f1p.t() @ flat, TOL_GRAD)
    d_act = T(qn + "6.d_act")
    chk("bwd fc1 dgrad", d_act, r((d_f1p @ W("pnp_net.fc1.weight")).view(B, 128, 8, 8)))
    for ci, gi, pxin, raw, act in reversed(pnp):
        d_act, d_raw = T(qn + f"{ci}.d_act"), T(qn + f"{ci}.d_raw")
        rr = raw.clone().requires_grad_(True)
        gam, bet = P[qn + f"{gi}.weight"].clone().requires_grad_(True), P[qn + f"{gi}.bias"].clone().requires_grad_(True)
        F.relu(F.group_norm(rr, 32, gam, bet, 1e-5)).backward(d_act)
        chk(qn + f"{gi} bwd groupnorm+relu", d_raw, r(rr.grad))
        chk(qn + f"{gi} bwd gamma", GR[qn + f"{gi}.weight"], gam.grad, TOL_SUM)
        chk(qn + f"{gi} bwd beta", GR[qn + f"{gi}.bias"], bet.grad, TOL_SUM)
        w = W(qn + f"{ci}.weight")
        cin = w.shape[1]
        chk(qn + f"{ci} bwd wgrad", GR[qn + f"{ci}.weight"], torch.nn.grad.conv2d_weight(pxin[:, :cin], w.shape, d_raw, stride=2, padding=1), TOL_GRAD)
        dxin = F.conv_transpose2d(d_raw, w, None, stride=2, padding=1, output_padding=1)
        prev = {0: None, 3: qn + "0.d_act", 6: qn + "3.d_act"}[ci]
        if prev is not None:
            chk(qn + f"{ci} bwd dgrad", T(prev), r(dxin))
        else:
            chk(qn + "0 bwd dgrad", T("d_pnp_in", 69), r(dxin))

    # ------------------------------------------------------------------ backward: geometric head
    d_head = T("d_head", 69)
    w23 = W(h + "23.weight")
    chk(h + "23 bwd wgrad", GR[h + "23.weight"], torch.nn.grad.conv2d_weight(hx, w23.shape, d_head), TOL_GRAD)
    chk(h + "23 bwd bias", GR[h + "23.bias"], d_head.sum((0, 2, 3)), TOL_GRAD)
    up_grad = F.conv_transpose2d(d_head, w23)   # gradient w.r.t. the last head activation (before its ReLU mask)
    for ci, bi, up, xin, hprev in reversed(head):
        raw = T(h + f"{ci}.raw")
        m, inv, sc, sh = bnc[h + str(bi)]
        d_act, d_raw = T(h + f"{ci}.d_act"), T(h + f"{ci}.d_raw")
        mask = fma(raw, sc, sh) > 0
        nxt_up = {3: False, 6: True, 10: False, 13: True, 17: False, 20: not eng.gemm_bnb}[ci]   # (20: the 1x1 conv's epilogue masks it when GDRN_GEMM_BNB=1)
        if nxt_up:   # the stored tensor is the unmasked gradient (it comes out of the upsampling's backward)
            chk(h + f"{ci} d_act (upsample backward)", d_act, r(up_grad))
            g = d_act * mask
        else:        # masked by the producing launch's epilogue
            chk(h + f"{ci} d_act (masked dgrad)", d_act, r(up_grad * mask))
            g = d_act
        dx, dgam, dbet = bn_bwd(g, raw, P[h + f"{bi}.weight"], m, inv)
        chk(h + f"{bi} bwd bn", d_raw, r(dx))
        chk(h + f"{bi} bwd gamma", GR[h + f"{bi}.weight"], dgam, TOL_SUM)
        chk(h + f"{bi} bwd beta", GR[h + f"{bi}.bias"], dbet, TOL_SUM)
        w = W(h + f"{ci}.weight")
        chk(h + f"{ci} bwd wgrad", GR[h + f"{ci}.weight"], torch.nn.grad.conv2d_weight(xin, w.shape, d_raw, padding=1), TOL_GRAD)
        din = F.conv_transpose2d(d_raw, w, None, stride=1, padding=1)
        if up:
            chk(h + f"{ci} d_up", T(h + f"{ci}.d_up"), r(din))
            xx = hprev.clone().requires_grad_(True)
            F.interpolate(xx, scale_factor=2, mode="bilinear", align_corners=True).backward(T(h + f"{ci}.d_up"))
            up_grad = xx.grad
        else:
            up_grad = din
    m, inv, sc, sh = bnc[h + "1"]
    d_h0, d_rawt = T(h + "0.d_act"), T(h + "0.d_raw")
    chk("head convT d_act (masked dgrad)", d_h0, r(up_grad * (fma(rawt, sc, sh) > 0)))
    dx, dgam, dbet = bn_bwd(d_h0, rawt, P[h + "1.weight"], m, inv)
    chk("head convT bwd bn", d_rawt, r(dx))
    chk(h + "1 bwd gamma", GR[h + "1.weight"], dgam, TOL_SUM)
    ff = feat.clone().requires_grad_(True)
    wt = W(h + "0.weight").clone().requires_grad_(True)
    F.conv_transpose2d(ff, wt, None, stride=2, padding=1, output_padding=1).backward(d_rawt)
    chk("head convT bwd wgrad", GR[h + "0.weight"], wt.grad, TOL_GRAD)
    d_next = ff.grad   # gradient w.r.t. the backbone output (before the ReLU mask of layer4.2)

    # ------------------------------------------------------------------ backward: residual blocks (last to first)
    g_res = None   # residual-path gradient flowing into the block output together with d_next
    for q, li, b, s, xin in reversed(blocks):
        raw1, a1, raw2, out = T(q + ".raw1"), T(q + ".a1"), T(q + ".raw2"), T(q + ".out")
        d_out, d_raw2, d_a1, d_raw1 = T(q + ".d_out"), T(q + ".d_raw2"), T(q + ".d_a1"), T(q + ".d_raw1")
        tot = d_next + (g_res if g_res is not None else 0)
        if (q + ".g2") in plan.tensors:   # no masking epilogue on the producing launch: the BatchNorm-backward pass masks and stores g2
            chk(q + " d_out (dgrad + residual)", d_out, r(tot))
            g2t = T(q + ".g2")
            chk(q + " g2 (relu mask)", g2t, r(d_out * (out > 0)))
            d_out = g2t
        else:
            chk(q + " d_out (masked dgrad + residual)", d_out, r(tot * (out > 0)))
        m2, i2, sc2, sh2 = bnc[q + ".bn2"]
        dx, dgam, dbet = bn_bwd(d_out, raw2, P[q + ".bn2.weight"], m2, i2)
        chk(q + ".bn2 bwd", d_raw2, r(dx))
        chk(q + ".bn2 bwd gamma", GR[q + ".bn2.weight"], dgam, TOL_SUM)
        chk(q + ".bn2 bwd beta", GR[q + ".bn2.bias"], dbet, TOL_SUM)
        w2 = W(q + ".conv2.weight")
        chk(q + ".conv2 bwd wgrad", GR[q + ".conv2.weight"], torch.nn.grad.conv2d_weight(a1, w2.shape, d_raw2, padding=1), TOL_GRAD)
        m1, i1, sc1, sh1 = bnc[q + ".bn1"]
        chk(q + ".conv2 bwd dgrad + bn1 relu mask", d_a1, r(F.conv_transpose2d(d_raw2, w2, None, stride=1, padding=1) * (fma(raw1, sc1, sh1) > 0)))
        dx, dgam, dbet = bn_bwd(d_a1, raw1, P[q + ".bn1.weight"], m1, i1)
        chk(q + ".bn1 bwd", d_raw1, r(dx))
        chk(q + ".bn1 bwd gamma", GR[q + ".bn1.weight"], dgam, TOL_SUM)
        w1 = W(q + ".conv1.weight")
        chk(q + ".conv1 bwd wgrad", GR[q + ".conv1.weight"], torch.nn.grad.conv2d_weight(xin, w1.shape, d_raw1, stride=s, padding=1), TOL_GRAD)
        if s == 2:
            rawd = T(q + ".rawd")
            md, idd, scd, shd = bnc[q + ".downsample.1"]
            d_rawd = T(q + ".d_rawd")
            dx, dgam, dbet = bn_bwd(d_out, rawd, P[q + ".downsample.1.weight"], md, idd)
            chk(q + ".downsample bn bwd", d_rawd, r(dx))
            wd = W(q + ".downsample.0.weight")
            chk(q + ".downsample bwd wgrad", GR[q + ".downsample.0.weight"], torch.nn.grad.conv2d_weight(xin, wd.shape, d_rawd, stride=2), TOL_GRAD)
            if (q + ".d_xd") in plan.tensors:
                d_xd = T(q + ".d_xd")
                chk(q + ".downsample bwd dgrad", d_xd, r(F.conv_transpose2d(d_rawd, wd, None, stride=2, padding=0, output_padding=1)))
            else:   # (r6) added inside conv1's data-gradient launch, never stored: checked as part of the previous block's d_out below
                d_xd = F.conv_transpose2d(d_rawd, wd, None, stride=2, padding=0, output_padding=1)
            d_next = F.conv_transpose2d(d_raw1, w1, None, stride=2, padding=1, output_padding=1)
            g_res = d_xd
        else:
            d_next = F.conv_transpose2d(d_raw1, w1, None, stride=1, padding=1)
            g_res = d_out
    # stem: max-pool backward with the ReLU mask recomputed, BatchNorm backward inside the weight-gradient kernel
    chk("stem d_pool", T("stem.d_pool"), r(d_next + g_res))
    pp = fma(raw0, sc0, sh0).clone().requires_grad_(True)
    F.max_pool2d(F.relu(pp), 3, 2, 1).backward(T("stem.d_pool"))
    g_stem = T("stem.g")
    chk("stem maxpool+relu bwd", g_stem, r(pp.grad))
    dx, dgam, dbet = bn_bwd(g_stem, raw0, P["backbone.bn1.weight"], m0, i0)
    chk("stem bn bwd gamma", GR["backbone.bn1.weight"], dgam, TOL_SUM)
    # (the stem's weight gradient is one of the two kernels that still accumulate with fp32 atomics: its error moves with the arrival order,
    #  measured 1.5e-5 .. 6.3e-5 over the plan variants in round 4 -- twice the deterministic kernels' bound)
    chk("stem conv wgrad (bn backward fused)", GR["backbone.conv1.weight"], torch.nn.grad.conv2d_weight(img, P["backbone.conv1.weight"].shape, r(dx), stride=2, padding=3), 2 * TOL_GRAD)

    # ------------------------------------------------------------------ verdict
    worst = sorted(res, key=lambda t: -t[1] / t[2])
    n_b = sum(1 for t in res if t[2] == TOL_BF16)
    print("teacher-forced bf16 parity at bs=%d %s: %d stages (%d bf16 tensors); worst relative L2 by bound:" % (B, env, len(res), n_b))
    for stage, e, tol in worst[:6]:
        print("   %-58s %.2e  (bound %.0e)" % (stage, e, tol))
    for tol, what in ((TOL_BF16, "bf16 tensors"), (TOL_GRAD, "fp32 weight gradients"), (TOL_SUM, "dgamma / dbeta"), (TOL_VEC, "fp32 outputs / BatchNorm vectors")):
        sel = [t for t in res if t[2] == tol]
        if sel:
            wst = max(sel, key=lambda t: t[1])
            print("   max over %3d %-34s %.2e  (%s)" % (len(sel), what, wst[1], wst[0]))
    med = sorted(t[1] for t in res if t[2] == TOL_BF16)[n_b // 2]
    print("   median over the bf16 tensors: %.2e" % med)
    bad = [(s, "%.2e" % e, tol) for s, e, tol in res if not (e < tol) or not math.isfinite(e)]
    assert not bad, bad[:20]
    assert len(res) >= 380
    return res
