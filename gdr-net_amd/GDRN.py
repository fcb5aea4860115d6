"""Drop-in for the reference's ``core/gdrn_modeling/models/GDRN.py`` on MI355X.

Keeps the reference's model-construction / forward / loss API surface:

* ``build_model_optimizer(cfg) -> (model, optimizer)``                     (GDRN.py:550-724)
* ``GDRN.forward(x, gt_xyz=..., ..., roi_extents=..., resize_ratios=..., do_loss=False)``
  returning ``out_dict`` or ``(out_dict, loss_dict)``                       (GDRN.py:83-306)
* ``get_xyz_mask_region_out_dim(cfg)``                                      (GDRN.py:524-547)
* sub-modules ``backbone`` / ``rot_head_net`` / ``pnp_net`` whose parameters and buffers carry the
  reference's state_dict names and logical shapes, so released checkpoints load.

The sub-modules are *parameter containers*: the whole path (forward, losses, backward) runs as one
autograd node on the hand-written HIP kernels of ``libgdrn_hip.so`` (engine.py); there is no ATen
compute fallback -- without the library or a GPU, forward raises ``GdrnHipError``.

Only the branches the shipped configs enable are implemented: ResNet-34, non-concat
region head (class-agnostic, L1 xyz / L1 mask / CE region), ConvPnPNet with 2D coords + region
attention, ``allo_rot6d`` + ``centroid_z`` (REL), PM_R (L1, norm-by-extent, optional symmetry), L1
centroid / z.  Any other setting raises ``NotImplementedError`` at construction.
"""
import logging
import os

import numpy as np
import torch
import torch.nn as nn

from . import cabi
from .engine import LOSS_NAMES, Engine

logger = logging.getLogger(__name__)

try:  # the reference pushes vis/* scalars into detectron2's EventStorage from inside forward (GDRN.py:302-303)
    from detectron2.utils.events import get_event_storage as _d2_get_event_storage
except Exception:  # detectron2 is not a dependency of this package
    _d2_get_event_storage = None


class _ScalarSink:
    def __init__(self):
        self.scalars = {}

    def put_scalars(self, **kw):
        self.scalars.update(kw)


_fallback_storage = _ScalarSink()


def get_event_storage():
    if _d2_get_event_storage is not None:
        try:
            return _d2_get_event_storage()
        except Exception:
            pass
    return _fallback_storage


# resnet_backbone.py:8-14 (only BasicBlock depths are on the path)
resnet_spec = {
    18: ("BasicBlock", [2, 2, 2, 2], [64, 64, 128, 256, 512], "resnet18"),
    34: ("BasicBlock", [3, 4, 6, 3], [64, 64, 128, 256, 512], "resnet34"),
}


def _normal_init(m, std):
    nn.init.normal_(m.weight, 0.0, std)
    if getattr(m, "bias", None) is not None:
        nn.init.constant_(m.bias, 0.0)


class _Container(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError(
            f"{type(self).__name__} is a parameter container; the path runs fused through GDRN.forward on the HIP engine"
        )


class BasicBlock(_Container):
    """Parameter layout of torchvision BasicBlock (resnet_backbone.py:3)."""

    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride


class ResNetBackboneNet(_Container):
    """resnet_backbone.py:17-51 (ResNet-34: BasicBlock, [3,4,6,3]); init: normal std 0.001, BN weight 1."""

    def __init__(self, layers=(3, 4, 6, 3), in_channel=3, freeze=False, rot_concat=False):
        super().__init__()
        if freeze or rot_concat:
            raise NotImplementedError("BACKBONE.FREEZE / ROT_CONCAT are not on the hot path (SURVEY.md section 8)")
        self.freeze, self.rot_concat = freeze, rot_concat
        self.inplanes = 64
        self.conv1 = nn.Conv2d(in_channel, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.layer1 = self._make_layer(64, layers[0])
        self.layer2 = self._make_layer(128, layers[1], 2)
        self.layer3 = self._make_layer(256, layers[2], 2)
        self.layer4 = self._make_layer(512, layers[3], 2)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                _normal_init(m, 0.001)

    def _make_layer(self, planes, blocks, stride=1):
        ds = None
        if stride != 1 or self.inplanes != planes:
            ds = nn.Sequential(nn.Conv2d(self.inplanes, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))
        blks = [BasicBlock(self.inplanes, planes, stride, ds)]
        self.inplanes = planes
        for _ in range(1, blocks):
            blks.append(BasicBlock(planes, planes))
        return nn.Sequential(*blks)


class RotWithRegionHead(_Container):
    """cdpn_rot_head_region.py:9-143, non-concat branch: features.0 ConvT, .1 BN, then (conv, BN) pairs at
    (3,4) (6,7) (10,11) (13,14) (17,18) (20,21) with bilinear x2 at .9/.16, .23 the 1x1 output conv."""

    def __init__(self, in_channels=512, num_filters=256, rot_output_dim=3, mask_output_dim=1, num_regions=64):
        super().__init__()
        f = nn.ModuleList()
        f.append(nn.ConvTranspose2d(in_channels, num_filters, 3, 2, 1, 1, bias=False))
        f.append(nn.BatchNorm2d(num_filters))
        f.append(nn.Identity())  # ReLU
        for i in range(3):
            if i >= 1:
                f.append(nn.Identity())  # UpsamplingBilinear2d(scale_factor=2)
            for _ in range(2):
                f.append(nn.Conv2d(num_filters, num_filters, 3, 1, 1, bias=False))
                f.append(nn.BatchNorm2d(num_filters))
                f.append(nn.Identity())  # ReLU
        self.rot_output_dim, self.mask_output_dim, self.region_output_dim = rot_output_dim, mask_output_dim, num_regions + 1
        f.append(nn.Conv2d(num_filters, mask_output_dim + rot_output_dim + num_regions + 1, 1, bias=True))
        self.features = f
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                _normal_init(m, 0.001)


class ConvPnPNet(_Container):
    """conv_pnp_net.py:41-109: 3 x (conv3x3 s2, GN(32), ReLU), fc1 8192->1024, fc2 ->256, fc_r, fc_t."""

    def __init__(self, nIn, featdim=128, rot_dim=6, num_layers=3, norm="GN", num_gn_groups=32, num_regions=64, drop_prob=0.0,
                 dropblock_size=5, mask_attention_type="none"):
        super().__init__()
        if norm != "GN" or num_gn_groups != 32 or featdim != 128 or num_layers != 3 or mask_attention_type != "none":
            raise NotImplementedError("only ConvPnPNet(GN(32), featdim 128, 3 layers, no mask attention) is on the hot path")
        if drop_prob != 0.0:
            raise NotImplementedError("DropBlock is config-dead (drop_prob=0.0 in every shipped config)")
        self.featdim, self.num_regions, self.mask_attention_type, self.drop_prob = featdim, num_regions, mask_attention_type, drop_prob
        f = nn.ModuleList()
        for i in range(3):
            f.append(nn.Conv2d(nIn if i == 0 else featdim, featdim, 3, 2, 1, bias=False))
            f.append(nn.GroupNorm(num_gn_groups, featdim))
            f.append(nn.Identity())
        self.features = f
        self.fc1 = nn.Linear(featdim * 8 * 8, 1024)
        self.fc2 = nn.Linear(1024, 256)
        self.fc_r = nn.Linear(256, rot_dim)
        self.fc_t = nn.Linear(256, 3)
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                _normal_init(m, 0.001)
        _normal_init(self.fc_r, 0.01)
        _normal_init(self.fc_t, 0.01)


def get_xyz_mask_region_out_dim(cfg):
    """GDRN.py:524-547."""
    r_head_cfg = cfg.MODEL.CDPN.ROT_HEAD
    if r_head_cfg.XYZ_LOSS_TYPE in ["MSE", "L1", "L2", "SmoothL1"]:
        r_out_dim = 3
    elif r_head_cfg.XYZ_LOSS_TYPE in ["CE_coor", "CE"]:
        r_out_dim = 3 * (r_head_cfg.XYZ_BIN + 1)
    else:
        raise NotImplementedError(f"unknown xyz loss type: {r_head_cfg.XYZ_LOSS_TYPE}")
    if r_head_cfg.MASK_LOSS_TYPE in ["L1", "BCE"]:
        mask_out_dim = 1
    elif r_head_cfg.MASK_LOSS_TYPE in ["CE"]:
        mask_out_dim = 2
    else:
        raise NotImplementedError(f"unknown mask loss type: {r_head_cfg.MASK_LOSS_TYPE}")
    region_out_dim = r_head_cfg.NUM_REGIONS + 1
    assert region_out_dim > 2, region_out_dim
    return r_out_dim, mask_out_dim, region_out_dim


def _check_supported(cfg):
    m = cfg.MODEL.CDPN
    r, p, t = m.ROT_HEAD, m.PNP_NET, m.TRANS_HEAD
    bad = []
    if m.get("USE_MTL", False): bad.append("USE_MTL")
    if m.BACKBONE.NUM_LAYERS != 34 or m.BACKBONE.INPUT_RES != 256 or m.BACKBONE.OUTPUT_RES != 64: bad.append("BACKBONE")
    if r.ROT_CONCAT or r.FREEZE or r.ROT_CLASS_AWARE or r.MASK_CLASS_AWARE or r.REGION_CLASS_AWARE: bad.append("ROT_HEAD flags")
    if r.XYZ_LOSS_TYPE != "L1" or r.MASK_LOSS_TYPE != "L1" or r.REGION_LOSS_TYPE != "CE": bad.append("loss types")
    if r.XYZ_LOSS_MASK_GT != "visib" or r.MASK_LOSS_GT != "trunc" or r.REGION_LOSS_MASK_GT != "visib": bad.append("loss masks")
    if r.NUM_LAYERS != 3 or r.NUM_FILTERS != 256 or r.CONV_KERNEL_SIZE != 3 or r.OUT_CONV_KERNEL_SIZE != 1 or r.NORM != "BN": bad.append("head arch")
    if not (1 < r.NUM_REGIONS <= 64): bad.append("NUM_REGIONS")
    if t.ENABLED or p.R_ONLY or p.FREEZE: bad.append("TRANS_HEAD/R_ONLY/FREEZE")
    if not (p.WITH_2D_COORD and p.REGION_ATTENTION and p.MASK_ATTENTION == "none"): bad.append("PNP inputs")
    if p.ROT_TYPE != "allo_rot6d" or p.TRANS_TYPE != "centroid_z" or p.Z_TYPE != "REL": bad.append("pose parametrisation")
    if not (p.PM_LW > 0 and p.PM_R_ONLY and p.PM_NORM_BY_EXTENT and p.PM_LOSS_TYPE == "L1"): bad.append("PM loss")
    if p.ROT_LW > 0 or p.TRANS_LW > 0 or p.get("BIND_LW", 0.0) > 0: bad.append("rot/trans/bind losses")
    if not (p.CENTROID_LW > 0 and p.CENTROID_LOSS_TYPE == "L1" and p.Z_LW > 0 and p.Z_LOSS_TYPE == "L1"): bad.append("centroid/z losses")
    if bad:
        raise NotImplementedError("config outside the MI355X hot-path scope (SURVEY.md section 8): " + ", ".join(bad))


class _PathFn(torch.autograd.Function):
    """The whole path (backbone -> head -> Patch-PnP -> pose -> losses) as ONE autograd node."""

    @staticmethod
    def forward(ctx, model, plan, kctx, *params):
        plan.run_forward(kctx)
        ctx.model, ctx.plan, ctx.kctx = model, plan, kctx
        ctx.keep = kctx.get("_keep")
        ctx.generation = plan.generation
        return plan.losses.clone()

    @staticmethod
    def backward(ctx, glosses):
        plan, e = ctx.plan, ctx.plan.e
        if not plan.has_backward:
            raise cabi.GdrnHipError("backward needs train-mode BatchNorm (model.train()) together with do_loss=True")
        if plan.generation != ctx.generation:
            # the plan keeps ONE set of activations per (batch size, mode): a second forward overwrote what this backward needs
            raise cabi.GdrnHipError("backward of a forward pass whose activations were overwritten by a later forward of the same "
                                    "batch size: call loss.backward() before the next model(...) call")
        dyn = e.loss_scale_dev() if hasattr(e, "loss_scale_dev") else None   # fp16 with the dynamic loss scale: the device-resident state (None otherwise)
        ls = 1.0 if dyn is not None else float(getattr(e, "_ls_host", getattr(e, "loss_scale", 1.0)))
        if dyn is not None:
            gl = glosses.to(torch.float32).contiguous()
            cabi.check(e.lib.gdrn_scaled_loss_weights(gl.data_ptr(), None, 8, dyn.ptr, plan.gw.data_ptr(), e._stream()), "scaled_loss_weights")
        else:
            plan.gw.copy_(glosses.to(torch.float32) * ls)   # (fp16, static: the engine's loss scale on the whole gradient chain)
        plan._gw_key = None
        plan.run_backward(ctx.kctx, on_bucket=ctx.model._on_bucket)
        red = getattr(ctx.model, "_reducer", None)
        if red is not None:
            # autograd clones these views into .grad on the compute stream right after we return: the bucket all-reduces
            # running on the reducer's side stream must have landed (and the mean be applied) before that
            red.finish()
        if dyn is not None:
            ctx.model._unscale_for_external_optimizer(e, dyn, 1.0)
        elif ls != 1.0:
            e.grad_flat.mul_(1.0 / ls)   # .grad is the unscaled gradient, as after GradScaler.unscale_()
        return (None, None, None) + tuple(e.grads[n] for n in e.param_names)


class _StagingShapeChanged(Exception):
    pass


class GDRN(nn.Module):
    def __init__(self, cfg, backbone, rot_head_net, trans_head_net=None, pnp_net=None):
        super().__init__()
        assert cfg.MODEL.CDPN.NAME == "GDRN", cfg.MODEL.CDPN.NAME
        _check_supported(cfg)
        self.backbone = backbone
        self.rot_head_net = rot_head_net
        self.pnp_net = pnp_net
        self.trans_head_net = trans_head_net
        self.cfg = cfg
        self.concat = cfg.MODEL.CDPN.ROT_HEAD.ROT_CONCAT
        self.r_out_dim, self.mask_out_dim, self.region_out_dim = get_xyz_mask_region_out_dim(cfg)
        # Arithmetic of the kernels: "bf16" (default), "fp16", "fp32" (parity mode).  An explicit cfg.MODEL.CDPN.HIP_DTYPE / GDRN_HIP_DTYPE
        # applies to training and inference alike.  Without one the reference's two AMP switches select fp16, the format of its autocast:
        # cfg.SOLVER.AMP.ENABLED (autocast + GradScaler around the train step, main_gdrn.py:53-56,141, engine.py:276-283) -> fp16 training
        # under the engine's loss scale (dynamic like GradScaler's by default, kept on the device: engine.LossScaleState; every path that hands
        # gradients out -- train_step with or without an optimizer, loss.backward() -- runs the finite check, and an overflowed pass leaves zero
        # gradients / a skipped fused step, never inf / NaN); cfg.TEST.AMP_TEST (autocast around the test-time forward, gdrn_evaluator.py:568)
        # -> fp16 inference (eval-mode forward) whatever the training arithmetic is.
        explicit = cfg.MODEL.CDPN.get("HIP_DTYPE", os.environ.get("GDRN_HIP_DTYPE"))
        solver, test = cfg.get("SOLVER", None), cfg.get("TEST", None)
        amp_train = bool(solver is not None and solver.get("AMP", None) is not None and solver.AMP.get("ENABLED", False))
        amp_test = bool(test is not None and test.get("AMP_TEST", False))
        self.hip_dtype = str(explicit) if explicit else ("fp16" if amp_train else "bf16")
        self.hip_dtype_eval = str(explicit) if explicit else ("fp16" if amp_test else self.hip_dtype)
        self._engs = {}          # arithmetic -> (parameter-storage key, Engine)
        self._eng = None         # the engine of the last engine() call (dist.attach / broadcast_parameters look at it)
        self._on_bucket = None   # set by dist.attach(): overlap the RCCL all-reduce with backward
        self._loss_w = None
        self.last_vis = None

    # -------------------------------------------------------------------------------------------
    def engine(self, dtype=None):
        """The HIP engine bound to the current parameter storage (rebuilt after .to()/.load_state_dict re-allocation) for the arithmetic of
        the current mode: hip_dtype in train mode, hip_dtype_eval in eval mode (they differ only for cfg.TEST.AMP_TEST); one engine per
        arithmetic in use."""
        dtype = dtype or (self.hip_dtype if self.training else self.hip_dtype_eval)
        params = dict(self.named_parameters())
        key = tuple(p.data_ptr() for p in params.values())
        hit = self._engs.get(dtype)
        if hit is None or hit[0] != key:
            dev = next(iter(params.values())).device
            if dev.type != "cuda":
                raise cabi.GdrnHipError("GDRN runs on the HIP engine only: move the model to an MI355X (`model.to('cuda')`)")
            r, p = self.cfg.MODEL.CDPN.ROT_HEAD, self.cfg.MODEL.CDPN.PNP_NET
            hit = (key, Engine(params, dict(self.named_buffers()), dtype=dtype, num_regions=r.NUM_REGIONS, bn_cell=self.__dict__.setdefault("_bn_cell", [0])))
            self._engs = {k: v for k, v in self._engs.items() if v[0] == key}   # engines of a re-allocated parameter set are dead
            self._engs[dtype] = hit
            lw = [r.XYZ_LW, r.XYZ_LW, r.XYZ_LW, r.MASK_LW, r.REGION_LW, p.PM_LW, p.CENTROID_LW, p.Z_LW]
            self._loss_w = torch.tensor(lw, dtype=torch.float32, device=dev)
        self._eng = hit[1]
        return self._eng

    @staticmethod
    def _f32(t, dev):
        return t.detach().to(device=dev, dtype=torch.float32).contiguous()

    @staticmethod
    def padded_batch(B):
        """inference batch size of B RoIs: next power of two up to 64, multiples of 64 beyond"""
        if B <= 1:
            return 1
        return 1 << (B - 1).bit_length() if B <= 64 else (B + 63) // 64 * 64

    def _pad_rows(self, name, t, Bp):
        """t ([B, ...] fp32 on the device) in the first rows of a persistent [Bp, ...] buffer; the other rows hold an inert RoI
        (zero image, unit extents / sizes / ratios / intrinsics) or what an earlier, larger batch left there -- finite either way."""
        bufs = self.__dict__.setdefault("_pad_bufs", {})
        key = (name, Bp, tuple(t.shape[1:]))
        buf = bufs.get(key)
        if buf is None:
            fill = 0.0 if name in ("img", "coord2d", "centers") else 1.0
            buf = bufs[key] = torch.full((Bp,) + tuple(t.shape[1:]), fill, dtype=t.dtype, device=t.device)
            if name == "cams":
                buf.copy_(torch.eye(3, dtype=t.dtype, device=t.device).expand(Bp, 3, 3))
        buf[: t.shape[0]].copy_(t, non_blocking=True)
        return buf

    def _pack_sym(self, sym_infos, B, dev):
        """list of (K,3,3) / None (pm_loss.py:91-92) -> fp32 [B][Kmax][9] + int32 counts."""
        mats = [None if s is None else torch.as_tensor(s).detach().cpu().to(torch.float32).reshape(-1, 9) for s in sym_infos]
        K = max([m.shape[0] for m in mats if m is not None], default=0)
        if K == 0:
            return None, None, 0
        sym = torch.zeros(B, K, 9, dtype=torch.float32)
        cnt = torch.zeros(B, dtype=torch.int32)
        for i, m in enumerate(mats):
            if m is not None:
                sym[i, : m.shape[0]] = m
                cnt[i] = m.shape[0]
        return sym.to(dev), cnt.to(dev), K

    def _prepare(self, x, do_loss, a, staging=None):
        """Validate / stage the call's tensors (fp32, contiguous, on the device) and pick the plan.
        staging: dict name -> persistent device buffer; when given, every input is copied into its buffer and the plan
        is bound to the buffers (fixed addresses for hipGraph replay) and the operand repack is left to the caller."""
        cfg = self.cfg
        gt_xyz, gt_mask_trunc, gt_mask_visib, gt_region, gt_ego_rot = a['gt_xyz'], a['gt_mask_trunc'], a['gt_mask_visib'], a['gt_region'], a['gt_ego_rot']
        gt_points, sym_infos, gt_trans, gt_trans_ratio = a['gt_points'], a['sym_infos'], a['gt_trans'], a['gt_trans_ratio']
        roi_coord_2d, roi_cams, roi_centers, roi_whs = a['roi_coord_2d'], a['roi_cams'], a['roi_centers'], a['roi_whs']
        roi_extents, resize_ratios = a['roi_extents'], a['resize_ratios']
        eng = self.engine()
        dev = eng.dev
        B = x.shape[0]
        assert tuple(x.shape[1:]) == (3, 256, 256), x.shape
        assert roi_coord_2d is not None and roi_extents is not None
        f32 = lambda t: self._f32(t, dev)

        def S(name, t):  # into the persistent buffer of that name (graph mode), else the tensor itself
            if staging is None:
                return t
            buf = staging.get(name)
            if buf is None:
                buf = staging[name] = torch.empty_like(t)
            elif buf.shape != t.shape or buf.dtype != t.dtype:
                raise _StagingShapeChanged(name)
            buf.copy_(t, non_blocking=True)
            return buf

        cams = f32(roi_cams)
        if cams.dim() == 2:
            cams = cams.unsqueeze(0).expand(B, 3, 3).contiguous()
        ins = [("img", f32(x)), ("coord2d", f32(roi_coord_2d)), ("extents", f32(roi_extents)), ("cams", cams),
               ("centers", f32(roi_centers)), ("whs", f32(roi_whs)), ("ratios", f32(resize_ratios).reshape(B))]
        Bp = B
        if not self.training and not do_loss and staging is None:
            # Inference: the reference's test loop feeds "all detections of one image" (gdrn_evaluator.py:549-578, data_loader.py:707-765),
            # so B changes with every call.  Eval-mode RoIs are independent (running statistics), hence the batch is rounded up to the next
            # power of two (multiples of 64 beyond) with inert RoIs in persistent buffers: at most 7 plans instead of one per B.
            Bp = self.padded_batch(B)
            if Bp != B:
                ins = [(n, self._pad_rows(n, t, Bp)) for n, t in ins]
        keep = [S(n, t) for n, t in ins]
        kctx = dict(img=keep[0].data_ptr(), coord2d=keep[1].data_ptr(), extents=keep[2].data_ptr(), cams=keep[3].data_ptr(),
                    centers=keep[4].data_ptr(), whs=keep[5].data_ptr(), ratios=keep[6].data_ptr(), _keep=keep)
        if do_loss:
            assert (gt_xyz is not None) and (gt_trans is not None) and (gt_trans_ratio is not None) and (gt_region is not None)
            assert (gt_points is not None) and (gt_ego_rot is not None)
            g = [S("gt_xyz", f32(gt_xyz)), S("mask_visib", f32(gt_mask_visib)), S("mask_trunc", f32(gt_mask_trunc)),
                 S("gt_region", gt_region.detach().to(device=dev, dtype=torch.int64).contiguous()),
                 S("gt_rot", f32(gt_ego_rot)), S("gt_trans", f32(gt_trans)), S("gt_trans_ratio", f32(gt_trans_ratio)),
                 S("points", f32(gt_points))]
            keep += g
            kctx.update(gt_xyz=g[0].data_ptr(), mask_visib=g[1].data_ptr(), mask_trunc=g[2].data_ptr(), gt_region=g[3].data_ptr(),
                        gt_rot=g[4].data_ptr(), gt_trans=g[5].data_ptr(), gt_trans_ratio=g[6].data_ptr(), points=g[7].data_ptr(),
                        npts=int(g[7].shape[1]))
            if cfg.MODEL.CDPN.PNP_NET.PM_LOSS_SYM:
                assert sym_infos is not None
                sym, cnt, K = self._pack_sym(sym_infos, B, dev)
                if K > 0:
                    sym, cnt = S("sym", sym), S("sym_count", cnt)
                    keep += [sym, cnt]
                    kctx.update(sym=sym.data_ptr(), sym_count=cnt.data_ptr(), Kmax=K)
        plan = eng.plan(Bp, self.training, do_loss)   # (building a plan fixes the layouts of the halo operands it is the first to use)
        if staging is None:
            eng.repack()
        return eng, plan, kctx

    def forward(
        self,
        x,
        gt_xyz=None,
        gt_xyz_bin=None,
        gt_mask_trunc=None,
        gt_mask_visib=None,
        gt_mask_obj=None,
        gt_region=None,
        gt_allo_quat=None,
        gt_ego_quat=None,
        gt_allo_rot6d=None,
        gt_ego_rot6d=None,
        gt_ego_rot=None,
        gt_points=None,
        sym_infos=None,
        gt_trans=None,
        gt_trans_ratio=None,
        roi_classes=None,
        roi_coord_2d=None,
        roi_cams=None,
        roi_centers=None,
        roi_whs=None,
        roi_extents=None,
        resize_ratios=None,
        do_loss=False,
    ):
        a = dict(gt_xyz=gt_xyz, gt_mask_trunc=gt_mask_trunc, gt_mask_visib=gt_mask_visib, gt_region=gt_region, gt_ego_rot=gt_ego_rot,
                 gt_points=gt_points, sym_infos=sym_infos, gt_trans=gt_trans, gt_trans_ratio=gt_trans_ratio, roi_coord_2d=roi_coord_2d,
                 roi_cams=roi_cams, roi_centers=roi_centers, roi_whs=roi_whs, roi_extents=roi_extents, resize_ratios=resize_ratios)
        cfg = self.cfg
        eng, plan, kctx = self._prepare(x, do_loss, a)
        dev, B = eng.dev, x.shape[0]

        if not do_loss:  # test
            # (the fused output-conv + tail kernel writes the fp32 logits only for callers that return the maps -- GDRN.py:183-190 / `_maps` below)
            kctx["want_maps"] = bool(cfg.TEST.USE_PNP)
            # the pose kernel writes into buffers that belong to THIS call (the plan's attributes are re-bound to them; the kernel reads the
            # pointers at launch): no copy launches behind the forward pass (r6: two 5 us device copies closed every inference call)
            plan.rot, plan.trans = torch.empty_like(plan.rot), torch.empty_like(plan.trans)
            with torch.no_grad():
                plan.run_forward(kctx)
            out_dict = {"rot": plan.rot[:B], "trans": plan.trans[:B]}   # (plan.B >= B: padded inference batch)
            if cfg.TEST.USE_PNP:
                out_dict.update(self._maps(plan, B))
            return out_dict

        need_grad = torch.is_grad_enabled() and plan.has_backward
        if torch.is_grad_enabled() and not plan.has_backward and any(p.requires_grad for p in eng.P.values()):
            logger.warning("do_loss=True in eval mode (model.eval()): the losses carry no grad_fn -- the backward graph needs model.train()")
        if need_grad:
            losses = _PathFn.apply(self, plan, kctx, *[eng.P[n] for n in eng.param_names])
        else:
            plan.run_forward(kctx)
            losses = plan.losses.clone()
        losses = losses * self._loss_w
        loss_dict = {k: losses[i] for i, k in enumerate(LOSS_NAMES)}
        self._put_vis(plan, kctx, gt_trans, gt_trans_ratio, dev)
        return {}, loss_dict

    # -------------------------------------------------------------------------------------------
    def train_step(self, x, optimizer=None, loss_weights=None, **kw):
        """One fused training step without the autograd round trip: forward + losses + backward
        (+ overlapped RCCL gradient all-reduce when dist.attach()ed) (+ fused optimizer step reading the
        engine's flat gradient buffer).  Same arithmetic as ``loss_dict = model(...); sum(loss_dict.values()).backward();
        optimizer.step()`` (core/gdrn_modeling/engine.py:244-280).  Returns the [8] tensor of weighted losses (device,
        order engine.LOSS_NAMES; the values of forward()'s loss_dict)."""
        assert self.training, "train_step needs model.train()"
        a = dict(gt_xyz=None, gt_mask_trunc=None, gt_mask_visib=None, gt_region=None, gt_ego_rot=None, gt_points=None, sym_infos=None,
                 gt_trans=None, gt_trans_ratio=None, roi_coord_2d=None, roi_cams=None, roi_centers=None, roi_whs=None, roi_extents=None,
                 resize_ratios=None)
        a.update({k: v for k, v in kw.items() if k in a})
        if loss_weights is None and self._on_bucket is None and os.environ.get("GDRN_GRAPH", "0") == "1":
            out = self._train_step_graph(x, optimizer, a)
            if out is not None:
                return out
        eng, plan, kctx = self._prepare(x, True, a)
        eng = plan.e
        # (r6) dL/dloss_k is known BEFORE the forward pass here (unlike under autograd): written first, so that the pose kernel can leave dL/dfc
        # itself (no combine / cast launches in front of the backward pass) and the loss-finalize launch the weighted loss vector this returns
        seam = eng.fc_tail and eng.h16 and bool(getattr(plan, "acc_rows", 0))
        dyn = eng.loss_scale_dev()   # fp16, dynamic loss scale: its state on the device (None: bf16 / fp32 / fp16 with a static scale)
        ls = 1.0 if dyn is not None else eng._ls_host   # static loss scale on dL/dloss, divided out where the optimizer reads the gradients
        if dyn is not None:
            # dL/dloss_k = weight_k x the scale as the DEVICE holds it (it changes without the host knowing): one 8-thread launch per step
            lw2 = None if loss_weights is None else torch.as_tensor(loss_weights, dtype=torch.float32, device=eng.dev).contiguous()
            cabi.check(eng.lib.gdrn_scaled_loss_weights(self._loss_w.data_ptr(), None if lw2 is None else lw2.data_ptr(), 8, dyn.ptr, plan.gw.data_ptr(),
                                                        eng._stream()), "scaled_loss_weights")
            plan._gw_key = None
        elif loss_weights is not None:
            plan.gw.copy_(self._loss_w * loss_weights * ls)
            plan._gw_key = None
        elif getattr(plan, "_gw_key", None) != (self._loss_w.data_ptr(), self._loss_w._version, ls):
            plan.gw.copy_(self._loss_w * ls)  # dL/dloss_k = the config's loss weights: written once, not every step
            plan._gw_key = (self._loss_w.data_ptr(), self._loss_w._version, ls)
        if seam:
            out = torch.empty(8, dtype=torch.float32, device=eng.dev)   # this call's own: the kernel reads the pointer at launch
            kctx["seeded"], kctx["loss_w"], kctx["weighted"] = True, self._loss_w.data_ptr(), out.data_ptr()
        plan.run_forward(kctx)
        if not seam:
            # the returned losses (weighted like forward()'s loss_dict) are formed HERE, on the main stream in front of the backward chain, which
            # ends ~0.1 ms before the side stream does: behind the final join the one small launch and its gap were the step's last 15 us
            out = plan.losses * self._loss_w
        red = getattr(self, "_reducer", None)
        # The optimizer update of a gradient bucket goes out as soon as the bucket is final, under the rest of the backward pass
        # (Ranger.step_buckets_*), instead of behind the whole pass: on one GPU on the engine's side stream right behind the bucket's
        # weight-gradient reduction; with a GradReducer attached (dist.attach) on the reducer's stream right behind the bucket's all-reduce,
        # with the 1/world factor folded into the update.  The bucket's operand copies for the next step follow on the same stream.
        mine = red is not None and self._on_bucket == red.on_bucket and red.defer_scale and (red.cuda or not red.active)
        # (fp16 with the dynamic loss scale: the update waits for the finite check over ALL gradients -- a bucket applied under the backward
        #  pass could not be taken back when a later bucket overflows)
        guard = dyn is not None
        early = (optimizer is not None and (mine or (red is None and self._on_bucket is None)) and eng.wgrad_stream and not guard
                 and hasattr(optimizer, "step_buckets_begin") and os.environ.get("GDRN_EARLY_OPT", "1") != "0")
        if early:
            if getattr(eng, "_bucket_of", None) is None or getattr(eng, "_bucket_of_bounds", None) != tuple(eng.bucket_bounds):
                off = {id(eng.P[n]): eng.grad_offsets[n] for n in eng.param_names}
                eng._bucket_of = lambda p_, off=off, bb=eng.bucket_bounds: next(i for i, (lo, hi) in enumerate(bb) if lo <= off[id(p_)] < hi)
                eng._bucket_of_bounds = tuple(eng.bucket_bounds)
            early = optimizer.step_buckets_begin({eng.P[n]: eng.grads[n] for n in eng.param_names}, eng._bucket_of, len(eng.bucket_bounds),
                                                 grad_scale=(red.grad_scale if red is not None else 1.0) / ls)
        if early:
            packed = []

            def bucket_done(b):   # called on the side stream behind the bucket's weight-gradient reduction
                if red is not None and red.active:
                    red.on_bucket(b)                       # all-reduce on the reducer's stream behind an event of this stream
                    with torch.cuda.stream(red.stream):    # ... and behind it, in stream order, the update + the operand copies
                        optimizer.step_bucket(b)
                        packed.append(eng.repack_bucket(b))
                    return
                optimizer.step_bucket(b)
                packed.append(eng.repack_bucket(b))

            try:
                plan.run_backward(kctx, on_bucket=bucket_done)
            except BaseException:
                optimizer.step_buckets_abort()   # step counters untouched
                raise
            if red is not None:
                red.wait()   # the main stream (next forward, the caller) behind every bucket's exchange + update
            optimizer.step_buckets_end()
            if red is not None and red.active and red.world > 1:
                self._dp_divergence_check(eng)
            if packed and all(packed):
                eng.mark_packed()
            return out
        plan.run_backward(kctx, on_bucket=self._on_bucket)
        gs = 1.0 / ls
        if red is not None:
            red.wait()
            gs = red.grad_scale / ls  # 1/world, folded into the fused optimizer's gradient read
        if guard:
            # fp16 with the dynamic loss scale, decided on the device (r6): finite check over the flat gradient buffer (behind the all-reduce: every
            # rank decides alike) -> fused Ranger that is a no-op when the flag is up, divides by the scale and counts applied steps itself ->
            # GradScaler.update's bookkeeping as a one-thread launch.  No host read: the host keeps launching (r5 read the flag every step).
            eng = plan.e
            cabi.check(eng.lib.gdrn_nonfinite_flag(eng.grad_flat.data_ptr(), eng.grad_flat.numel(), dyn.ptr, eng._stream()), "nonfinite_flag")
            grads = {eng.P[n]: eng.grads[n] for n in eng.param_names}
            if optimizer is not None and hasattr(optimizer, "step_dyn") and optimizer.step_dyn(grads, gs, dyn):
                cabi.check(eng.lib.gdrn_loss_scale_update(dyn.ptr, 1, eng._stream()), "loss_scale_update")
                return out
            # gradients for an optimizer that is not the fused Ranger (or none): unscaled, zeroed if the step overflowed; such an optimizer's
            # step is skipped on the host's reading of the state (GradScaler.step does the same read)
            self._unscale_for_external_optimizer(eng, dyn, gs)
            if optimizer is not None and not dyn.read()["last_overflowed"]:
                optimizer.step(grads=grads) if getattr(optimizer, "takes_grad_scale", False) else self._step_foreign(optimizer, grads)
            return out
        if optimizer is not None:
            eng = plan.e
            grads = {eng.P[n]: eng.grads[n] for n in eng.param_names}
            if gs != 1.0 and not getattr(optimizer, "takes_grad_scale", False):
                eng.grad_flat.mul_(gs)
                gs = 1.0
            if gs != 1.0:
                optimizer.step(grads=grads, grad_scale=gs)
            else:
                optimizer.step(grads=grads)
        elif gs != 1.0:
            plan.e.grad_flat.mul_(gs)
        return out

    @staticmethod
    def _unscale_for_external_optimizer(eng, dyn, factor):
        """fp16, dynamic loss scale, gradients leaving the engine (autograd's .grad, train_step(optimizer=None)): finite check, then g <- g * factor /
        scale or 0 when the pass overflowed (no inf / NaN reaches an optimizer -- ADVICE r5), then the scale's bookkeeping (a skipped step halves it)."""
        st = eng._stream()
        cabi.check(eng.lib.gdrn_nonfinite_flag(eng.grad_flat.data_ptr(), eng.grad_flat.numel(), dyn.ptr, st), "nonfinite_flag")
        cabi.check(eng.lib.gdrn_unscale_or_zero(eng.grad_flat.data_ptr(), eng.grad_flat.numel(), float(factor), dyn.ptr, st), "unscale_or_zero")
        cabi.check(eng.lib.gdrn_loss_scale_update(dyn.ptr, 0, st), "loss_scale_update")

    @staticmethod
    def _step_foreign(optimizer, grads):
        for p, g in grads.items():
            p.grad = g
        optimizer.step()

    def grad_overflowed(self):
        """did the last fp16 backward pass (dynamic loss scale) overflow?  Its gradients were zeroed / its fused optimizer step skipped.  Reads the
        device state (a host synchronisation)."""
        e = self._eng
        return bool(e is not None and e.ls_state is not None and e.ls_state.read()["last_overflowed"])

    def _dp_divergence_check(self, eng):
        """Data-parallel safety net of the per-bucket optimizer (the update of a bucket runs on the reducer's stream right behind its all-reduce,
        ADVICE r3: that path has only run with one real rank).  Ranks that apply the same summed gradients hold bit-identical parameters; on the
        first GDRN_DP_CHECK_STEPS (default 2) data-parallel steps every rank compares a checksum of its parameters with the other ranks' and
        raises instead of training on silently diverged replicas.  Two host synchronisations in the life of a run."""
        n = self.__dict__.get("_dp_checked", 0)
        if n >= int(os.environ.get("GDRN_DP_CHECK_STEPS", "2")):
            return
        import torch.distributed as dist

        if not dist.is_initialized():
            return   # (a reducer with an explicit world size and no process group: the emulated-rank tests)
        self.__dict__["_dp_checked"] = n + 1

        red = self._reducer
        with torch.no_grad():
            cs = torch.stack([eng.P[k].detach().double().sum() for k in eng.param_names] +
                             [eng.P[k].detach().double().abs().sum() for k in eng.param_names]).sum(0, keepdim=True)
        got = [torch.empty_like(cs) for _ in range(red.world)]
        dist.all_gather(got, cs, group=red.group)
        vals = [float(g.item()) for g in got]
        if any(v != vals[0] for v in vals) or not all(v == v for v in vals):
            raise cabi.GdrnHipError(f"data-parallel replicas diverged after step {n + 1}: parameter checksums {vals} "
                               "(set GDRN_EARLY_OPT=0 to run the optimizer behind the whole backward pass)")

    def _train_step_graph(self, x, optimizer, a):
        """The step as ONE hipGraph replay: operand repack + forward + losses + backward (~350 kernel launches, many of
        them 5-20 us long).  Opt-in (GDRN_GRAPH=1): on an otherwise idle host the Python launch loop keeps up with the GPU
        at bs=64 (9.90 vs 9.95 ms/step measured), the replay pays off when the host is busy or the batch is small.  Inputs are copied into
        persistent buffers, the graph is captured on the third call for a batch size (after two eager warm-up steps) and
        replayed from then on; the optimizer step stays outside (its scalars change every step).  Returns None when the
        eager path has to run (warm-up, symmetric objects with a changing symmetry count, ...)."""
        B = int(x.shape[0])
        if self.engine().loss_scale_dynamic or self.engine()._ls_host != 1.0:
            return None   # fp16: dL/dloss carries a loss scale that changes between steps and the optimizer waits for the finite check -- eager path
        st = self.__dict__.setdefault("_graph_state", {}).setdefault(B, dict(calls=0, staging={}, graph=None, plan=None, ok=True))
        if not st["ok"]:
            return None
        st["calls"] += 1
        if st["calls"] <= 2:
            return None
        try:
            eng, plan, kctx = self._prepare(x, True, a, staging=st["staging"])
        except _StagingShapeChanged:
            st["ok"] = False  # e.g. a different number of symmetry transforms: stay on the eager path for this size
            return None
        if st["graph"] is None:
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                eng.repack(force=True)
                plan.run_forward(kctx)
                plan.gw.copy_(self._loss_w)
                plan.run_backward(kctx, on_bucket=None)
            st.update(graph=g, plan=plan, kctx=kctx)
        else:
            ref = st["kctx"]  # the captured launches have these addresses / counts baked in
            if plan is not st["plan"] or set(kctx) != set(ref) or any(kctx[k] != ref[k] for k in kctx if k != "_keep"):
                st["ok"] = False
                return None
        st["graph"].replay()
        eng.bn_epoch += 1  # the replayed kernels moved the BatchNorm running statistics (run_forward was not called)
        plan.generation += 1  # ... and overwrote the plan's activations: a pending loss.backward() of an earlier model(...) call must not use them
        if optimizer is not None:
            optimizer.step(grads={eng.P[n]: eng.grads[n] for n in eng.param_names})
        return plan.losses * self._loss_w  # weighted like forward()'s loss_dict

    def _maps(self, plan, B):
        """mask / coor_x / coor_y / coor_z / region as NCHW fp32 (GDRN.py:235-237)."""
        e, lib = plan.e, plan.e.lib
        if not getattr(plan, "head_valid", True):
            raise cabi.GdrnHipError("the head logits of this forward pass were not written (inference without cfg.TEST.USE_PNP hands the fused head "
                                    "kernel head = NULL): plan.head_out holds an earlier call's values")
        st = e._stream()
        C_ = e.head_c
        full = torch.empty(B, C_, 64, 64, dtype=torch.float32, device=e.dev)
        cabi.check(lib.gdrn_nhwc_to_nchw_f32(plan.head_out.data_ptr(), plan.hs, 0, C_, full.data_ptr(), B, 4096, cabi.F32, st), "nhwc_to_nchw")
        return {"mask": full[:, 0:1], "coor_x": full[:, 1:2], "coor_y": full[:, 2:3], "coor_z": full[:, 3:4], "region": full[:, 4:]}

    def _put_vis(self, plan, kctx, gt_trans, gt_trans_ratio, dev):
        """vis/* scalars of GDRN.py:246-303.  Values stay on the device (``self.last_vis``); they are
        fetched (one host sync) only when a detectron2 EventStorage is active, as in the reference."""
        vis = torch.cat([plan.vis.mean(0), plan.trans[0], plan.fc_out[0, 6:9]])
        self.last_vis = (vis, gt_trans, gt_trans_ratio)
        if _d2_get_event_storage is None:
            return
        try:
            storage = _d2_get_event_storage()
        except Exception:
            return
        storage.put_scalars(**self.vis_dict())

    def vis_dict(self):
        """Materialise the reference's vis/* dictionary (host sync)."""
        vis, gt_trans, gt_trans_ratio = self.last_vis
        v = vis.detach().cpu().numpy().astype(np.float64)
        gt, gtr = gt_trans[0].detach().cpu().numpy(), gt_trans_ratio[0].detach().cpu().numpy()
        return {
            "vis/error_R": float(v[0]), "vis/error_t": float(v[1]) * 100,
            "vis/error_tx": abs(v[2] - gt[0]) * 100, "vis/error_ty": abs(v[3] - gt[1]) * 100, "vis/error_tz": abs(v[4] - gt[2]) * 100,
            "vis/tx_pred": v[2], "vis/ty_pred": v[3], "vis/tz_pred": v[4],
            "vis/tx_net": v[5], "vis/ty_net": v[6], "vis/tz_net": v[7],
            "vis/tx_gt": float(gt[0]), "vis/ty_gt": float(gt[1]), "vis/tz_gt": float(gt[2]),
            "vis/tx_rel_gt": float(gtr[0]), "vis/ty_rel_gt": float(gtr[1]), "vis/tz_rel_gt": float(gtr[2]),
        }


def build_model_optimizer(cfg):
    """GDRN.py:550-724: ResNet-34 backbone, RotWithRegionHead, ConvPnPNet, 3 parameter groups
    (backbone / rot head / pnp with LR_MULT), optimizer from cfg.SOLVER.OPTIMIZER_CFG."""
    backbone_cfg = cfg.MODEL.CDPN.BACKBONE
    r_head_cfg = cfg.MODEL.CDPN.ROT_HEAD
    pnp_net_cfg = cfg.MODEL.CDPN.PNP_NET
    _check_supported(cfg)
    assert "resnet" in backbone_cfg.ARCH
    params_lr_list = []
    _, layers, channels, _ = resnet_spec[backbone_cfg.NUM_LAYERS]
    backbone_net = ResNetBackboneNet(layers, backbone_cfg.INPUT_CHANNEL, freeze=backbone_cfg.FREEZE, rot_concat=r_head_cfg.ROT_CONCAT)
    params_lr_list.append({"params": [p for p in backbone_net.parameters() if p.requires_grad], "lr": float(cfg.SOLVER.BASE_LR)})
    r_out_dim, mask_out_dim, region_out_dim = get_xyz_mask_region_out_dim(cfg)
    rot_head_net = RotWithRegionHead(channels[-1], r_head_cfg.NUM_FILTERS, rot_output_dim=r_out_dim, mask_output_dim=mask_out_dim,
                                     num_regions=r_head_cfg.NUM_REGIONS)
    params_lr_list.append({"params": [p for p in rot_head_net.parameters() if p.requires_grad], "lr": float(cfg.SOLVER.BASE_LR)})
    pnp_net_in_channel = r_out_dim + 2 + r_head_cfg.NUM_REGIONS
    pnp_head_cfg = pnp_net_cfg.PNP_HEAD_CFG
    pnp_head_type = pnp_head_cfg.pop("type")
    if pnp_head_type != "ConvPnPNet":
        raise NotImplementedError(f"pnp head {pnp_head_type} is not on the hot path")
    pnp_head_cfg.update(nIn=pnp_net_in_channel, rot_dim=6, num_regions=r_head_cfg.NUM_REGIONS, featdim=128, num_layers=3,
                        mask_attention_type=pnp_net_cfg.MASK_ATTENTION)
    pnp_net = ConvPnPNet(**pnp_head_cfg)
    params_lr_list.append({"params": [p for p in pnp_net.parameters() if p.requires_grad],
                           "lr": float(cfg.SOLVER.BASE_LR) * pnp_net_cfg.LR_MULT})
    model = GDRN(cfg, backbone_net, rot_head_net, trans_head_net=None, pnp_net=pnp_net)
    optimizer = build_optimizer_with_params(cfg, params_lr_list)
    if cfg.MODEL.WEIGHTS == "":
        pre = cfg.MODEL.CDPN.BACKBONE.get("PRETRAINED", "")
        if pre == "":
            logger.warning("Randomly initialize weights for backbone!")
        elif os.path.exists(pre) or _resolve_pretrained(pre) is not None:
            sd = torch.load(pre if os.path.exists(pre) else _resolve_pretrained(pre), map_location="cpu")
            sd = sd.get("state_dict", sd)
            sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}  # as mmcv's load_checkpoint (GDRN.py:721)
            model.backbone.load_state_dict(sd, strict=False)
        else:
            # the reference resolves "torchvision://resnet34" / http(s) URLs through mmcv's load_checkpoint; silently training
            # from the std=0.001 random init instead would change convergence -- ask for a local file (as checkpoint.py does)
            raise FileNotFoundError(f"cfg.MODEL.CDPN.BACKBONE.PRETRAINED={pre!r} is neither a local file nor in the torch hub cache "
                                    f"({os.path.join(_hub_dir(), 'checkpoints')}; URL schemes need network access): download the weights and point "
                                    "PRETRAINED at the file, or set it to '' for random init")
    model.to(torch.device(cfg.MODEL.DEVICE))
    return model, optimizer


def _hub_dir():
    return os.path.join(os.environ.get("TORCH_HOME", os.path.join(os.path.expanduser("~"), ".cache", "torch")), "hub")


def _resolve_pretrained(pre):
    """"torchvision://resnet34" (configs/_base_/gdrn_base.py:21) -> the file torch hub would have downloaded, if it is in the local hub cache
    ($TORCH_HOME/hub/checkpoints/resnet34-<hash>.pth) and unambiguous; None otherwise (no network access here).  Only the torchvision scheme:
    mmcv's "open-mmlab://" / "modelzoo://" zoo holds DIFFERENT (caffe-style) weights under the same model names -- mapping those names onto a
    torchvision file would load the wrong backbone silently (strict=False), so they fall through to the FileNotFoundError."""
    import glob

    scheme = "torchvision://"
    if not pre.startswith(scheme):
        return None
    hits = sorted(glob.glob(os.path.join(_hub_dir(), "checkpoints", pre[len(scheme):] + "-*.pth")))
    if len(hits) > 1:
        logger.warning("several hub-cache files match %s: %s -- point PRETRAINED at the one you mean", pre, hits)
        return None
    return hits[0] if hits else None


def build_optimizer_with_params(cfg, params):
    """core/utils/solver_utils.py:47-57 for the optimizers the shipped configs name (Ranger; plus the
    torch.optim classes by name)."""
    ocfg = dict(cfg.SOLVER.OPTIMIZER_CFG)
    typ = ocfg.pop("type")
    ocfg.pop("_delete_", None)
    if typ.lower() == "ranger":
        from .ranger import Ranger

        return Ranger(params, **ocfg)
    return getattr(torch.optim, typ)(params, **ocfg)
