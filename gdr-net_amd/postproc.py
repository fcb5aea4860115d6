"""Inference post-processing on the device.

Host-side mirror of what the reference's evaluator runs per instance on the CPU between the network and
cv2's PnP-RANSAC (``GDRN_Evaluator.process_pnp_ransac``, core/gdrn_modeling/gdrn_evaluator.py:316-377):

* ``get_out_coor(cfg, coor_x, coor_y, coor_z)``   -- engine_utils.py:92-105
* ``get_out_mask(cfg, pred_mask)``                -- engine_utils.py:108-126
* ``get_img_model_points_with_coords2d``          -- gdrn_evaluator.py:89-126, batched here

all served by ONE kernel launch for the whole batch (``gdrn_correspondences``): no device->host copy of the dense
maps, no Python loop over instances.  Same names / argument meaning / error behaviour as the reference for the
configurations on the path (L1 xyz / mask heads); other loss types raise ``NotImplementedError`` like an
unsupported config elsewhere in this package.  There is no CPU fallback.
"""
import torch

from . import cabi


def _check_l1(cfg):
    rh = cfg.MODEL.CDPN.ROT_HEAD
    if rh.MASK_LOSS_TYPE != "L1":
        raise NotImplementedError(f"unknown mask loss type on the MI355X path: {rh.MASK_LOSS_TYPE}")


def _maps(t):
    if t.device.type != "cuda":
        raise cabi.GdrnHipError("post-processing runs on the GPU (no CPU fallback)")
    return t.detach().to(torch.float32).contiguous()


def _run(mask, coor_x, coor_y, coor_z, coord2d=None, extents=None, im_hw=None, mask_thr=0.5, want_points=False):
    lib = cabi.load()
    mask, coor_x, coor_y, coor_z = _maps(mask), _maps(coor_x), _maps(coor_y), _maps(coor_z)
    N, c, H, W = mask.shape
    if c != 1 or coor_x.shape[1] != 1 or coor_y.shape[1] != 1 or coor_z.shape[1] != 1:
        raise NotImplementedError("only the one-channel (L1) mask / coordinate heads are on the MI355X path")
    dev, HW = mask.device, H * W
    out_mask = torch.empty(N, 1, H, W, dtype=torch.float32, device=dev)
    out_xyz = torch.empty(N, 3, H, W, dtype=torch.float32, device=dev)
    counts = torch.zeros(N, dtype=torch.int32, device=dev)
    if extents is None:
        extents = torch.ones(N, 3, dtype=torch.float32, device=dev)
    if im_hw is None:
        im_hw = torch.ones(N, 2, dtype=torch.float32, device=dev)
    extents, im_hw = _maps(extents), _maps(im_hw)
    img = mod = c2d = None
    if want_points:
        c2d = _maps(coord2d)
        img = torch.empty(N, HW, 2, dtype=torch.float32, device=dev)
        mod = torch.empty(N, HW, 3, dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    cabi.check(lib.gdrn_correspondences(cabi.ptr(mask), cabi.ptr(coor_x), cabi.ptr(coor_y), cabi.ptr(coor_z), HW, 1, cabi.ptr(c2d),
                                        cabi.ptr(extents), cabi.ptr(im_hw), float(mask_thr), N, HW, cabi.ptr(out_mask), cabi.ptr(out_xyz),
                                        cabi.ptr(img), cabi.ptr(mod), cabi.ptr(counts), st), "correspondences")
    return out_mask, out_xyz, img, mod, counts


def get_out_coor(cfg, coor_x, coor_y, coor_z):
    """[N,3,H,W] fp32: the three predicted coordinate maps side by side (engine_utils.py:92-105)."""
    if not (coor_x.shape[1] == 1 and coor_y.shape[1] == 1 and coor_z.shape[1] == 1):
        raise NotImplementedError("classification (CE_coor) coordinate heads are outside the MI355X path")
    return _run(torch.zeros_like(coor_x), coor_x, coor_y, coor_z)[1]


def get_out_mask(cfg, pred_mask):
    """[N,1,H,W] fp32 in [0,1]: per-RoI min-max normalised mask (engine_utils.py:108-126, L1)."""
    _check_l1(cfg)
    bs, c, h, w = pred_mask.shape
    assert c == 1, c
    return _run(pred_mask, pred_mask, pred_mask, pred_mask)[0]


def get_img_model_points_with_coords2d(cfg, out_dict, roi_coord_2d, roi_extents, im_H, im_W, mask_thr=None):
    """Batched gdrn_evaluator.py:89-126.  out_dict: the model's eval output with cfg.TEST.USE_PNP (mask, coor_x/y/z);
    im_H / im_W: per-RoI image sizes (sequence or tensor of length N, or scalars).
    Returns (out_mask [N,1,H,W], out_xyz [N,3,H,W], image_points [N,HW,2], model_points [N,HW,3], counts [N] int32):
    the first counts[n] rows of RoI n are its 2D-3D correspondences in the reference's row-major order."""
    _check_l1(cfg)
    mask = out_dict["mask"]
    N, dev = mask.shape[0], mask.device
    hw = torch.empty(N, 2, dtype=torch.float32)
    hw[:, 0] = torch.as_tensor(im_H, dtype=torch.float32)
    hw[:, 1] = torch.as_tensor(im_W, dtype=torch.float32)
    thr = cfg.MODEL.CDPN.ROT_HEAD.MASK_THR_TEST if mask_thr is None else mask_thr
    return _run(mask, out_dict["coor_x"], out_dict["coor_y"], out_dict["coor_z"], coord2d=roi_coord_2d, extents=roi_extents,
                im_hw=hw.to(dev), mask_thr=thr, want_points=True)
