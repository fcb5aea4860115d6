"""GPU RoI cropper / target builder: the step immediately before the per-RoI hot path.

Host-side mirror of the crop / target part of the reference's ``GDRN_DatasetFromList.read_data``
(core/gdrn_modeling/data_loader.py:411-444 test mode, :460-545 and :617-632 train mode) and of
``crop_resize_by_warp_affine`` (core/utils/data_utils.py:80-92), batched: the frames, the per-instance
object-coordinate patches and the masks are device tensors, and ONE ``gdrn_roi_affine`` + ``gdrn_roi_crop_inputs``
(+ ``gdrn_roi_targets``) launch produces the whole batch in the layout ``GDRN.forward`` takes
(``roi_img`` [B,3,256,256], ``roi_coord_2d`` [B,2,64,64], ``roi_xyz`` [B,3,64,64], ``roi_mask_*`` [B,64,64],
``roi_region`` [B,64,64], ``roi_wh``, ``resize_ratio``, ``trans_ratio`` ...), so at >20 k RoI/s per GPU the batch
never has to pass through cv2 on 4 CPU workers (common_base.py:87).

What stays on the host, as in the reference: file decoding, colour / background augmentation, the random DZI box
jitter (``aug_bbox``, base_data_loader.py:120-152 -- its *result* ``bbox_center`` / ``scale`` is an input here) and
the scalar per-instance bookkeeping.  ``INPUT.SMOOTH_XYZ`` / ``TRAIN.VIS`` (median-blurred / bilinear xyz) and the
classification (``CE``) xyz targets are not on the path and raise ``NotImplementedError``.  There is no CPU fallback.
"""
import ctypes as C

import numpy as np
import torch

from . import cabi


def get_2d_coord_np(width, height, low=0, high=1, fmt="CHW"):
    """core/utils/data_utils.py:222-241 (host, once per frame size)."""
    x = np.linspace(low, high, width, dtype=np.float32)
    y = np.linspace(low, high, height, dtype=np.float32)
    xy = np.asarray(np.meshgrid(x, y))
    if fmt == "HWC":
        return xy.transpose(1, 2, 0)
    if fmt == "CHW":
        return xy
    raise ValueError(f"Unknown format: {fmt}")


def aug_bbox(cfg, bbox_xyxy, im_H, im_W, rng=np.random):
    """Dynamic zoom-in of a training box (core/base_data_loader.py:120-152): returns (bbox_center, scale), the crop centre and
    square crop size ``RoiCropper`` takes.  ``rng`` needs ``random_sample`` / ``rand`` (numpy's global generator by default, as
    in the reference, so a seeded run draws the same numbers in the same order)."""
    x1, y1, x2, y2 = np.asarray(bbox_xyxy).copy()
    cx, cy = 0.5 * (x1 + x2), 0.5 * (y1 + y2)
    bh, bw = y2 - y1, x2 - x1
    inp = cfg.INPUT
    dzi = inp.DZI_TYPE.lower()
    if dzi == "uniform":
        scale_ratio = 1 + inp.DZI_SCALE_RATIO * (2 * rng.random_sample() - 1)
        shift_ratio = inp.DZI_SHIFT_RATIO * (2 * rng.random_sample(2) - 1)
        bbox_center = np.array([cx + bw * shift_ratio[0], cy + bh * shift_ratio[1]])
        scale = max(y2 - y1, x2 - x1) * scale_ratio * inp.DZI_PAD_SCALE
    elif dzi == "roi10d":
        lo, hi = -0.15, 0.15  # every corner coordinate moves by up to 15 % of the box size
        x1 += bw * (rng.rand() * (hi - lo) + lo)
        x2 += bw * (rng.rand() * (hi - lo) + lo)
        y1 += bh * (rng.rand() * (hi - lo) + lo)
        y2 += bh * (rng.rand() * (hi - lo) + lo)
        x1 = min(max(x1, 0), im_W)
        x2 = min(max(x1, 0), im_W)  # the reference clamps x2 from x1 (base_data_loader.py:141); kept for identical crops
        y1 = min(max(y1, 0), im_H)
        y2 = min(max(y2, 0), im_H)
        bbox_center = np.array([0.5 * (x1 + x2), 0.5 * (y1 + y2)])
        scale = max(y2 - y1, x2 - x1) * inp.DZI_PAD_SCALE
    elif dzi == "truncnorm":
        raise NotImplementedError("DZI truncnorm not implemented yet.")
    else:
        bbox_center = np.array([cx, cy])
        scale = max(y2 - y1, x2 - x1)
    return bbox_center, min(scale, max(im_H, im_W)) * 1.0


def detection_box_to_roi(cfg, bbox_xyxy, im_H, im_W):
    """Test-mode crop geometry of a detection box (core/gdrn_modeling/data_loader.py:411-423): (bbox_center, scale)."""
    x1, y1, x2, y2 = bbox_xyxy
    bbox_center = np.array([0.5 * (x1 + x2), 0.5 * (y1 + y2)])
    scale = max(max(y2 - y1, 1), max(x2 - x1, 1)) * cfg.INPUT.DZI_PAD_SCALE
    return bbox_center, min(scale, max(im_H, im_W)) * 1.0


def _dev(t, dtype, device, what):
    if not isinstance(t, torch.Tensor) or t.device.type != "cuda":
        raise cabi.GdrnHipError(f"{what} must be a device tensor: the RoI cropper runs on the GPU (no CPU fallback)")
    if t.dtype == torch.bool and dtype == torch.uint8:
        t = t.view(torch.uint8) if t.is_contiguous() else t.to(torch.uint8)
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.to(device).contiguous()


class RoiCropper:
    """Batched RoI inputs / targets on the device.

    ``cfg`` is the reference config (``MODEL.PIXEL_MEAN/PIXEL_STD``, ``MODEL.CDPN.BACKBONE.INPUT_RES/OUTPUT_RES``,
    ``MODEL.CDPN.ROT_HEAD.NUM_REGIONS``); ``extents`` [ncls,3] and ``fps_points`` [ncls,nfps,3] are what
    ``_get_extents`` / ``_get_fps_points`` (data_loader.py:189-233) return, indexed by ``roi_cls``."""

    def __init__(self, cfg, extents=None, fps_points=None, device=None):
        self.lib = cabi.load()  # raises when libgdrn_hip.so is missing
        m = cfg.MODEL
        rh = m.CDPN.ROT_HEAD
        if cfg.get("INPUT", {}).get("SMOOTH_XYZ", False) or cfg.get("TRAIN", {}).get("VIS", False):
            raise NotImplementedError("INPUT.SMOOTH_XYZ / TRAIN.VIS targets are not on the MI355X path")
        if "CE" in rh.XYZ_LOSS_TYPE or "cls" in m.CDPN.NAME:
            raise NotImplementedError(f"classification xyz targets are not on the MI355X path: {rh.XYZ_LOSS_TYPE}")
        self.input_res = int(m.CDPN.BACKBONE.INPUT_RES)
        self.out_res = int(m.CDPN.BACKBONE.OUTPUT_RES)
        self.num_regions = int(rh.NUM_REGIONS)
        mean = [float(v) for v in m.get("PIXEL_MEAN", [0, 0, 0])]
        std = [float(v) for v in m.get("PIXEL_STD", [255.0, 255.0, 255.0])]
        self._mean, self._std = (C.c_double * 3)(*mean), (C.c_double * 3)(*std)
        self.device = torch.device(device or m.get("DEVICE", "cuda"))
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.extents = None if extents is None else torch.as_tensor(np.asarray(extents, np.float32)).to(self.device).contiguous()
        self.fps_points = None
        if fps_points is not None:
            fps = np.asarray(fps_points, np.float64)
            if fps.ndim != 3 or fps.shape[2] != 3:
                raise ValueError("fps_points must be [num_classes, num_fps, 3]")
            self.fps_points = torch.as_tensor(fps).to(self.device).contiguous()
        self._coord2d = {}

    def coord_2d(self, H, W):
        """The frame-sized 2D coordinate map the reference builds per image (data_loader.py:367), cached on the device."""
        key = (int(H), int(W))
        if key not in self._coord2d:
            self._coord2d[key] = torch.as_tensor(np.ascontiguousarray(get_2d_coord_np(W, H, fmt="HWC"))).to(self.device)
        return self._coord2d[key]

    # ------------------------------------------------------------------------------------------------------------
    def _tasks(self, rois, train):
        keep, tasks = [], []
        for r in rois:
            img = _dev(r["image"], torch.uint8, self.device, "image")
            if img.dim() != 3 or img.shape[2] != 3:
                raise ValueError("image must be [H, W, 3] uint8")
            H, W = int(img.shape[0]), int(img.shape[1])
            c2d = self.coord_2d(H, W)
            cx, cy = (float(v) for v in r["bbox_center"])
            scale = float(r["scale"])
            if not scale > 0:
                raise ValueError("scale must be positive")
            x1, y1, x2, y2 = (float(v) for v in r["bbox"])
            t = cabi.RoiTask(image=cabi.ptr(img), coord2d=cabi.ptr(c2d), cx=cx, cy=cy, scale=scale, bw=max(x2 - x1, 1), bh=max(y2 - y1, 1), H=H, W=W,
                             x1=0, y1=0, x2=-1, y2=-1, cls=int(r.get("roi_cls", 0)))
            keep += [img, c2d]
            if train:
                xyz = _dev(r["xyz_crop"], torch.float32, self.device, "xyz_crop")
                xx1, yy1, xx2, yy2 = (int(v) for v in r["xyxy"])
                if xyz.shape != (yy2 - yy1 + 1, xx2 - xx1 + 1, 3) or xx1 < 0 or yy1 < 0 or xx2 >= W or yy2 >= H:
                    raise ValueError("xyz_crop does not match xyxy / the frame")  # numpy raises on the same paste (data_loader.py:468)
                seg = _dev(r["segmentation"], torch.uint8, self.device, "segmentation")
                if seg.shape != (H, W):
                    raise ValueError("segmentation must be [H, W]")
                t.xyz_crop, t.seg = cabi.ptr(xyz), cabi.ptr(seg)
                keep += [xyz, seg]
                if r.get("mask_trunc") is not None:
                    tr = _dev(r["mask_trunc"], torch.uint8, self.device, "mask_trunc")
                    if tr.shape != (H, W):
                        raise ValueError("mask_trunc must be [H, W]")
                    t.trunc = cabi.ptr(tr)
                    keep.append(tr)
                t.x1, t.y1, t.x2, t.y2 = xx1, yy1, xx2, yy2
                ox, oy = (float(v) for v in r["centroid_2d"])
                t.ox, t.oy, t.tz = ox, oy, float(r["trans"][2])
                if not 0 <= t.cls < (self.extents.shape[0] if self.extents is not None else 0):
                    raise ValueError("roi_cls outside the extents table")
            tasks.append(t)
        return tasks, keep

    def prepare(self, rois, train=False):
        """Validate ``rois`` and upload the per-RoI task table (host work a loader thread can do ahead of time).
        ``rois``: list of per-instance dicts -- ``image`` (device u8 [H,W,3]), ``bbox_center`` (2), ``scale``, ``bbox``
        (xyxy), and for ``train=True`` also ``xyz_crop`` (device fp32 [h,w,3]), ``xyxy``, ``segmentation`` (device u8/bool
        [H,W]), optional ``mask_trunc``, ``roi_cls``, ``trans`` (3), ``centroid_2d`` (2)."""
        if len(rois) == 0:
            raise ValueError("empty RoI batch")
        if train and (self.extents is None or (self.num_regions > 1 and self.fps_points is None)):
            raise ValueError("train-mode targets need the extents (and fps_points) tables")
        if train and self.num_regions > 1 and self.fps_points.shape[1] != self.num_regions:
            raise ValueError("fps_points must hold NUM_REGIONS points per class")
        tasks, keep = self._tasks(rois, train)
        return dict(tasks=tasks, keep=keep, tab=cabi.to_device_table(tasks, self.device), train=train)

    def launch(self, prep):
        """The three launches for a prepared batch; returns the batch dict with the reference's keys (device tensors)."""
        tasks, tab, train = prep["tasks"], prep["tab"], prep["train"]
        B = len(tasks)
        dev, ir, orr = self.device, self.input_res, self.out_res
        f32 = dict(dtype=torch.float32, device=dev)
        minv = torch.empty(B, 2, 6, dtype=torch.float64, device=dev)
        out = dict(roi_img=torch.empty(B, 3, ir, ir, **f32), roi_coord_2d=torch.empty(B, 2, orr, orr, **f32), roi_wh=torch.empty(B, 2, **f32),
                   resize_ratio=torch.empty(B, **f32))
        trans_ratio = torch.empty(B, 3, **f32) if train else None
        st = torch.cuda.current_stream(dev).cuda_stream
        lib, p = self.lib, cabi.ptr
        cabi.check(lib.gdrn_roi_affine(p(tab), B, ir, orr, p(minv), p(out["roi_wh"]), p(out["resize_ratio"]), p(trans_ratio), st), "roi_affine")
        cabi.check(lib.gdrn_roi_crop_inputs(p(tab), p(minv), B, ir, orr, self._mean, self._std, p(out["roi_img"]), p(out["roi_coord_2d"]), st),
                   "roi_crop_inputs")
        if train:
            out.update(roi_xyz=torch.empty(B, 3, orr, orr, **f32), roi_mask_trunc=torch.empty(B, orr, orr, **f32),
                       roi_mask_visib=torch.empty(B, orr, orr, **f32), roi_mask_obj=torch.empty(B, orr, orr, **f32), trans_ratio=trans_ratio)
            region = torch.empty(B, orr, orr, dtype=torch.int32, device=dev) if self.num_regions > 1 else None
            cabi.check(lib.gdrn_roi_targets(p(tab), p(minv), B, orr, p(self.fps_points) if region is not None else None, self.num_regions, p(self.extents),
                                            p(out["roi_xyz"]), p(out["roi_mask_trunc"]), p(out["roi_mask_visib"]), p(out["roi_mask_obj"]), p(region), st),
                       "roi_targets")
            if region is not None:
                out["roi_region"] = region
            cls = torch.as_tensor([t.cls for t in tasks], dtype=torch.long, device=dev)
            out["roi_cls"], out["roi_extent"] = cls, self.extents[cls]
        out["bbox_center"] = torch.as_tensor([[t.cx, t.cy] for t in tasks], **f32)
        out["scale"] = torch.as_tensor([t.scale for t in tasks], **f32)
        return out

    def __call__(self, rois, train=False):
        """``launch(prepare(rois, train))``.  The source tensors referenced by the table are only read by launches on the
        current stream, so dropping them afterwards is safe (stream-ordered reuse)."""
        return self.launch(self.prepare(rois, train))
