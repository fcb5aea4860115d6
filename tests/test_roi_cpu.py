"""CPU tests of the RoI cropper / target builder row: the oracle against golden G8 (outputs of
the reference's own xyz_to_region / get_2d_coord_np) and against size-independent properties of an affine crop; host-side
argument checking of the product module.  The cv2 part of the oracle is restated from OpenCV's published algorithm
("parity unpinned": OpenCV is not in the image) -- the property tests below bound it against exact-arithmetic resampling."""
import os

import numpy as np
import pytest
from scipy.ndimage import map_coordinates

from gdrnet_amd import synth
from oracle import roi_oracle as R

HERE = os.path.dirname(os.path.abspath(__file__))


def test_g8_xyz_to_region_matches_the_reference():
    g = np.load(os.path.join(HERE, "golden", "g8_roi_targets.npz"))
    inp = synth.make_region_inputs()
    for i in range(inp["xyz"].shape[0]):
        got = R.xyz_to_region(inp["xyz"][i], inp["fps_points"][i])
        assert np.array_equal(got.astype(np.int32), g[f"region{i}"]), i
    assert g["region1"][0, 0] == 4  # duplicated fps point: the first index wins
    assert not g["region2"].any()  # all background


def test_g8_coord2d_matches_the_reference():
    from gdrnet_amd import roi_data

    g = np.load(os.path.join(HERE, "golden", "g8_roi_targets.npz"))
    for fn in (R.get_2d_coord_np, roi_data.get_2d_coord_np):
        c = fn(640, 480, fmt="HWC")
        assert c.dtype == np.float32 and c.shape == (480, 640, 2)
        assert np.array_equal(c[0, :, 0], g["coord2d_640x480_row0"]) and np.array_equal(c[:, 0, 1], g["coord2d_640x480_col0"])
        assert fn(720, 540).astype(np.float64).sum() == g["coord2d_720x540_sum"][0]


def test_affine_transform_is_a_scale_and_shift():
    c, s = np.array([300.3, 200.7]), 187.3
    M = R.get_affine_transform(c, s, 0, 256)
    k = 256 / np.float32(s)
    assert np.allclose(M, [[k, 0, 128 - k * np.float32(c[0])], [0, k, 128 - k * np.float32(c[1])]], rtol=1e-6, atol=1e-4)
    Mi = R.cv_invert_affine(M)
    assert np.allclose(Mi[:, :2] @ M[:, :2], np.eye(2), atol=1e-12)
    assert np.allclose(R.get_affine_transform(c, s, 0, 256, inv=True), Mi, rtol=1e-6, atol=1e-4)


def test_identity_crop_copies_the_image():
    img = np.floor(synth.hash_uniform(3, "img", (96, 80, 3)) * 256).astype(np.uint8)
    f = synth.hash_uniform(3, "f", (96, 80, 2)).astype(np.float32)
    for a, interp in ((img, R.INTER_LINEAR), (f, R.INTER_LINEAR), (f, R.INTER_NEAREST)):
        out = R.crop_resize_by_warp_affine(a, np.array([32.0, 32.0]), 64.0, 64, interpolation=interp)
        assert np.array_equal(out, a[:64, :64])


@pytest.mark.parametrize("center,scale", [((300.3, 200.7), 187.3), ((20.0, 30.0), 150.0), ((630.0, 470.0), 77.7), ((320.0, 240.0), 640.0)])
def test_warp_tracks_exact_resampling(center, scale):
    """bilinear results stay within the 1/32-pixel quantisation of exact bilinear resampling; nearest picks round(x)."""
    from scipy.ndimage import gaussian_filter

    raw = synth.hash_uniform(4, "img", (480, 640, 3)) * 255
    smooth = gaussian_filter(raw, (4, 4, 0)).astype(np.float32)
    grad = max(np.abs(np.diff(smooth, axis=0)).max(), np.abs(np.diff(smooth, axis=1)).max())
    c = np.array(center)
    M = R.get_affine_transform(c, scale, 0, 256)
    Mi = R.cv_invert_affine(M)
    ys, xs = np.mgrid[0:256, 0:256]
    sx = Mi[0, 0] * xs + Mi[0, 1] * ys + Mi[0, 2]
    sy = Mi[1, 0] * xs + Mi[1, 1] * ys + Mi[1, 2]
    padded = np.pad(smooth, ((2, 2), (2, 2), (0, 0)))  # BORDER_CONSTANT 0: taps outside the frame read 0
    ref = np.stack([map_coordinates(padded[:, :, k], [sy + 2, sx + 2], order=1, mode="constant") for k in range(3)], -1)
    inside = (sx > 0) & (sx < 639) & (sy > 0) & (sy < 479)
    out_f = R.cv_warp_affine(smooth, M, (256, 256), R.INTER_LINEAR)
    assert np.abs(out_f - ref)[inside].max() <= grad * (2.0 / 32) + 1e-3
    u8 = np.clip(np.rint(smooth), 0, 255).astype(np.uint8)
    out_u = R.cv_warp_affine(u8, M, (256, 256), R.INTER_LINEAR)
    assert np.abs(out_u.astype(np.float64) - ref)[inside].max() <= grad * (2.0 / 32) + 1.01  # + u8 rounding of source and result
    far = (sx < -1) | (sx > 640) | (sy < -1) | (sy > 480)
    assert not out_u[far].any() and not out_f[far].any()
    near = R.cv_warp_affine(smooth, M, (256, 256), R.INTER_NEAREST)
    ix, iy = np.floor(sx + 0.5).astype(int), np.floor(sy + 0.5).astype(int)
    ok = (ix >= 0) & (ix < 640) & (iy >= 0) & (iy < 480)
    pick = np.where(ok[..., None], smooth[np.clip(iy, 0, 479), np.clip(ix, 0, 639)], 0)
    assert (near != pick).any(-1).mean() < 0.01  # only positions within 2^-10 of a pixel boundary may differ


def test_targets_are_consistent():
    d = synth.make_roi_frames(6)
    for r in d["rois"]:
        H, W = d["frames"][r["frame"]].shape[:2]
        t = R.roi_targets(r["xyz_crop"], r["xyxy"], r["segmentation"], r["mask_trunc"], (H, W), r["bbox_center"], r["scale"], r["bbox"],
                          d["extents"][r["roi_cls"]], d["fps_points"][r["roi_cls"]], r["trans"], r["centroid_2d"])
        assert t["roi_xyz"].shape == (3, 64, 64) and t["roi_region"].shape == (64, 64) and t["roi_region"].dtype == np.int32
        assert np.array_equal(t["roi_region"] > 0, t["roi_mask_obj"] > 0)
        assert (t["roi_mask_visib"] <= t["roi_mask_obj"]).all() and (t["roi_mask_trunc"] <= t["roi_mask_visib"]).all()
        assert np.array_equal(t["roi_xyz"][:, t["roi_mask_obj"] == 0], np.full((3, int((t["roi_mask_obj"] == 0).sum())), 0.5, np.float32))
        assert t["roi_region"].max() <= 64 and np.isclose(t["resize_ratio"], 64 / r["scale"])


def test_product_module_has_no_cpu_fallback():
    import torch

    from gdrnet_amd import cabi, roi_data
    from gdrnet_amd.cfg import lm13_cfg

    crop = roi_data.RoiCropper.__new__(roi_data.RoiCropper)  # host-side checks only: no device in this container
    crop.device, crop.extents, crop._coord2d = torch.device("cpu"), None, {}
    with pytest.raises(cabi.GdrnHipError):
        crop._tasks([dict(image=torch.zeros(8, 8, 3, dtype=torch.uint8), bbox_center=(4, 4), scale=8.0, bbox=(0, 0, 8, 8))], train=False)
    cfg = lm13_cfg(device="cpu")
    cfg.INPUT = dict(SMOOTH_XYZ=True)
    with pytest.raises(NotImplementedError):
        roi_data.RoiCropper(cfg)
    src = open(roi_data.__file__).read()
    assert "oracle" not in src.replace("no CPU fallback", "")


def test_g8_aug_bbox_matches_the_reference():
    """aug_bbox (core/base_data_loader.py:120-152) draws the same numbers in the same order as the reference under numpy's
    seeded global generator, for the uniform (all GDR-Net configs), roi10d and "none" DZI types."""
    from gdrnet_amd import roi_data
    from gdrnet_amd.cfg import lm13_cfg

    g = np.load(os.path.join(HERE, "golden", "g8_roi_targets.npz"))
    for dzi in ("uniform", "roi10d", "none"):
        cfg = lm13_cfg(device="cpu")
        cfg.INPUT.DZI_TYPE = dzi
        np.random.seed(1234)
        got = []
        for b in g["aug_bbox_boxes"]:
            c, sc = roi_data.aug_bbox(cfg, b, 480, 640)
            got.append([c[0], c[1], sc])
        assert np.array_equal(np.array(got), g["aug_bbox_" + dzi]), dzi
    cfg.INPUT.DZI_TYPE = "truncnorm"
    with pytest.raises(NotImplementedError):
        roi_data.aug_bbox(cfg, g["aug_bbox_boxes"][0], 480, 640)
    c, sc = roi_data.detection_box_to_roi(lm13_cfg(device="cpu"), (100.0, 80.0, 260.0, 200.0), 480, 640)
    assert c.tolist() == [180.0, 140.0] and sc == 240.0
    assert roi_data.detection_box_to_roi(lm13_cfg(device="cpu"), (0.0, 0.0, 639.0, 479.0), 480, 640)[1] == 640.0  # clamped to max(H, W)
