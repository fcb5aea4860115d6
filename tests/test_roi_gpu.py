"""GPU parity of the RoI cropper / target builder through the C ABI: bit-exact against the
oracle (integer / index / u8 work, and fp32 work evaluated in the reference's operation order)."""
import numpy as np
import pytest
import torch

from gdrnet_amd import cabi, roi_data, synth
from gdrnet_amd.cfg import lm13_cfg
from oracle import roi_oracle as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _device_rois(d):
    frames = [torch.from_numpy(f).to(DEV) for f in d["frames"]]
    rois = []
    for r in d["rois"]:
        q = dict(r)
        q["image"] = frames[r["frame"]]
        q["xyz_crop"] = torch.from_numpy(r["xyz_crop"]).to(DEV)
        q["segmentation"] = torch.from_numpy(r["segmentation"]).to(DEV)
        q["mask_trunc"] = None if r["mask_trunc"] is None else torch.from_numpy(r["mask_trunc"]).to(DEV)
        rois.append(q)
    return rois


def _oracle(d, r, train):
    frame = d["frames"][r["frame"]]
    H, W = frame.shape[:2]
    img, c2d = R.roi_inputs(frame, R.get_2d_coord_np(W, H, fmt="HWC"), r["bbox_center"], r["scale"])
    o = dict(roi_img=img, roi_coord_2d=c2d)
    if train:
        o.update(R.roi_targets(r["xyz_crop"], r["xyxy"], r["segmentation"], r["mask_trunc"], (H, W), r["bbox_center"], r["scale"], r["bbox"],
                               d["extents"][r["roi_cls"]], d["fps_points"][r["roi_cls"]], r["trans"], r["centroid_2d"]))
    return o


def _cropper(d):
    return roi_data.RoiCropper(lm13_cfg(device=DEV), extents=d["extents"], fps_points=d["fps_points"], device=DEV)


@pytest.mark.parametrize("train", [False, True])
def test_roi_cropper_matches_oracle_bit_exactly(train):
    d = synth.make_roi_frames(9)
    out = _cropper(d)(_device_rois(d), train=train)
    torch.cuda.synchronize()
    keys = ["roi_img", "roi_coord_2d"] + (["roi_xyz", "roi_mask_trunc", "roi_mask_visib", "roi_mask_obj", "roi_region", "roi_wh", "trans_ratio"] if train else [])
    for n, r in enumerate(d["rois"]):
        ref = _oracle(d, r, train)
        for k in keys:
            got = out[k][n].cpu().numpy()
            assert got.dtype == ref[k].dtype and got.shape == ref[k].shape, (n, k, got.dtype, got.shape)
            assert np.array_equal(got, ref[k]), (n, k, float(np.abs(got.astype(np.float64) - ref[k]).max()))
        if train:
            assert out["resize_ratio"][n].item() == ref["resize_ratio"]
            assert out["roi_cls"][n].item() == r["roi_cls"] and np.array_equal(out["roi_extent"][n].cpu().numpy(), d["extents"][r["roi_cls"]])
    assert out["roi_img"].shape == (9, 3, 256, 256) and out["roi_coord_2d"].shape == (9, 2, 64, 64)
    if not train:
        assert "roi_xyz" not in out


def test_roi_cropper_full_batch_properties():
    """bs = 64 (BASELINE config 2): masks / labels are consistent over the whole batch and a sample of its RoIs equals
    the oracle bit for bit."""
    d = synth.make_roi_frames(64, seed=9)
    out = _cropper(d)(_device_rois(d), train=True)
    torch.cuda.synchronize()
    assert np.array_equal((out["roi_region"] > 0).cpu().numpy(), (out["roi_mask_obj"] > 0).cpu().numpy())
    assert bool((out["roi_mask_trunc"] <= out["roi_mask_visib"]).all()) and bool((out["roi_mask_visib"] <= out["roi_mask_obj"]).all())
    assert bool(torch.isfinite(out["roi_img"]).all()) and float(out["roi_img"].max()) <= 1.0 and float(out["roi_img"].min()) >= 0.0
    for n in (0, 1, 2, 3, 17, 40, 63):
        ref = _oracle(d, d["rois"][n], True)
        for k in ("roi_img", "roi_coord_2d", "roi_xyz", "roi_mask_visib", "roi_region", "trans_ratio"):
            assert np.array_equal(out[k][n].cpu().numpy(), ref[k]), (n, k)


def test_identity_crop_and_test_mode_only_inputs():
    img = torch.from_numpy(np.floor(synth.hash_uniform(3, "img", (96, 80, 3)) * 256).astype(np.uint8)).to(DEV)
    crop = roi_data.RoiCropper(lm13_cfg(device=DEV), device=DEV)
    crop.input_res = crop.out_res = 64
    out = crop([dict(image=img, bbox_center=(32.0, 32.0), scale=64.0, bbox=(0, 0, 64, 64))])
    want = img[:64, :64].permute(2, 0, 1).double().div(255.0).float()
    assert torch.equal(out["roi_img"][0], want)
    c2d = crop.coord_2d(96, 80)
    assert torch.equal(out["roi_coord_2d"][0], c2d[:64, :64].permute(2, 0, 1))
    assert out["roi_wh"].tolist() == [[64.0, 64.0]] and out["resize_ratio"].tolist() == [1.0]


def test_roi_c_abi_argument_errors():
    lib = cabi.load()
    t = torch.zeros(256, dtype=torch.uint8, device=DEV)
    m = torch.zeros(12, dtype=torch.float64, device=DEV)
    assert lib.gdrn_roi_affine(None, 1, 256, 64, cabi.ptr(m), None, None, None, None) == -1
    assert lib.gdrn_roi_affine(cabi.ptr(t), 0, 256, 64, cabi.ptr(m), None, None, None, None) == -1
    assert lib.gdrn_roi_crop_inputs(cabi.ptr(t), cabi.ptr(m), 1, 256, 64, None, None, cabi.ptr(t), None, None) == -1
    assert lib.gdrn_roi_targets(cabi.ptr(t), cabi.ptr(m), 1, 64, None, 64, cabi.ptr(t), cabi.ptr(t), cabi.ptr(t), cabi.ptr(t), cabi.ptr(t), cabi.ptr(t), None) == -1
    crop = roi_data.RoiCropper(lm13_cfg(device=DEV), device=DEV)
    with pytest.raises(ValueError):
        crop([])
    with pytest.raises(cabi.GdrnHipError):
        crop([dict(image=torch.zeros(8, 8, 3, dtype=torch.uint8), bbox_center=(4, 4), scale=8.0, bbox=(0, 0, 8, 8))])


def test_cropped_batch_feeds_the_model():
    """the cropper's device tensors are the model's inputs / targets as they are (engine_utils.py:6-60 key mapping)."""
    from gdrnet_amd import GDRN

    B = 4
    d = synth.make_roi_frames(B, seed=11)
    out = _cropper(d)(_device_rois(d), train=True)
    batch = {k: (v.to(DEV) if isinstance(v, torch.Tensor) else v) for k, v in synth.make_batch(B, seed=1).items()}
    for k in ("roi_img", "roi_coord_2d", "roi_xyz", "roi_mask_trunc", "roi_mask_visib", "roi_mask_obj", "roi_region", "roi_wh", "resize_ratio",
              "roi_cls", "roi_extent"):
        assert out[k].shape == batch[k].shape, k
        batch[k] = out[k]
    batch["roi_center"], batch["roi_trans_ratio"] = out["bbox_center"], out["trans_ratio"]
    cfg = lm13_cfg(device=DEV)
    cfg.MODEL.CDPN.HIP_DTYPE = "bf16"
    model, _ = GDRN.build_model_optimizer(cfg)
    model.load_state_dict(synth.make_state_dict(0))
    model.train()
    _, loss_dict = model(batch["roi_img"], **synth.model_kwargs(batch, do_loss=True))
    total = sum(loss_dict.values())
    assert bool(torch.isfinite(total)) and len(loss_dict) == 8
    total.backward()
