"""Pin the oracle (oracle/gdrn_oracle.py) against outputs of the reference itself.

The fixtures in tests/golden/*.npz were produced by tests/golden/make_golden.py, which imports
the reference (core/gdrn_modeling/models/GDRN.py etc.) on CPU.  Inputs/weights are regenerated
here from the hash RNG, so only reference *outputs* are stored.
"""
import os

import numpy as np
import pytest
import torch

from gdrnet_amd import synth
from oracle import gdrn_oracle as O


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def stats(t):
    t = t.detach().double().flatten()
    idx = torch.linspace(0, t.numel() - 1, 64).long()
    return np.concatenate([[t.mean().item(), t.abs().mean().item(), t.norm().item()], t[idx].numpy()])


@pytest.fixture(scope="module")
def g5(golden_dir):
    return np.load(os.path.join(golden_dir, "g5_e2e.npz"))


@pytest.mark.parametrize("B,tag", [(2, "b2"), (4, "b4")])
def test_forward_loss_backward_vs_reference(g5, B, tag):
    torch.set_num_threads(8)
    sd = synth.make_state_dict(0)
    for v in sd.values():
        if v.is_floating_point():
            v.requires_grad_(True)
    batch = synth.make_batch(B, seed=1)
    bufs = {}
    out = O.gdrn_forward(sd, batch, do_loss=True, training=True, bufs=bufs)
    # fp32 oracle vs fp32 reference: same ATen kernels, same op order -> tight
    assert rel(out["rot6d"].detach(), g5[f"{tag}/rot6d"]) < 2e-5
    assert rel(out["t_"].detach(), g5[f"{tag}/t_"]) < 2e-5
    assert rel(out["rot_allo"].detach(), g5[f"{tag}/rot_allo"]) < 2e-5
    assert rel(out["rot"].detach(), g5[f"{tag}/rot_train"]) < 2e-5
    assert rel(out["trans"].detach(), g5[f"{tag}/trans"]) < 2e-5
    assert rel(stats(out["mask"]), g5[f"{tag}/mask_stats"]) < 2e-5
    assert rel(stats(out["coor_x"]), g5[f"{tag}/coor_x_stats"]) < 2e-5
    assert rel(stats(out["region"]), g5[f"{tag}/region_stats"]) < 2e-5
    names = list(g5[f"{tag}/loss_names"])
    vals = np.array([out["loss_dict"][k].item() for k in names])
    np.testing.assert_allclose(vals, g5[f"{tag}/loss_values"], rtol=2e-5)
    re, te = O.mean_re_te(out["trans"], out["rot"], batch["trans"], batch["ego_rot"])
    assert abs(re - g5[f"{tag}/vis_error_R"]) < 1e-2 and abs(te * 100 - g5[f"{tag}/vis_error_t"]) < 1e-3
    if tag == "b2":
        full = torch.cat([out["mask"], out["coor_x"], out["coor_y"], out["coor_z"], out["region"]], 1)
        assert rel(full.detach(), g5["b2/head_out_full"]) < 2e-5
    sum(out["loss_dict"].values()).backward()
    gn = dict(zip(g5[f"{tag}/grad_names"], g5[f"{tag}/grad_norms"]))
    worst = 0.0
    for k, v in sd.items():
        if k in gn:
            worst = max(worst, abs(v.grad.double().norm().item() - gn[k]) / max(gn[k], 1e-12))
    assert worst < 5e-3, worst  # BN backward at tiny batch is ill-conditioned
    for key in g5.files:
        if key.startswith(f"{tag}/grad/"):
            n = key[len(f"{tag}/grad/"):]
            assert rel(sd[n].grad, g5[key]) < 5e-3, n
    for key in g5.files:
        if key.startswith(f"{tag}/buf/") and not key.endswith("nbt"):
            n = key[len(f"{tag}/buf/"):]
            assert rel(bufs[n], g5[key]) < 1e-5, n
    assert int(bufs["backbone.bn1.num_batches_tracked"]) == int(g5[f"{tag}/buf/nbt"])



G11_CASES = [("lm13_b64_s1", 64, 1, 13), ("lm13_b64_s2", 64, 2, 13), ("lm13_b64_s3", 64, 3, 13), ("lmo_b32_s3", 32, 3, 8)]


@pytest.mark.parametrize("tag,B,seed,ncls", G11_CASES)
def test_oracle_vs_reference_at_baseline_sizes_g11(golden_dir, tag, B, seed, ncls):
    """golden G11 = the reference's own GDRN.forward(do_loss=True) (GDRN.py:83-306), train mode, at BASELINE.json's batch sizes (LM-13 bs = 64,
    seeds 1-3; LM-O bs = 32): the oracle reproduces its pose outputs and its 8 losses to 2e-5 there too -- the bs = 64 / 32 parity tests of the
    HIP path (tests/test_e2e_gpu.py) that compare with the ORACLE are thereby pinned to the reference at the size they run at, not only
    transitively through the B <= 4 fixtures (missing 4 / weak 3)."""
    g = np.load(os.path.join(golden_dir, "g11_baseline_sizes.npz"))
    torch.set_num_threads(8)
    batch = synth.make_batch(B, seed=seed, num_classes=ncls)
    with torch.no_grad():
        out = O.gdrn_forward(synth.make_state_dict(0), batch, do_loss=True, training=True, bufs={})
    for k in ("rot6d", "t_", "rot", "trans"):
        assert rel(out[k], g[f"{tag}/{k}"]) < 2e-5, (tag, k, rel(out[k], g[f"{tag}/{k}"]))
    names = list(g[f"{tag}/loss_names"])
    vals = np.array([out["loss_dict"][k].item() for k in names])
    np.testing.assert_allclose(vals, g[f"{tag}/loss_values"], rtol=2e-5)
    vis = dict(zip(list(g[f"{tag}/vis_names"]), g[f"{tag}/vis_values"]))
    re, te = O.mean_re_te(out["trans"], out["rot"], batch["trans"], batch["ego_rot"])
    assert abs(re - vis["vis/error_R"]) < 1e-2 and abs(te * 100 - vis["vis/error_t"]) < 1e-3

@pytest.mark.parametrize("B,tag", [(2, "b2"), (4, "b4")])
def test_inference_vs_reference(g5, B, tag):
    sd = synth.make_state_dict(0)
    batch = synth.make_batch(B, seed=1)
    with torch.no_grad():
        out = O.gdrn_forward(sd, batch, do_loss=False, training=False)
    assert rel(out["rot"], g5[f"{tag}/eval_rot"]) < 2e-5
    assert rel(out["trans"], g5[f"{tag}/eval_trans"]) < 2e-5


def g10_state_dict(g10):
    """conditioned synthetic weights + the converged BatchNorm buffers stored in G10 (tests/golden/make_golden.py::golden_g10)"""
    sd = synth.conditioned_state_dict(0)
    for k in g10.files:
        if k.startswith("buf/"):
            sd[k[4:]] = torch.from_numpy(g10[k])
    return sd


def test_inference_on_converged_statistics_vs_reference_g10(golden_dir):
    """G10's fp32 leg: the reference's eval-mode forward on the conditioned weights with converged running statistics (the state the
    autocast comparison of tests/test_fp16_gpu.py starts from) -- oracle vs reference; and the stored distance of the reference under
    fp16 autocast (gdrn_evaluator.py:568) to its own fp32 inference is what the golden says it is."""
    g = np.load(os.path.join(golden_dir, "g10_autocast.npz"))
    sd = g10_state_dict(g)
    batch = synth.make_batch(4, seed=77)
    with torch.no_grad():
        out = O.gdrn_forward(sd, batch, do_loss=False, training=False)
    maps = torch.cat([out["mask"], out["coor_x"], out["coor_y"], out["coor_z"], out["region"]], 1)
    assert rel(out["rot"], g["fp32/rot"]) < 2e-5 and rel(out["trans"], g["fp32/trans"]) < 2e-5
    assert rel(maps[:2], g["fp32/maps2"]) < 2e-5
    d = [rel(g["ac_fp16/rot"], g["fp32/rot"]), rel(g["ac_fp16/trans"], g["fp32/trans"])]
    np.testing.assert_allclose(d, g["ac_fp16/dist_to_fp32"][:2], rtol=1e-6)
    assert 1e-3 < d[0] < 5e-2 and str(g["ac_fp16/maps_dtype"]) == "torch.float16"   # the autocast leg did run in half precision


def test_fp64_noise_floor(g5):
    """The reference's own fp32 path sits ~4e-5 (rel. L2) from an fp64 evaluation of the same graph
    (SURVEY.md section 0); the 1e-4 target is judged with that headroom in mind."""
    sd = O.to_dtype(synth.make_state_dict(0), torch.float64)
    batch = O.to_dtype(synth.make_batch(2, seed=1), torch.float64)
    with torch.no_grad():
        out = O.gdrn_forward(sd, batch, do_loss=True, training=True)
    assert rel(out["rot6d"], g5["b2/rot6d"]) < 2e-4
    assert rel(out["trans"], g5["b2/trans"]) < 2e-4


def test_losses_edge_cases(golden_dir):
    g = np.load(os.path.join(golden_dir, "g4_loss.npz"))
    B = 4
    batch = synth.make_batch(B, seed=7, num_classes=21, cam="ycbv", with_sym=True)
    mk = lambda name, shape, s=1.0: torch.from_numpy((synth.hash_normal(11, name, shape) * s).astype(np.float32))
    m, x, y, z = mk("m", (B, 1, 64, 64)), mk("x", (B, 1, 64, 64)), mk("y", (B, 1, 64, 64)), mk("z", (B, 1, 64, 64))
    region = mk("r", (B, 65, 64, 64), 2.0)
    rot6d = mk("r6", (B, 6))
    t_ = mk("t", (B, 3), 0.3) + torch.tensor([0.0, 0.0, 1.0])
    rot, trans = O.pose_decode_train(
        O.ortho6d_to_mat_batch(rot6d), t_, batch["roi_cam"], batch["roi_center"], batch["resize_ratio"], batch["roi_wh"]
    )
    for case in ("normal", "zero_visib"):
        b2 = dict(batch)
        if case == "zero_visib":
            b2["roi_mask_visib"] = torch.zeros_like(batch["roi_mask_visib"])
        for tag, sym in (("nosym", False), ("sym", True)):
            L = O.gdrn_loss(m, x, y, z, region, rot, t_, b2, sym=sym)
            names = list(g[f"{case}/{tag}/names"])
            vals = np.array([L[k].item() for k in names])
            np.testing.assert_allclose(vals, g[f"{case}/{tag}/values"], rtol=1e-5, atol=1e-7)


def test_pose_decode(golden_dir):
    g = np.load(os.path.join(golden_dir, "g3_pose.npz"))
    N = g["rot6d"].shape[0]
    pb = synth.make_batch(N, seed=21)
    center = torch.from_numpy(g["center"])
    r6, t_ = torch.from_numpy(g["rot6d"]), torch.from_numpy(g["t_"])
    R = O.ortho6d_to_mat_batch(r6)
    assert rel(R, g["R_allo"]) < 1e-6
    rtr, ttr = O.pose_decode_train(R, t_, pb["roi_cam"], center, pb["resize_ratio"], pb["roi_wh"])
    np.testing.assert_allclose(ttr.numpy(), g["trans_train"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(rtr.numpy(), g["rot_train"], rtol=0, atol=2e-6)
    rte, tte = O.pose_decode_test(R, t_, pb["roi_cam"], center, pb["resize_ratio"], pb["roi_wh"])
    np.testing.assert_allclose(rte.numpy(), g["rot_test"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(tte.numpy(), g["trans_test"], rtol=1e-6, atol=1e-7)


def test_postproc_oracle_vs_reference_golden(golden_dir):
    """G7: get_out_coor / get_out_mask / get_img_model_points_with_coords2d restatement == the reference's own functions,
    bit for bit (fp32 numpy in the reference's operation order), incl. the flat-mask RoI that yields no points."""
    from gdrnet_amd import synth

    g = np.load(os.path.join(golden_dir, "g7_postproc.npz"))
    inp = synth.make_postproc_inputs(3, 64)
    t = lambda k: torch.from_numpy(inp[k])
    out_mask, out_xyz, pts = O.correspondences_batch(t("mask"), t("coor_x"), t("coor_y"), t("coor_z"), t("coord2d"), t("extents"),
                                                     inp["im_hw"], mask_thr=0.5)
    np.testing.assert_array_equal(out_xyz, g["out_xyz"])
    np.testing.assert_array_equal(out_mask, g["out_mask"])  # NaN == NaN for the flat mask under assert_array_equal
    for i, (ip, mp) in enumerate(pts):
        np.testing.assert_array_equal(ip, g[f"img_pts{i}"])
        np.testing.assert_array_equal(mp, g[f"model_pts{i}"])
    assert len(pts[2][0]) == 0 and len(pts[0][0]) > 1000


def test_ranger_oracle(golden_dir):
    """G6: 7 steps of the reference's own Ranger on a 3-tensor toy == oracle/ranger_oracle.py (the CPU baseline's optimizer step)."""
    from oracle import ranger_oracle as R

    g = np.load(os.path.join(golden_dir, "g6_ranger.npz"))
    ps = [torch.from_numpy(synth.hash_normal(31, f"p{i}", s).astype(np.float32)) for i, s in enumerate(((8, 4, 3, 3), (16, 8), (16,)))]
    state = [dict() for _ in ps]
    for step in range(7):
        grads = [torch.from_numpy(synth.hash_normal(32 + step, f"g{i}", tuple(p.shape)).astype(np.float32)) for i, p in enumerate(ps)]
        R.ranger_step(ps, grads, state, lr=1e-2, weight_decay=0)
        for i, p in enumerate(ps):
            np.testing.assert_allclose(p.numpy(), g[f"step{step}/p{i}"], rtol=1e-6, atol=1e-8)
