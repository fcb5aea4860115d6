"""ORACLE -- test infrastructure, NOT product code.

CPU restatement (plain ``torch.nn.functional`` ops, fp32 or fp64) of GDR-Net's per-RoI hot
path.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; the product package (``gdr-net_amd/``) never does.

Parity pinning: the reference repository has no result-pinning tests for this path
(SURVEY.md section 4).  This oracle is pinned instead against outputs of the *reference itself*,
imported in the build container with import shims by ``tests/golden/make_golden.py``
(fixtures ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` replays them).

Each function cites the reference file:line it restates (paths relative to the reference
checkout).  Third-party arithmetic that is not in the reference tree (torch ATen conv / BN / GN /
bilinear / softmax, torchvision BasicBlock) is restated from its published definition;
``pvnet_net/resnet.py:44-74`` is the in-tree textual twin of BasicBlock.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

RESNET34_LAYERS = (3, 4, 6, 3)


# ----------------------------------------------------------------------------------------------
# building blocks
# ----------------------------------------------------------------------------------------------
def batch_norm(x, sd, prefix, training, bufs=None, momentum=0.1, eps=1e-5):
    """nn.BatchNorm2d (resnet_backbone.py:24, layer_utils.py:30) through ``F.batch_norm``:
    train mode normalises with the biased batch variance,
    ``y = (x - mean_c) * rsqrt(var_c + eps) * weight_c + bias_c``, and updates
    ``running = (1 - momentum) * running + momentum * stat`` with the *unbiased* variance;
    eval mode uses the running statistics."""
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    rm = sd[prefix + ".running_mean"].detach().clone().to(x.dtype)
    rv = sd[prefix + ".running_var"].detach().clone().to(x.dtype)
    if bufs is not None and (prefix + ".running_mean") in bufs:
        rm, rv = bufs[prefix + ".running_mean"].clone(), bufs[prefix + ".running_var"].clone()
    y = F.batch_norm(x, rm, rv, w, b, training, momentum, eps)
    if training and bufs is not None:
        bufs[prefix + ".running_mean"], bufs[prefix + ".running_var"] = rm, rv
        bufs[prefix + ".num_batches_tracked"] = bufs.get(
            prefix + ".num_batches_tracked", sd[prefix + ".num_batches_tracked"]
        ) + 1
    return y


def basic_block(x, sd, p, stride, training, bufs):
    """torchvision BasicBlock (resnet_backbone.py:3; twin pvnet_net/resnet.py:44-74)."""
    out = F.conv2d(x, sd[p + ".conv1.weight"], None, stride, 1)
    out = F.relu(batch_norm(out, sd, p + ".bn1", training, bufs))
    out = F.conv2d(out, sd[p + ".conv2.weight"], None, 1, 1)
    out = batch_norm(out, sd, p + ".bn2", training, bufs)
    if (p + ".downsample.0.weight") in sd:
        idn = F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride, 0)
        idn = batch_norm(idn, sd, p + ".downsample.1", training, bufs)
    else:
        idn = x
    return F.relu(out + idn)


def backbone_forward(x, sd, training=True, bufs=None, taps=None):
    """ResNetBackboneNet.forward, non-concat (resnet_backbone.py:69-80)."""
    x = F.conv2d(x, sd["backbone.conv1.weight"], None, 2, 3)
    x = F.relu(batch_norm(x, sd, "backbone.bn1", training, bufs))
    x = F.max_pool2d(x, 3, 2, 1)
    if taps is not None:
        taps["stem"] = x
    for li, nb in enumerate(RESNET34_LAYERS, start=1):
        for b in range(nb):
            stride = 2 if (b == 0 and li > 1) else 1
            x = basic_block(x, sd, f"backbone.layer{li}.{b}", stride, training, bufs)
        if taps is not None:
            taps[f"layer{li}"] = x
    return x


def head_forward(x, sd, training=True, bufs=None, taps=None):
    """RotWithRegionHead.forward, non-concat branch (cdpn_rot_head_region.py:80-136,182-193)."""
    p = "rot_head_net.features."
    x = F.conv_transpose2d(x, sd[p + "0.weight"], None, stride=2, padding=1, output_padding=1)
    x = F.relu(batch_norm(x, sd, p + "1", training, bufs))
    for conv_i, bn_i in ((3, 4), (6, 7), (10, 11), (13, 14), (17, 18), (20, 21)):
        if conv_i in (10, 17):  # nn.UpsamplingBilinear2d(scale_factor=2) == align_corners=True
            x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
        x = F.conv2d(x, sd[p + f"{conv_i}.weight"], None, 1, 1)
        x = F.relu(batch_norm(x, sd, p + f"{bn_i}", training, bufs))
        if taps is not None:
            taps[f"head{conv_i}"] = x
    x = F.conv2d(x, sd[p + "23.weight"], sd[p + "23.bias"], 1, 0)
    mask = x[:, :1]
    xyz = x[:, 1:4]
    region = x[:, 4:]
    return mask, xyz[:, 0:1], xyz[:, 1:2], xyz[:, 2:3], region


def pnp_forward(coor_feat, region, extents, sd, taps=None):
    """ConvPnPNet.forward (conv_pnp_net.py:111-157).  NB the reference de-normalises
    ``coor_feat[:, :3]`` *in place* (conv_pnp_net.py:121-122); functional here."""
    bs = coor_feat.shape[0]
    xyz = (coor_feat[:, :3] - 0.5) * extents.view(bs, 3, 1, 1)
    x = torch.cat([xyz, coor_feat[:, 3:], region], dim=1)
    if taps is not None:
        taps["pnp_in"] = x
    p = "pnp_net.features."
    for conv_i, gn_i in ((0, 1), (3, 4), (6, 7)):
        x = F.conv2d(x, sd[p + f"{conv_i}.weight"], None, 2, 1)
        x = F.relu(F.group_norm(x, 32, sd[p + f"{gn_i}.weight"], sd[p + f"{gn_i}.bias"], 1e-5))
        if taps is not None:
            taps[f"pnp{conv_i}"] = x
    x = x.reshape(bs, 128 * 8 * 8)  # NCHW flatten: index c*64 + h*8 + w
    x = F.leaky_relu(F.linear(x, sd["pnp_net.fc1.weight"], sd["pnp_net.fc1.bias"]), 0.1)
    x = F.leaky_relu(F.linear(x, sd["pnp_net.fc2.weight"], sd["pnp_net.fc2.bias"]), 0.1)
    rot = F.linear(x, sd["pnp_net.fc_r.weight"], sd["pnp_net.fc_r.bias"])
    t = F.linear(x, sd["pnp_net.fc_t.weight"], sd["pnp_net.fc_t.bias"])
    return rot, t


# ----------------------------------------------------------------------------------------------
# pose decode
# ----------------------------------------------------------------------------------------------
def ortho6d_to_mat_batch(poses):
    """core/utils/rot_reps.py:34-49 (normalize = F.normalize eps 1e-12, rot_reps.py:9-17)."""
    x_raw, y_raw = poses[:, 0:3], poses[:, 3:6]
    x = F.normalize(x_raw, p=2, dim=1)
    z = F.normalize(torch.cross(x, y_raw, dim=1), p=2, dim=1)
    y = torch.cross(z, x, dim=1)
    return torch.stack((x, y, z), dim=2)


def quat2mat(q):
    """core/utils/pose_utils.py:323-370 (eps=0)."""
    q = q / q.norm(p=2, dim=1, keepdim=True)
    qw, qx, qy, qz = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    X, Y, Z = qx * 2.0, qy * 2.0, qz * 2.0
    wX, wY, wZ = qw * X, qw * Y, qw * Z
    xX, xY, xZ = qx * X, qx * Y, qx * Z
    yY, yZ, zZ = qy * Y, qy * Z, qz * Z
    return torch.stack(
        [1.0 - (yY + zZ), xY - wZ, xZ + wY, xY + wZ, 1.0 - (xX + zZ), yZ - wX, xZ - wY, yZ + wX, 1.0 - (xX + yY)],
        dim=1,
    ).reshape(-1, 3, 3)


def allo_to_ego_mat(translation, rot_allo, eps=1e-4):
    """core/utils/utils.py:208-236."""
    obj_ray = translation / (torch.norm(translation, dim=1, keepdim=True) + eps)
    angle = obj_ray[:, 2:3].acos()
    cam_ray = torch.zeros_like(obj_ray)
    cam_ray[:, 2] = 1.0
    axis = torch.cross(cam_ray, obj_ray, dim=1)
    axis = axis / (torch.norm(axis, dim=1, keepdim=True) + eps)
    s = torch.sin(angle / 2.0)
    q = torch.cat([torch.cos(angle / 2.0), axis[:, 0:1] * s, axis[:, 1:2] * s, axis[:, 2:3] * s], dim=1)
    return torch.matmul(quat2mat(q), rot_allo)


def centroid_z_to_trans(pred_centroids, pred_z_vals, roi_cams, roi_centers, resize_ratios, roi_whs):
    """pose_from_pred_centroid_z.py:176-212 (Z_TYPE == REL)."""
    cx = pred_centroids[:, 0:1] * roi_whs[:, 0:1] + roi_centers[:, 0:1]
    cy = pred_centroids[:, 1:2] * roi_whs[:, 1:2] + roi_centers[:, 1:2]
    z = pred_z_vals * resize_ratios.view(-1, 1)
    return torch.cat(
        [z * (cx - roi_cams[:, 0:1, 2]) / roi_cams[:, 0:1, 0], z * (cy - roi_cams[:, 1:2, 2]) / roi_cams[:, 1:2, 1], z],
        dim=1,
    )


def pose_decode_train(pred_rot_m, pred_t_, roi_cams, roi_centers, resize_ratios, roi_whs, eps=1e-4):
    """pose_from_predictions_train (pose_from_pred_centroid_z.py:144-227), allo rot6d."""
    trans = centroid_z_to_trans(pred_t_[:, :2], pred_t_[:, 2:3], roi_cams, roi_centers, resize_ratios, roi_whs)
    return allo_to_ego_mat(trans, pred_rot_m, eps=eps), trans


def _axangle2mat(axis, angle):
    """transforms3d.axangles.axangle2mat (Rodrigues; unpinned third-party, requirements.txt:24)."""
    x, y, z = axis / np.linalg.norm(axis)
    c, s = math.cos(angle), math.sin(angle)
    C = 1 - c
    return np.array(
        [
            [x * x * C + c, x * y * C - z * s, x * z * C + y * s],
            [y * x * C + z * s, y * y * C + c, y * z * C - x * s],
            [z * x * C - y * s, z * y * C + x * s, z * z * C + c],
        ]
    )


def pose_decode_test(pred_rot_m, pred_t_, roi_cams, roi_centers, resize_ratios, roi_whs):
    """pose_from_predictions_test (pose_from_pred_centroid_z.py:52-141) + numpy
    allocentric_to_egocentric (core/utils/utils.py:39-94): per-RoI loop, no eps, math.acos."""
    trans = centroid_z_to_trans(pred_t_[:, :2], pred_t_[:, 2:3], roi_cams, roi_centers, resize_ratios, roi_whs)
    R = pred_rot_m.detach().cpu().numpy()
    T = trans.detach().cpu().numpy()
    out = np.zeros_like(R)
    for i in range(R.shape[0]):
        t = T[i]
        obj_ray = t / np.linalg.norm(t)
        angle = math.acos(max(-1.0, min(1.0, float(obj_ray[2]))))
        if angle > 0:
            rot = _axangle2mat(np.cross(np.array([0.0, 0.0, 1.0]), obj_ray), angle)
            out[i] = rot.dot(R[i])
        else:
            out[i] = R[i]
    return torch.from_numpy(out), trans


# ----------------------------------------------------------------------------------------------
# losses
# ----------------------------------------------------------------------------------------------
def re_deg(R_est, R_gt):
    """lib/pysixd/pose_error.py:400-413."""
    tr = np.trace(R_est.dot(R_gt.T))
    tr = tr if tr <= 3 else 3
    return np.rad2deg(np.arccos(min(1.0, max(-1.0, 0.5 * (tr - 1.0)))))


def get_closest_rot_batch(pred_rots, gt_rots, sym_infos):
    """core/utils/pose_utils.py:430-482 (host loop, argmin rotation error over symmetries)."""
    out = gt_rots.clone()
    P = pred_rots.detach().cpu().numpy().astype(np.float64)
    G = gt_rots.detach().cpu().numpy().astype(np.float64)
    for i, sym in enumerate(sym_infos):
        if sym is None:
            continue
        S = sym.detach().cpu().numpy() if isinstance(sym, torch.Tensor) else np.asarray(sym)
        S = S.reshape(-1, 3, 3).astype(np.float64)
        best, best_err = G[i], re_deg(P[i], G[i])
        for k in range(S.shape[0]):
            cand = G[i].dot(S[k])
            e = re_deg(P[i], cand)
            if e < best_err:
                best, best_err = cand, e
        out[i] = torch.from_numpy(best).to(gt_rots.dtype)
    return out


def gdrn_loss(mask, coor_x, coor_y, coor_z, region, pred_rot, pred_t_, batch, sym=False):
    """GDRN.gdrn_loss, live branches of the benchmark configs (GDRN.py:345-471) and
    PyPMLoss r_only / norm_by_extent (pm_loss.py:82-114, lib/pysixd/misc.py:930-949)."""
    m_visib = batch["roi_mask_visib"]
    gt_xyz = batch["roi_xyz"]
    den = m_visib.sum().clamp(min=1.0)
    L = {}
    mv = m_visib[:, None]
    L["loss_coor_x"] = (coor_x * mv - gt_xyz[:, 0:1] * mv).abs().sum() / den
    L["loss_coor_y"] = (coor_y * mv - gt_xyz[:, 1:2] * mv).abs().sum() / den
    L["loss_coor_z"] = (coor_z * mv - gt_xyz[:, 2:3] * mv).abs().sum() / den
    L["loss_mask"] = (mask[:, 0] - batch["roi_mask_trunc"]).abs().mean()
    gt_region = batch["roi_region"].long()
    L["loss_region"] = F.cross_entropy(region * mv, gt_region * m_visib.long(), reduction="sum") / den
    gt_rot = batch["ego_rot"]
    if sym:
        gt_rot = get_closest_rot_batch(pred_rot, gt_rot, batch["sym_info"])
    pts = batch["roi_points"]
    est = torch.matmul(pred_rot[:, None], pts[..., None]).squeeze(-1)
    tgt = torch.matmul(gt_rot[:, None], pts[..., None]).squeeze(-1)
    w = (1.0 / batch["roi_extent"].max(1, keepdim=True)[0]).view(-1, 1, 1)
    L["loss_PM_R"] = 3 * (w * est - w * tgt).abs().mean()
    L["loss_centroid"] = (pred_t_[:, :2] - batch["roi_trans_ratio"][:, :2]).abs().mean()
    L["loss_z"] = (pred_t_[:, 2] - batch["roi_trans_ratio"][:, 2]).abs().mean()
    return L


def mean_re_te(pred_trans, pred_rot, gt_trans, gt_rot):
    """models/model_utils.py:45-57."""
    P, G = pred_rot.detach().cpu().numpy(), gt_rot.detach().cpu().numpy()
    tp, tg = pred_trans.detach().cpu().numpy(), gt_trans.detach().cpu().numpy()
    re = np.mean([re_deg(P[i], G[i]) for i in range(P.shape[0])], dtype=np.float32)
    te = np.mean(np.linalg.norm(tp - tg, axis=1).astype(np.float32), dtype=np.float32)
    return float(re), float(te)


# ----------------------------------------------------------------------------------------------
# whole path
# ----------------------------------------------------------------------------------------------
def gdrn_forward(sd, batch, do_loss, training=True, bufs=None, sym=False, taps=None):
    """GDRN.forward at the benchmark configs (GDRN.py:110-306): backbone -> head -> cat/softmax ->
    Patch-PnP -> rot6d -> pose decode (-> losses)."""
    feat = backbone_forward(batch["roi_img"], sd, training, bufs, taps)
    mask, cx, cy, cz, region = head_forward(feat, sd, training, bufs, taps)
    coor_feat = torch.cat([cx, cy, cz, batch["roi_coord_2d"]], dim=1)
    region_softmax = F.softmax(region[:, 1:], dim=1)
    rot6d, t_ = pnp_forward(coor_feat, region_softmax, batch["roi_extent"], sd, taps)
    rot_m = ortho6d_to_mat_batch(rot6d)
    args = (batch["roi_cam"], batch["roi_center"], batch["resize_ratio"], batch["roi_wh"])
    out = dict(mask=mask, coor_x=cx, coor_y=cy, coor_z=cz, region=region, rot6d=rot6d, t_=t_, rot_allo=rot_m)
    if do_loss:
        rot, trans = pose_decode_train(rot_m, t_, *args)
        out.update(rot=rot, trans=trans)
        out["loss_dict"] = gdrn_loss(mask, cx, cy, cz, region, rot, t_, batch, sym=sym)
    else:
        rot, trans = pose_decode_test(rot_m, t_, *args)
        out.update(rot=rot, trans=trans)
    return out


def to_dtype(obj, dtype):
    """Cast every floating tensor of a (nested) dict/list to ``dtype``."""
    if isinstance(obj, torch.Tensor):
        return obj.to(dtype) if obj.is_floating_point() else obj
    if isinstance(obj, dict):
        return type(obj)((k, to_dtype(v, dtype)) for k, v in obj.items())
    if isinstance(obj, (list, tuple)):
        return type(obj)(to_dtype(v, dtype) for v in obj)
    return obj


# ------------------------------------------------------------------------------------------------
# Inference post-processing -- CPU restatement, numpy like the reference
def get_out_coor(coor_x, coor_y, coor_z):
    """core/gdrn_modeling/engine_utils.py:92-105, XYZ_LOSS_TYPE == "L1" branch (one channel per coordinate)."""
    assert coor_x.shape[1] == coor_y.shape[1] == coor_z.shape[1] == 1
    return torch.cat([coor_x, coor_y, coor_z], dim=1)


def get_out_mask(pred_mask):
    """engine_utils.py:108-120, MASK_LOSS_TYPE == "L1": per-RoI min-max normalisation, no epsilon."""
    bs, c, h, w = pred_mask.shape
    assert c == 1
    mask_max = torch.max(pred_mask.view(bs, -1), dim=-1)[0].view(bs, 1, 1, 1)
    mask_min = torch.min(pred_mask.view(bs, -1), dim=-1)[0].view(bs, 1, 1, 1)
    return (pred_mask - mask_min) / (mask_max - mask_min)


def get_img_model_points_with_coords2d(mask_pred_crop, xyz_pred_crop, coord2d_crop, im_H, im_W, extent, mask_thr=0.5):
    """core/gdrn_modeling/gdrn_evaluator.py:89-126 with max_num_points < 4 (no random sub-sampling).
    mask HW, xyz HWC in [0,1], coord2d HW2 in [0,1] (numpy fp32) -> (image_points [n,2], model_points [n,3])."""
    xyz = np.array(xyz_pred_crop, dtype=np.float32, copy=True)
    c2d = np.array(coord2d_crop, dtype=np.float32, copy=True)
    extent = np.asarray(extent, dtype=np.float32)
    for c in range(3):
        xyz[:, :, c] = (xyz[:, :, c] - np.float32(0.5)) * extent[c]
    c2d[:, :, 0] = c2d[:, :, 0] * np.float32(im_W)
    c2d[:, :, 1] = c2d[:, :, 1] * np.float32(im_H)
    sel = (
        (mask_pred_crop > np.float32(mask_thr))
        & (np.abs(xyz[:, :, 0]) > np.float32(0.0001) * extent[0])
        & (np.abs(xyz[:, :, 1]) > np.float32(0.0001) * extent[1])
        & (np.abs(xyz[:, :, 2]) > np.float32(0.0001) * extent[2])
    )
    return c2d[sel].reshape(-1, 2), xyz[sel].reshape(-1, 3)


def correspondences_batch(mask, coor_x, coor_y, coor_z, roi_coord_2d, roi_extents, im_hw, mask_thr=0.5):
    """the evaluator's per-instance loop (gdrn_evaluator.py:325-377) over a batch: returns out_mask [N,1,H,W],
    out_xyz [N,3,H,W], and per-RoI (image_points, model_points) lists."""
    out_xyz = get_out_coor(coor_x, coor_y, coor_z).numpy()
    out_mask = get_out_mask(mask).numpy()
    pts = []
    for i in range(out_xyz.shape[0]):
        pts.append(get_img_model_points_with_coords2d(
            np.squeeze(out_mask[i]), out_xyz[i].transpose(1, 2, 0), roi_coord_2d[i].numpy().transpose(1, 2, 0),
            im_H=float(im_hw[i][0]), im_W=float(im_hw[i][1]), extent=roi_extents[i].numpy(), mask_thr=mask_thr))
    return out_mask, out_xyz, pts
