"""ORACLE -- test infrastructure, NOT product code.

CPU restatement (numpy, integer / fp32 / fp64 exactly as the reference's dependencies compute) of
the step immediately *before* GDR-Net's per-RoI hot path: the RoI cropper / target builder of the
reference's data loader.  Only ``tests/`` may import this module.

Reference sites restated (paths relative to the reference checkout):

* ``core/utils/data_utils.py:80-137``  ``crop_resize_by_warp_affine`` / ``get_affine_transform``
* ``core/utils/data_utils.py:146-158`` ``get_3rd_point`` / ``get_dir``
* ``core/utils/data_utils.py:213-219`` ``xyz_to_region``
* ``core/base_data_loader.py:114-118`` ``normalize_image``
* ``core/gdrn_modeling/data_loader.py:411-444`` (test-mode RoI inputs) and ``:460-545,617-632``
  (train-mode inputs, masks, xyz, region labels, ``trans_ratio``)

Parity pinning
--------------
``xyz_to_region`` and ``get_2d_coord_np`` ARE pinned: fixture ``tests/golden/g8_roi_targets.npz``
holds outputs of the reference's own functions imported in the build container
(``tests/golden/make_golden.py``).

``cv2.warpAffine`` / ``cv2.getAffineTransform`` are **parity unpinned**: OpenCV (``opencv-python``,
unpinned in the reference's ``requirements.txt:20``) is a third-party dependency that is absent
from this image and the reference holds no test vectors for it.  They are restated here from
OpenCV 4.x's published algorithm (``modules/imgproc/src/imgwarp.cpp``):

* ``getAffineTransform``: the 6x6 system ``[x y 1 0 0 0; 0 0 0 x y 1] m = [u; v]`` solved in double
  by Gaussian elimination with partial pivoting (``cv::solve(..., DECOMP_LU)`` -> ``hal::LU64f``).
* ``warpAffine`` (no ``WARP_INVERSE_MAP``): inverts the 2x3 matrix in double, then walks the
  destination grid in fixed point -- ``AB_BITS = 10`` coordinate bits, source positions rounded to
  1/32 pixel (``INTER_BITS = 5``), ``round_delta = 512`` (nearest) or ``16`` (linear) -- and hands
  integer positions + a 5+5-bit fraction index to ``remap``:
  nearest copies the pixel (``BORDER_CONSTANT`` 0 outside), bilinear uses the 32x32 table of
  products ``(1-fy/32,(fy/32)) x (1-fx/32, fx/32)``; for ``uint8`` images the table is in 15-bit
  fixed point and the result is ``(sum + 2^14) >> 15``; for ``float32`` images the four products
  are accumulated left to right in fp32.
"""
import numpy as np

AB_BITS = 10
AB_SCALE = 1 << AB_BITS
INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS
INTER_REMAP_COEF_BITS = 15
INTER_NEAREST = 0
INTER_LINEAR = 1


# ----------------------------------------------------------------------------------------------
# affine transform of a crop
# ----------------------------------------------------------------------------------------------
def get_dir(src_point, rot_rad):
    """data_utils.py:151-158."""
    sn, cs = np.sin(rot_rad), np.cos(rot_rad)
    return [src_point[0] * cs - src_point[1] * sn, src_point[0] * sn + src_point[1] * cs]


def get_3rd_point(a, b):
    """data_utils.py:146-148 (fp32 arithmetic on fp32 rows)."""
    direct = a - b
    return b + np.array([-direct[1], direct[0]], dtype=np.float32)


def cv_get_affine_transform(src, dst):
    """cv2.getAffineTransform: 3 point pairs (fp32) -> 2x3 double matrix, LU with partial pivoting."""
    src = np.asarray(src, np.float32).astype(np.float64)
    dst = np.asarray(dst, np.float32).astype(np.float64)
    A = np.zeros((6, 6), np.float64)
    b = np.zeros(6, np.float64)
    for i in range(3):
        A[i, 0], A[i, 1], A[i, 2] = src[i, 0], src[i, 1], 1.0
        A[i + 3, 3], A[i + 3, 4], A[i + 3, 5] = src[i, 0], src[i, 1], 1.0
        b[i], b[i + 3] = dst[i, 0], dst[i, 1]
    m = 6
    for i in range(m):
        k = i
        for j in range(i + 1, m):
            if abs(A[j, i]) > abs(A[k, i]):
                k = j
        if abs(A[k, i]) < np.finfo(np.float64).eps * 100:
            return np.zeros((2, 3), np.float64)  # singular: cv::solve leaves the result zero
        if k != i:
            A[[i, k], i:] = A[[k, i], i:]
            b[[i, k]] = b[[k, i]]
        d = -1.0 / A[i, i]
        for j in range(i + 1, m):
            alpha = A[j, i] * d
            for kk in range(i + 1, m):
                A[j, kk] = A[j, kk] + alpha * A[i, kk]
            b[j] = b[j] + alpha * b[i]
    for i in range(m - 1, -1, -1):
        s = b[i]
        for kk in range(i + 1, m):
            s = s - A[i, kk] * b[kk]
        b[i] = s / A[i, i]
    return b.reshape(2, 3)


def get_affine_transform(center, scale, rot, output_size, inv=False):
    """data_utils.py:94-137 (``shift`` is always zero at the call sites)."""
    center = np.asarray(center)
    if isinstance(scale, (int, float)):
        scale = np.array([scale, scale], dtype=np.float32)
    else:
        scale = np.asarray(scale, dtype=np.float32)
    if isinstance(output_size, (int, float)):
        output_size = (output_size, output_size)
    src_w = scale[0]
    dst_w, dst_h = output_size[0], output_size[1]
    rot_rad = np.pi * rot / 180
    src_dir = get_dir([0, src_w * -0.5], rot_rad)
    dst_dir = np.array([0, dst_w * -0.5], np.float32)
    src = np.zeros((3, 2), dtype=np.float32)
    dst = np.zeros((3, 2), dtype=np.float32)
    src[0, :] = center
    src[1, :] = center + np.asarray(src_dir, np.float64)
    dst[0, :] = [dst_w * 0.5, dst_h * 0.5]
    dst[1, :] = np.array([dst_w * 0.5, dst_h * 0.5], np.float32) + dst_dir
    src[2, :] = get_3rd_point(src[0, :], src[1, :])
    dst[2, :] = get_3rd_point(dst[0, :], dst[1, :])
    return cv_get_affine_transform(dst, src) if inv else cv_get_affine_transform(src, dst)


def cv_invert_affine(M):
    """The in-place inversion warpAffine applies to the forward matrix (double)."""
    M = np.array(M, np.float64).reshape(-1)
    D = M[0] * M[4] - M[1] * M[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[4] * D, M[0] * D
    m0, m1, m3, m4 = A11, M[1] * (-D), M[3] * (-D), A22
    b1 = -m0 * M[2] - m1 * M[5]
    b2 = -m3 * M[2] - m4 * M[5]
    return np.array([[m0, m1, b1], [m3, m4, b2]], np.float64)


# ----------------------------------------------------------------------------------------------
# warpAffine
# ----------------------------------------------------------------------------------------------
def _cv_round(x):
    return np.rint(x).astype(np.int64)


def _sat_short(v):
    return np.clip(v, -32768, 32767)


def warp_coords(Minv, dsize, interpolation):
    """Fixed-point source coordinates of every destination pixel: (sx, sy, fx, fy) int64 [h][w]."""
    w, h = int(dsize[0]), int(dsize[1])
    xs = np.arange(w, dtype=np.float64)
    ys = np.arange(h, dtype=np.float64)
    adelta = _cv_round(Minv[0, 0] * xs * AB_SCALE)
    bdelta = _cv_round(Minv[1, 0] * xs * AB_SCALE)
    rd = AB_SCALE // 2 if interpolation == INTER_NEAREST else AB_SCALE // INTER_TAB_SIZE // 2
    X0 = _cv_round((Minv[0, 1] * ys + Minv[0, 2]) * AB_SCALE) + rd
    Y0 = _cv_round((Minv[1, 1] * ys + Minv[1, 2]) * AB_SCALE) + rd
    X = X0[:, None] + adelta[None, :]
    Y = Y0[:, None] + bdelta[None, :]
    if interpolation == INTER_NEAREST:
        return _sat_short(X >> AB_BITS), _sat_short(Y >> AB_BITS), None, None
    X >>= AB_BITS - INTER_BITS
    Y >>= AB_BITS - INTER_BITS
    return _sat_short(X >> INTER_BITS), _sat_short(Y >> INTER_BITS), X & (INTER_TAB_SIZE - 1), Y & (INTER_TAB_SIZE - 1)


def _fetch(img, sy, sx):
    """img[sy, sx] with BORDER_CONSTANT 0 outside; img is [H][W][C]."""
    H, W = img.shape[:2]
    ok = (sx >= 0) & (sx < W) & (sy >= 0) & (sy < H)
    v = img[np.clip(sy, 0, H - 1), np.clip(sx, 0, W - 1)]
    return np.where(ok[..., None], v, np.zeros((), img.dtype))


def cv_warp_affine(img, M, dsize, flags=INTER_LINEAR):
    """cv2.warpAffine(img, M, dsize, flags=flags) for uint8 / float32 images, BORDER_CONSTANT 0.
    As in OpenCV, a single-channel [H][W][1] input gives a 2-D [h][w] result."""
    img = np.asarray(img)
    assert img.dtype in (np.uint8, np.float32)
    src = img[:, :, None] if img.ndim == 2 else img
    sx, sy, fx, fy = warp_coords(cv_invert_affine(M), dsize, flags)
    if flags == INTER_NEAREST:
        out = _fetch(src, sy, sx)
    else:
        s00, s01 = _fetch(src, sy, sx), _fetch(src, sy, sx + 1)
        s10, s11 = _fetch(src, sy + 1, sx), _fetch(src, sy + 1, sx + 1)
        if img.dtype == np.uint8:
            ax, ay = fx, fy  # weights a*b*32 with a+b = 32 per axis: exact 15-bit integers
            w00, w01 = (32 - ay) * (32 - ax) * 32, (32 - ay) * ax * 32
            w10, w11 = ay * (32 - ax) * 32, ay * ax * 32
            zero = (ax == 0) & (ay == 0)  # 32768 saturates to short 32767; OpenCV moves the missing 1 to tap 11
            w00 = np.where(zero, 32767, w00)
            w11 = np.where(zero, 1, w11)
            acc = (s00.astype(np.int64) * w00[..., None] + s01.astype(np.int64) * w01[..., None]
                   + s10.astype(np.int64) * w10[..., None] + s11.astype(np.int64) * w11[..., None])
            out = np.clip((acc + (1 << (INTER_REMAP_COEF_BITS - 1))) >> INTER_REMAP_COEF_BITS, 0, 255).astype(np.uint8)
        else:
            one, sc = np.float32(1.0), np.float32(1.0 / INTER_TAB_SIZE)
            tx, ty = fx.astype(np.float32) * sc, fy.astype(np.float32) * sc
            w00, w01 = ((one - ty) * (one - tx))[..., None], ((one - ty) * tx)[..., None]
            w10, w11 = (ty * (one - tx))[..., None], (ty * tx)[..., None]
            out = ((s00 * w00 + s01 * w01) + s10 * w10) + s11 * w11
            out = out.astype(np.float32)
    return out[:, :, 0] if out.shape[2] == 1 else out


def crop_resize_by_warp_affine(img, center, scale, output_size, rot=0, interpolation=INTER_LINEAR):
    """data_utils.py:80-92."""
    if isinstance(scale, (int, float)):
        scale = (scale, scale)
    if isinstance(output_size, int):
        output_size = (output_size, output_size)
    trans = get_affine_transform(center, scale, rot, output_size)
    return cv_warp_affine(img, trans, (int(output_size[0]), int(output_size[1])), flags=interpolation)


# ----------------------------------------------------------------------------------------------
# targets
# ----------------------------------------------------------------------------------------------
def get_2d_coord_np(width, height, low=0, high=1, fmt="CHW"):
    """data_utils.py:222-241."""
    x = np.linspace(low, high, width, dtype=np.float32)
    y = np.linspace(low, high, height, dtype=np.float32)
    xy = np.asarray(np.meshgrid(x, y))
    return xy.transpose(1, 2, 0) if fmt == "HWC" else xy


def xyz_to_region(xyz_crop, fps_points):
    """data_utils.py:213-219: label = 1 + argmin_k ||xyz - fps_k|| (double, scipy ``cdist``), 0 on background."""
    bh, bw = xyz_crop.shape[:2]
    mask = ((xyz_crop[:, :, 0] != 0) | (xyz_crop[:, :, 1] != 0) | (xyz_crop[:, :, 2] != 0)).astype("uint8")
    p = xyz_crop.reshape(bh * bw, 1, 3).astype(np.float64)
    f = np.asarray(fps_points, np.float64)[None]
    d = p - f
    dist = np.sqrt(d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1] + d[..., 2] * d[..., 2])
    return mask * (np.argmin(dist, axis=1).reshape(bh, bw) + 1)


def normalize_image(image_chw, pixel_mean, pixel_std):
    """base_data_loader.py:114-118 (float64 arithmetic; the caller casts to float32)."""
    mean = np.array(pixel_mean).reshape(-1, 1, 1)
    std = np.array(pixel_std).reshape(-1, 1, 1)
    return (image_chw - mean) / std


def roi_inputs(image, coord_2d, bbox_center, scale, input_res=256, out_res=64, pixel_mean=(0, 0, 0), pixel_std=(255.0, 255.0, 255.0)):
    """data_loader.py:425-439 / :487-498: ``roi_img`` [3][in][in] and ``roi_coord_2d`` [2][out][out], fp32."""
    roi_img = crop_resize_by_warp_affine(image, bbox_center, scale, input_res, interpolation=INTER_LINEAR).transpose(2, 0, 1)
    roi_img = normalize_image(roi_img, pixel_mean, pixel_std).astype("float32")
    roi_coord_2d = crop_resize_by_warp_affine(coord_2d, bbox_center, scale, out_res, interpolation=INTER_LINEAR).transpose(2, 0, 1)
    return roi_img, roi_coord_2d.astype("float32")


def roi_targets(xyz_crop, xyxy, seg, mask_trunc, im_hw, bbox_center, scale, bbox_xyxy, roi_extent, fps_points, trans, centroid_2d, out_res=64):
    """data_loader.py:460-545,617-632 with the base config (nearest masks / xyz, L1 xyz loss, 64 regions, no
    SMOOTH_XYZ): returns the dict of train-mode targets."""
    im_H, im_W = im_hw
    x1, y1, x2, y2 = xyxy
    xyz = np.zeros((im_H, im_W, 3), dtype=np.float32)
    xyz[y1 : y2 + 1, x1 : x2 + 1, :] = xyz_crop
    mask_obj = ((xyz[:, :, 0] != 0) | (xyz[:, :, 1] != 0) | (xyz[:, :, 2] != 0)).astype(bool).astype(np.float32)
    mask_visib = seg.astype("float32") * mask_obj
    mask_trunc = mask_visib if mask_trunc is None else mask_visib * mask_trunc.astype("float32")
    bw = max(bbox_xyxy[2] - bbox_xyxy[0], 1)
    bh = max(bbox_xyxy[3] - bbox_xyxy[1], 1)
    crop = lambda a: crop_resize_by_warp_affine(a, bbox_center, scale, out_res, interpolation=INTER_NEAREST)  # noqa: E731
    roi_mask_trunc, roi_mask_visib, roi_mask_obj = crop(mask_trunc[:, :, None]), crop(mask_visib[:, :, None]), crop(mask_obj[:, :, None])
    roi_xyz = crop(xyz)
    roi_region = xyz_to_region(roi_xyz, fps_points).astype(np.int32)
    roi_xyz = roi_xyz.transpose(2, 0, 1).copy()
    roi_extent = np.asarray(roi_extent, np.float32)
    for c in range(3):
        roi_xyz[c] = roi_xyz[c] / roi_extent[c] + 0.5
    resize_ratio = out_res / scale
    # the reference needs NumPy < 1.24 (np.bool, data_loader.py:471): float32 scalar / Python float is evaluated in double there
    z_ratio = float(trans[2]) / resize_ratio
    delta_c = np.asarray(centroid_2d, np.float64) - np.asarray(bbox_center, np.float64)
    return dict(
        roi_xyz=roi_xyz.astype("float32"), roi_mask_trunc=roi_mask_trunc.astype("float32"), roi_mask_visib=roi_mask_visib.astype("float32"),
        roi_mask_obj=roi_mask_obj.astype("float32"), roi_region=roi_region,
        roi_wh=np.array([bw, bh], dtype=np.float32), resize_ratio=np.float32(resize_ratio),
        trans_ratio=np.array([delta_c[0] / bw, delta_c[1] / bh, z_ratio]).astype(np.float32),
    )
