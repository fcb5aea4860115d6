"""ctypes binding of the C-ABI HIP library ``libgdrn_hip.so`` (declared in ``include/gdrn_hip.h``).

The product path has no CPU fallback: if the library is missing or a call returns a non-zero
status, ``GdrnHipError`` is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GDRN_HIP_LIB") or os.path.join(_HERE, "lib", "libgdrn_hip.so")   # (GDRN_HIP_LIB: A/B runs of two builds on one box)
# the same sources built with IEEE half as the 16-bit format (csrc/common.h): the reference's fp16 autocast arithmetic
LIB_PATH_F16 = os.environ.get("GDRN_HIP_LIB_F16") or os.path.join(_HERE, "lib", "libgdrn_hip_f16.so")

F32, BF16, F16 = 0, 1, 2
PREZEROED = 0x100  # GDRN_PREZEROED
ACC_ROWS = 0x200   # GDRN_ACC_ROWS (gdrn_head_tail_loss_fwd: per-workgroup partial rows behind acc[0..7] instead of atomics)
P = C.c_void_p
I = C.c_int
LL = C.c_longlong
F = C.c_float
D = C.c_double


class GdrnHipError(RuntimeError):
    pass


class ConvParams(C.Structure):
    _fields_ = [
        ("x", P), ("w", P), ("y", P), ("bias", P), ("addend", P), ("stats", P),
        ("bnb_x", P), ("bnb_mask", P), ("bnb_mean", P), ("bnb_invstd", P), ("bnb_scale", P), ("bnb_shift", P), ("bnb_rows", P),
        ("bnb_cs", I), ("w_frag", I),
        ("Hi", I), ("Wi", I), ("Cin", I), ("x_cs", I),
        ("Ho", I), ("Wo", I), ("Cout", I), ("y_cs", I), ("add_cs", I),
        ("KH", I), ("KW", I), ("stride", I), ("pad", I),
        ("mode", I), ("act", I), ("out_f32", I),
        ("M", I), ("w_rows", I), ("dtype", I),
        ("xf_mode", I), ("xf_relu", I),
        ("xf_x2", P), ("xf_a", P), ("xf_b", P), ("xf_c", P), ("xf_c2", P), ("xf_msc", P), ("xf_msh", P), ("xf_out", P),
        ("halo_waves", I), ("v3_min_wg", I),
    ]


class WgradParams(C.Structure):
    _fields_ = [
        ("x", P), ("dy", P), ("dw", P), ("ws", P),
        ("Hi", I), ("Wi", I), ("Cin", I), ("x_cs", I),
        ("Ho", I), ("Wo", I), ("Cout", I), ("dy_cs", I),
        ("KH", I), ("KW", I), ("stride", I), ("pad", I),
        ("M", I), ("dtype", I), ("splits", I), ("variant", I),
    ]


class S2Params(C.Structure):
    """gdrn_s2_params: the stride-2 3x3 conv (+ fused 1x1 shortcut) of csrc/conv3x3s2.hip"""
    _fields_ = [
        ("x", P), ("w", P), ("y", P), ("bias", P), ("stats", P), ("wd", P), ("yd", P), ("bias_d", P), ("stats_d", P),
        ("Hi", I), ("Wi", I), ("Cin", I), ("x_cs", I),
        ("Ho", I), ("Wo", I), ("Cout", I), ("y_cs", I), ("yd_cs", I),
        ("N", I), ("w_rows", I), ("wd_rows", I), ("act", I), ("dtype", I),
        ("bnb_x", P), ("bnb_mask", P), ("bnb_mean", P), ("bnb_invstd", P), ("bnb_rows", P), ("bnb_cs", I),   # (ABI 5) BatchNorm-backward epilogue
    ]


class S2dParams(C.Structure):
    """gdrn_s2d_params: data gradient of the stride-2 3x3 conv (+ the shortcut's, + BatchNorm-backward epilogue) of csrc/conv3x3s2_dgrad.hip"""
    _fields_ = [
        ("dy", P), ("w", P), ("dx", P), ("dyd", P), ("wdd", P), ("bnb_x", P), ("bnb_mask", P), ("bnb_mean", P), ("bnb_invstd", P), ("bnb_rows", P),
        ("bnb_cs", I),
        ("Hi", I), ("Wi", I), ("Cin", I), ("dx_cs", I),
        ("Ho", I), ("Wo", I), ("Cout", I), ("dy_cs", I), ("dyd_cs", I),
        ("N", I), ("w_rows", I), ("wdd_rows", I), ("dtype", I),
        ("stats", P), ("bias", P), ("act", I),   # (ABI 5) forward-conv epilogue: the launch as the ConvTranspose2d forward
    ]


class PoseParams(C.Structure):
    _fields_ = [
        ("fc", P), ("fs", I), ("cams", P), ("centers", P), ("whs", P), ("ratios", P), ("extents", P),
        ("gt_rot", P), ("gt_trans", P), ("gt_trans_ratio", P), ("points", P), ("npts", I),
        ("sym", P), ("sym_count", P), ("Kmax", I), ("N", I), ("train", I),
        ("rot", P), ("trans", P), ("losses", P), ("dfc", P), ("vis", P),
        # (ABI 5, all nullable) per-RoI loss rows | dL/dloss + combined gradient | the fc tail (fc2 finish + fc_r / fc_t) in the same launch
        ("loss_rows", P), ("gw", P), ("dfc_comb", P), ("fc2_ws", P), ("fc2_bias", P), ("f2_out", P), ("w_rt", P), ("b_rt", P), ("fc_w", P), ("fc2_splits", I),
    ]


class PackTask(C.Structure):
    _fields_ = [
        ("src", P), ("dst", P), ("scale", P),
        ("A1", I), ("A2", I), ("T", I), ("B", I), ("A1v", I), ("A2v", I), ("Bv", I), ("flip", I),
        ("s1", LL), ("s2", LL), ("st", LL), ("sb", LL), ("n", LL), ("frag", I), ("pad_", I),
    ]


def transpose_blocks(lib, t):
    """frag / pad_ of a row-major PackTask set for the tiled-transpose path of gdrn_pack_multi / gdrn_unpack_multi when it qualifies (its
    unit-stride source index is a1, a2 or t rather than b; no flip, no scale); returns the workgroup count, or 0 (task left unchanged)."""
    if t.frag or t.flip or t.scale or t.sb == 1:
        return 0
    for u, (stride, size) in enumerate(((t.s1, t.A1), (t.s2, t.A2), (t.st, t.T)), start=1):
        if stride == 1 and size >= 32:
            t.frag, t.pad_ = 3, u
            nb = int(lib.gdrn_pack_transpose_blocks(C.byref(t)))
            if nb > 0:
                return nb
            t.frag, t.pad_ = 0, 0
    return 0


class ZeroTask(C.Structure):
    _fields_ = [("p", P), ("n16", LL)]


class WreduceTask(C.Structure):
    _fields_ = [("ws", P), ("dst", P), ("nsplit", I), ("Cout", I), ("Cin", I), ("cin_valid", I), ("s_co", LL), ("s_ci", LL), ("s_t", LL)]


class RangerTask(C.Structure):
    _fields_ = [("p", P), ("g", P), ("m", P), ("v", P), ("slow", P), ("rows", I), ("cols", I), ("gc", I), ("lr", F)]


class RoiTask(C.Structure):
    _fields_ = [
        ("image", P), ("coord2d", P), ("xyz_crop", P), ("seg", P), ("trunc", P),
        ("cx", D), ("cy", D), ("scale", D), ("bw", D), ("bh", D), ("ox", D), ("oy", D), ("tz", D),
        ("H", I), ("W", I), ("x1", I), ("y1", I), ("x2", I), ("y2", I), ("cls", I), ("pad_", I),
    ]


def to_device_table(structs, device):
    """ctypes struct list -> uint8 device tensor holding the C array."""
    import torch

    arr = (type(structs[0]) * len(structs))(*structs)
    return torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)


# name -> argtypes (all return int status, except the two tile queries which return ints too)
_SIGS = {
    "gdrn_version": [],
    "gdrn_last_hip_error": [C.c_char_p, I],
    "gdrn_workspace_bytes": [I, P],
    "gdrn_device_info": [I, C.c_char_p, C.POINTER(I), C.c_char_p],
    "gdrn_conv_gemm": [C.POINTER(ConvParams), P],
    "gdrn_correspondences": [P, P, P, P, LL, I, P, P, P, F, I, I, P, P, P, P, P, P],
    "gdrn_pack_stem_w32": [P, P, I, P],
    "gdrn_stem_stats_rows": [I],
    "gdrn_stem_conv": [P, P, P, P, I, I, P],
    "gdrn_stem_conv_pool": [P, P, P, P, P, I, I, P],
    "gdrn_head_conv_tail_fwd": [P, I, P, I, P, P, P, P, I, P, I, I, I, I, I, P],
    "gdrn_head_conv_tail_loss_rows": [I, I],
    "gdrn_head_out_dgrad_rows": [I, I],
    "gdrn_head_out_dgrad": [P, I, P, I, P, I, P, P, P, P, P, I, P, I, I, I, P],
    "gdrn_head_conv_tail_loss_fwd": [P, I, P, I, P, P, P, P, I, P, I, P, P, P, P, P, I, I, I, I, P],
    "gdrn_stem_wgrad_parts": [I],
    "gdrn_stem_wgrad": [P, P, P, P, P, P, I, P, P, I, P],
    "gdrn_linear_splitk": [P, P, P, P, I, I, I, I, I, I, I, P, I, P],
    "gdrn_conv_tile": [C.POINTER(ConvParams), C.POINTER(I), C.POINTER(I)],
    "gdrn_conv_stats_rows": [C.POINTER(ConvParams)],
    "gdrn_pack_wfrag": [P, P, I, I, I, P],
    "gdrn_pack_wfrag32": [P, P, I, I, I, P],
    "gdrn_conv3x3_wfrag": [C.POINTER(ConvParams)],
    "gdrn_conv3x3_halo": [C.POINTER(ConvParams), P],
    "gdrn_conv3x3_tile": [C.POINTER(ConvParams), C.POINTER(I), C.POINTER(I), C.POINTER(I)],
    "gdrn_conv3x3_stats_rows": [C.POINTER(ConvParams)],
    "gdrn_conv3x3_halo_waves": [C.POINTER(ConvParams)],
    "gdrn_conv_wgrad": [C.POINTER(WgradParams), P],
    "gdrn_conv3x3_wgrad": [C.POINTER(WgradParams), P],
    "gdrn_conv3x3_wgrad_ok": [C.POINTER(WgradParams)],
    "gdrn_conv3x3_wgrad_splits": [C.POINTER(WgradParams)],
    "gdrn_conv3x3_wgrad_multi": [P, P, I, I, P],
    "gdrn_conv3x3_wgrad_multi_lds": [P, P, I, I, I, P],
    "gdrn_wgrad_reduce_multi": [P, P, I, I, P],
    "gdrn_pack4": [P, P, I, I, I, I, I, I, I, LL, LL, LL, LL, I, I, P],
    "gdrn_unpack4": [P, P, I, I, I, I, I, I, I, LL, LL, LL, LL, I, P],
    "gdrn_pack_stem_w": [P, P, I, P],
    "gdrn_unpack_stem_w": [P, P, P],
    "gdrn_pack_image": [P, P, I, I, I, I, I, I, P],
    "gdrn_cast_from_f32": [P, P, LL, I, P],
    "gdrn_cast_to_f32": [P, P, LL, I, P],
    "gdrn_nhwc_to_nchw_f32": [P, I, I, I, P, I, I, I, P],
    "gdrn_bn_finalize": [P, I, I, D, P, P, P, P, P, F, F, P, P, P, P, P, P],
    "gdrn_bn_eval_params": [P, P, P, P, F, I, P, P, P],
    "gdrn_bn_apply": [P, P, P, P, P, LL, I, I, I, P],
    "gdrn_bn_bwd_reduce_rows": [LL, I, I],
    "gdrn_bn_bwd_reduce": [P, P, P, P, P, P, P, LL, I, P, I, P],
    "gdrn_bn_bwd_coef": [P, I, I, LL, P, P, P, P, P, P, P, P, P],
    "gdrn_bn_bwd_apply": [P, P, P, P, P, P, P, P, LL, I, P, P, I, P],
    "gdrn_bn_relu_maxpool_fwd": [P, P, P, P, P, I, I, I, I, I, P],
    "gdrn_maxpool_bwd_rows": [I, I, I, I, I],
    "gdrn_maxpool_bwd": [P, P, P, P, P, P, I, I, I, I, P, P, P, I, P],
    "gdrn_upsample2x_fwd": [P, P, I, I, I, I, I, P],
    "gdrn_upsample2x_bwd": [P, P, I, I, I, I, I, P],
    "gdrn_conv3x3s2_ok": [C.POINTER(S2Params)],
    "gdrn_conv3x3s2_stats_rows": [C.POINTER(S2Params)],
    "gdrn_conv3x3s2": [C.POINTER(S2Params), P],
    "gdrn_conv3x3s2_dgrad_ok": [C.POINTER(S2dParams)],
    "gdrn_conv3x3s2_dgrad_rows": [C.POINTER(S2dParams)],
    "gdrn_conv3x3s2_dgrad": [C.POINTER(S2dParams), P],
    "gdrn_block64_eval_ok": [I, I, I, I],
    "gdrn_block64_eval": [P, P, P, P, P, P, I, I, I, I, P],
    "gdrn_bn_relu_upsample2x_fwd": [P, P, P, P, I, I, I, I, I, P],
    "gdrn_upsample2x_bwd_bnsums": [P, P, P, P, P, P, P, I, I, I, I, P, I, P],
    "gdrn_gn_relu_fwd": [P, P, P, P, P, I, I, I, I, F, I, P],
    "gdrn_gn_relu_bwd": [P, P, P, P, P, P, P, P, I, I, I, I, I, P],
    "gdrn_leaky_bwd": [P, P, P, LL, I, P],
    "gdrn_bias_grad": [P, I, I, I, P, I, P],
    "gdrn_head_tail_fwd": [P, I, P, P, P, I, I, I, I, I, P],
    "gdrn_head_tail_loss_fwd": [P, I, P, P, P, I, P, P, P, P, P, I, I, I, I, P],
    "gdrn_map_loss_fwd": [P, I, P, P, P, P, I, I, I, P, P],
    "gdrn_map_loss_finalize": [P, I, I, P, P],
    "gdrn_head_tail_loss_rows": [I, I, I, I, I],
    "gdrn_map_loss_finalize_rows": [P, I, I, I, P, P],
    "gdrn_loss_finalize": [P, I, I, I, P, P, P, P, P],
    "gdrn_linear_splits": [I, I],
    "gdrn_head_tail_bwd": [P, I, P, P, I, P, P, P, P, P, P, P, P, I, I, I, I, I, P],
    "gdrn_pose_loss": [C.POINTER(PoseParams), P],
    "gdrn_combine3": [P, P, P, I, P],
    "gdrn_ranger_step": [P, P, P, P, P, I, I, I, F, F, F, F, F, F, I, I, F, P],
    "gdrn_pack_chunk": [],
    "gdrn_zero_chunk": [],
    "gdrn_zero_multi": [P, P, I, I, P],
    "gdrn_nonfinite_flag": [P, LL, P, P],
    "gdrn_pack_multi": [P, P, I, I, I, P],
    "gdrn_pack_transpose_blocks": [P],
    "gdrn_unpack_multi": [P, P, I, I, P],
    "gdrn_ranger_multi": [P, P, I, I, F, F, F, F, F, I, I, F, F, P],
    "gdrn_ranger_multi_dyn": [P, P, I, I, F, F, F, F, I, I, F, F, P, P],
    "gdrn_loss_scale_update": [P, I, P],
    "gdrn_unscale_or_zero": [P, LL, F, P, P],
    "gdrn_scaled_loss_weights": [P, P, I, P, P, P],
    "gdrn_roi_affine": [P, I, I, I, P, P, P, P, P],
    "gdrn_roi_crop_inputs": [P, P, I, I, I, C.POINTER(D), C.POINTER(D), P, P, P],
    "gdrn_roi_targets": [P, P, I, I, P, I, P, P, P, P, P, P, P],
}

_SIGS["gdrn_half_format"] = []
EXPORTS = tuple(_SIGS.keys())
_libs = {}


def lib_path(dtype=BF16):
    return LIB_PATH_F16 if dtype == F16 else LIB_PATH


def load(dtype=BF16):
    """Load (once) and return the ctypes handle of the library build that computes `dtype` (F32 and BF16: libgdrn_hip.so; F16:
    libgdrn_hip_f16.so -- identical entry points, IEEE half as the 16-bit format); raises GdrnHipError if the library is absent."""
    kind = F16 if dtype == F16 else BF16
    if kind in _libs:
        return _libs[kind]
    # The library shares the host framework's HIP runtime: streams and device pointers cross the boundary.  torch ships its own libamdhip64
    # and must map it FIRST, so that the loader binds libgdrn_hip.so to that copy (same SONAME) -- dlopen'ed before `import torch`, the library
    # pulls /opt/rocm's runtime in instead, torch then runs on a runtime it was not built with and the first kernel launch of the process fails
    # with hipErrorNoDevice (r5: `python __graft_entry__.py smoke`, where build() loads the libraries before anything imported torch).
    import torch  # noqa: F401

    path = lib_path(kind)
    if not os.path.exists(path):
        raise GdrnHipError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the product path."
        )
    lib = C.CDLL(path)
    for name, args in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = LL if name == "gdrn_workspace_bytes" else I
    if lib.gdrn_half_format() != kind:
        raise GdrnHipError(f"{path} computes 16-bit dtype code {lib.gdrn_half_format()}, expected {kind}")
    _libs[kind] = lib
    return lib


def check(status, what=""):
    if status != 0:
        names = {-1: "invalid argument", -2: "unsupported shape", -3: "launch failure"}
        detail = ""
        if status == -3:   # which HIP error the launch reported (whichever build of the library made the call: ask both that are loaded)
            for lib in _libs.values():
                buf = C.create_string_buffer(64)
                code = lib.gdrn_last_hip_error(buf, 64)
                if code:
                    detail = f" (HIP error {code}: {buf.value.decode(errors='replace')})"
                    break
        raise GdrnHipError(f"libgdrn_hip: {what} failed: {names.get(status, status)}{detail}")


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()
