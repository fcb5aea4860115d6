"""Config objects for the GDR-Net hot path.

The reference reads an ``mmcv.Config`` (attribute access + ``.get`` + ``.pop``) at
``core/gdrn_modeling/models/GDRN.py:111-113,503,550-724``.  mmcv is not a dependency of
this package; ``CfgNode`` is a minimal attribute-dict that satisfies exactly those
accesses, and a real ``mmcv.Config`` can be passed instead (only attribute access,
``.get`` and ``PNP_HEAD_CFG.pop`` are used).

``lm13_cfg`` / ``lmo_cfg`` / ``ycbv_cfg`` reproduce the *values* of the keys the hot path
reads from ``configs/_base_/gdrn_base.py:5-132`` merged with
``configs/gdrn/lm/a6_cPnP_lm13.py:40-68`` (and the LM-O / YCB-V overrides,
SURVEY.md section 8): the network graph is identical for all of them; only batch size and
``PM_LOSS_SYM`` differ.
"""
import copy


class CfgNode(dict):
    """dict with attribute access (recursively applied to nested dicts)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, dict) and not isinstance(v, CfgNode):
            return CfgNode(v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return CfgNode({k: copy.deepcopy(v, memo) for k, v in self.items()})

    def merge(self, other):
        for k, v in other.items():
            if isinstance(v, dict) and isinstance(self.get(k), dict):
                self[k].merge(v)
            else:
                self[k] = v
        return self


def _base():
    # values of configs/_base_/gdrn_base.py:5-132 (only keys the hot path reads)
    return CfgNode(
        MODEL=dict(
            DEVICE="cuda",
            WEIGHTS="",
            PIXEL_MEAN=[0, 0, 0],  # gdrn_base.py:12-13: images to [0, 1]
            PIXEL_STD=[255.0, 255.0, 255.0],
            CDPN=dict(
                NAME="GDRN",
                TASK="rot",
                USE_MTL=False,
                BACKBONE=dict(
                    PRETRAINED="",
                    ARCH="resnet",
                    NUM_LAYERS=34,
                    INPUT_CHANNEL=3,
                    INPUT_RES=256,
                    OUTPUT_RES=64,
                    FREEZE=False,
                ),
                ROT_HEAD=dict(
                    FREEZE=False,
                    ROT_CONCAT=False,
                    XYZ_BIN=64,
                    NUM_LAYERS=3,
                    NUM_FILTERS=256,
                    CONV_KERNEL_SIZE=3,
                    NORM="BN",
                    NUM_GN_GROUPS=32,
                    OUT_CONV_KERNEL_SIZE=1,
                    NUM_CLASSES=13,
                    ROT_CLASS_AWARE=False,
                    XYZ_LOSS_TYPE="L1",
                    XYZ_LOSS_MASK_GT="visib",
                    XYZ_LW=1.0,
                    MASK_CLASS_AWARE=False,
                    MASK_LOSS_TYPE="L1",
                    MASK_LOSS_GT="trunc",
                    MASK_LW=1.0,
                    MASK_THR_TEST=0.5,
                    NUM_REGIONS=8,
                    REGION_CLASS_AWARE=False,
                    REGION_LOSS_TYPE="CE",
                    REGION_LOSS_MASK_GT="visib",
                    REGION_LW=1.0,
                ),
                PNP_NET=dict(
                    FREEZE=False,
                    R_ONLY=False,
                    LR_MULT=1.0,
                    PNP_HEAD_CFG=dict(type="ConvPnPNet", norm="GN", num_gn_groups=32, drop_prob=0.0),
                    WITH_2D_COORD=False,
                    REGION_ATTENTION=False,
                    MASK_ATTENTION="none",
                    ROT_TYPE="ego_rot6d",
                    TRANS_TYPE="centroid_z",
                    Z_TYPE="REL",
                    NUM_PM_POINTS=3000,
                    PM_LOSS_TYPE="L1",
                    PM_SMOOTH_L1_BETA=1.0,
                    PM_LOSS_SYM=False,
                    PM_NORM_BY_EXTENT=False,
                    PM_R_ONLY=True,
                    PM_DISENTANGLE_T=False,
                    PM_DISENTANGLE_Z=False,
                    PM_T_USE_POINTS=False,
                    PM_LW=1.0,
                    ROT_LOSS_TYPE="angular",
                    ROT_LW=0.0,
                    CENTROID_LOSS_TYPE="L1",
                    CENTROID_LW=0.0,
                    Z_LOSS_TYPE="L1",
                    Z_LW=0.0,
                    TRANS_LOSS_TYPE="L1",
                    TRANS_LOSS_DISENTANGLE=True,
                    TRANS_LW=0.0,
                    BIND_LOSS_TYPE="L1",
                    BIND_LW=0.0,
                ),
                TRANS_HEAD=dict(ENABLED=False, FREEZE=True, LR_MULT=1.0),
            ),
        ),
        SOLVER=dict(
            IMS_PER_BATCH=24,
            BASE_LR=1e-4,
            WEIGHT_DECAY=0.0,
            OPTIMIZER_CFG=dict(type="Ranger", lr=1e-4, weight_decay=0),
            # schedule: configs/_base_/common_base.py:106-124 overridden by configs/gdrn/lm/a6_cPnP_lm13.py:22-32
            TOTAL_EPOCHS=160,
            GAMMA=0.1,
            LR_SCHEDULER_NAME="flat_and_anneal",
            WARMUP_METHOD="linear",
            WARMUP_FACTOR=0.001,
            WARMUP_ITERS=1000,
            ANNEAL_METHOD="cosine",
            ANNEAL_POINT=0.72,
            POLY_POWER=0.9,
            REL_STEPS=(0.5, 0.75),
            CHECKPOINT_PERIOD=5,
            CHECKPOINT_BY_EPOCH=True,
            AMP=dict(ENABLED=False),   # configs/_base_/common_base.py:130 (True in the 30 single-object AMP configs): fp16 kernels + loss scale here
        ),
        INPUT=dict(DZI_TYPE="uniform", DZI_PAD_SCALE=1.5, DZI_SCALE_RATIO=0.25, DZI_SHIFT_RATIO=0.25, SMOOTH_XYZ=False),  # common_base.py:49,53; a6_cPnP_lm13.py:5
        TEST=dict(USE_PNP=False, AMP_TEST=False),   # AMP_TEST: common_base.py:173 -> fp16 inference (gdrn_evaluator.py:568)
    )


def lm13_cfg(device="cuda", **over):
    """LM 13-object config: configs/gdrn/lm/a6_cPnP_lm13.py:40-68 over gdrn_base.py."""
    cfg = _base()
    cfg.MODEL.DEVICE = device
    cfg.MODEL.CDPN.ROT_HEAD.merge(dict(NUM_REGIONS=64, NUM_CLASSES=13))
    cfg.MODEL.CDPN.PNP_NET.merge(
        dict(
            REGION_ATTENTION=True,
            WITH_2D_COORD=True,
            ROT_TYPE="allo_rot6d",
            TRANS_TYPE="centroid_z",
            PM_NORM_BY_EXTENT=True,
            PM_R_ONLY=True,
            CENTROID_LOSS_TYPE="L1",
            CENTROID_LW=1.0,
            Z_LOSS_TYPE="L1",
            Z_LW=1.0,
        )
    )
    cfg.merge(over)
    return cfg


def lmo_cfg(device="cuda", **over):
    """LM-O config (configs/gdrn/lmo/a6_cPnP_AugAAETrunc_BG0.5_lmo_real_pbr0.1_40e.py): same graph, 8 classes."""
    cfg = lm13_cfg(device)
    cfg.MODEL.CDPN.ROT_HEAD.NUM_CLASSES = 8
    cfg.merge(over)
    return cfg


def ycbv_cfg(device="cuda", **over):
    """YCB-V config (configs/gdrn/ycbv/a6_cPnP_AugAAETrunc_BG0.5_Rsym_ycbv_real_pbr_visib20_10e.py:85):
    same graph, 21 classes, symmetric PM loss."""
    cfg = lm13_cfg(device)
    cfg.MODEL.CDPN.ROT_HEAD.NUM_CLASSES = 21
    cfg.MODEL.CDPN.PNP_NET.PM_LOSS_SYM = True
    cfg.merge(over)
    return cfg
