"""gdr-net_amd -- MI355X-native (gfx950) implementation of GDR-Net's per-RoI hot path.

Imported as ``gdrnet_amd`` (see ``gdrnet_amd/__init__.py``).  Sub-modules:

* ``cabi``    -- ctypes loader of the C-ABI HIP library ``libgdrn_hip.so`` (include/gdrn_hip.h)
* ``cfg``     -- attribute-dict config with the reference's ``MODEL.CDPN`` keys
* ``synth``   -- repo-owned hash RNG, deterministic weights, synthetic RoI batches
* ``engine``  -- forward / backward executor of the path on HIP kernels: per-model state (operand copies, gradient buckets, streams)
* ``plan``    -- its per-batch-size launch lists (static buffers + pre-bound C-ABI calls), forward and backward
* ``GDRN``    -- drop-in for ``core/gdrn_modeling/models/GDRN.py`` (GDRN, build_model_optimizer)
* ``ranger``  -- fused Ranger optimizer step on HIP
* ``dist``    -- RCCL gradient all-reduce for the one-process-per-GPU data-parallel path
* ``postproc`` -- on-device inference post-processing (correspondence extraction for PnP-RANSAC)
* ``roi_data`` -- GPU RoI cropper / target builder (the data loader's warpAffine crops, masks, region labels)
* ``checkpoint`` -- MyCheckpointer / PeriodicCheckpointer in the reference's file format
"""
import os as _os

# HIP maps streams onto a pool of hardware queues (4 by default); with the engine's side stream, the reducer's stream and RCCL's own streams
# alive, two of them can land on the queue of the compute stream and serialise with it (measured: the data-parallel step 8.45 instead of
# 7.5 ms).  Ask for 8 queues -- effective when this package is imported before the process initialises HIP; the side / reducer streams are
# additionally created at low priority (engine.make_stream), which keeps them off the default-priority queues either way.
# A process-wide setting: applied only when the variable is unset (export GPU_MAX_HW_QUEUES yourself to keep another value), logged at INFO.
if "GPU_MAX_HW_QUEUES" not in _os.environ:
    _os.environ["GPU_MAX_HW_QUEUES"] = "8"
    import logging as _logging

    _logging.getLogger(__name__).info("GPU_MAX_HW_QUEUES=8 set for this process (export the variable to keep another value)")

__version__ = "0.1.0"
