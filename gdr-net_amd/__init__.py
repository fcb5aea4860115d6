"""gdr-net_amd -- MI355X-native (gfx950) implementation of GDR-Net's per-RoI hot path.

Imported as ``gdrnet_amd`` (see ``gdrnet_amd/__init__.py``).  Sub-modules:

* ``cabi``    -- ctypes loader of the C-ABI HIP library ``libgdrn_hip.so`` (include/gdrn_hip.h)
* ``cfg``     -- attribute-dict config with the reference's ``MODEL.CDPN`` keys
* ``synth``   -- repo-owned hash RNG, deterministic weights, synthetic RoI batches
* ``engine``  -- forward / backward executor of the path on HIP kernels
* ``GDRN``    -- drop-in for ``core/gdrn_modeling/models/GDRN.py`` (GDRN, build_model_optimizer)
* ``ranger``  -- fused Ranger optimizer step on HIP
* ``dist``    -- RCCL gradient all-reduce for the one-process-per-GPU data-parallel path
* ``postproc`` -- on-device inference post-processing (correspondence extraction for PnP-RANSAC)
* ``roi_data`` -- GPU RoI cropper / target builder (the data loader's warpAffine crops, masks, region labels)
* ``checkpoint`` -- MyCheckpointer / PeriodicCheckpointer in the reference's file format
"""
__version__ = "0.1.0"
