"""Every kernel-level parity test of tests/test_kernels_gpu.py a second time on libgdrn_hip_f16.so -- the same kernel sources built with IEEE
half as the 16-bit format (csrc/common.h, -DGDRN_HALF_F16: v_mfma_f32_*_f16, fp16 storage): the arithmetic of the reference's fp16 autocast
(core/gdrn_modeling/main_gdrn.py:53-56,141; gdrn_evaluator.py:568).  The module source is executed again with `BF16` bound to the fp16 dtype
code, so the operands are rounded to fp16, the references see the same operands, and the output tolerance is 8e-4 instead of bf16's 6e-3
(2^-12 against 2^-9 output rounding).  The fp32 instantiations are not repeated."""
import importlib.util
import os

from gdrnet_amd import cabi

_src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "test_kernels_gpu.py")
_spec = importlib.util.spec_from_file_location("_test_kernels_fp16_impl", _src)
_mod = importlib.util.module_from_spec(_spec)
_mod.__dict__["__HALF__"] = cabi.F16
_spec.loader.exec_module(_mod)
assert _mod.IS_F16 and _mod.TOL[cabi.F16] == 8e-4

pytestmark = _mod.pytestmark
H = _mod.H   # the module-scoped fixture (loads the fp16 library)
for _k, _v in list(vars(_mod).items()):
    if _k.startswith("test_"):
        globals()[_k] = _v
