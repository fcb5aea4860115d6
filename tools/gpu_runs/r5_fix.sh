#!/bin/bash
# round 5: the tests that failed in r5_full2 after their fixes, the parity-mode seeds against the fp64 oracle under both kernel policies, smoke
O=gpurun_out/r5_fix
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 600 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc $?"; grep "smoke\|Error" $O/smoke.log
timeout 1200 python -m pytest tests/test_e2e_gpu.py tests/test_fp16_gpu.py tests/test_kernels_gpu.py -q -m gpu -s -k "seeds_at_bs64 or bf16_train_step_vs_reference or fused_batchnorm_applies or conditioned_network or nonfinite or overflow or per_bucket_optimizer_counts" > $O/tests.log 2>&1; echo "rc $?" >> $O/tests.log
grep -E "fp32 \[|passed|failed|^FAILED|^rc" $O/tests.log
