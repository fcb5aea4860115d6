#!/bin/bash
# round 5: the whole GPU suite + smoke as the driver runs them (build() and smoke() in ONE process too), the figures DESIGN.md quotes from tests
# that print them, then the profile set of the final sources
mkdir -p gpurun_out/r5_final
export PYTHONUNBUFFERED=1
timeout 1800 python -m pytest tests -q -m gpu > gpurun_out/r5_final/gputests.log 2>&1; echo "rc $?" >> gpurun_out/r5_final/gputests.log; grep -E "passed|failed|^FAILED|^rc|^ERROR" gpurun_out/r5_final/gputests.log | tail -8
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5_final/smoke.log 2>&1; echo "smoke rc $?"; grep "^smoke" gpurun_out/r5_final/smoke.log
timeout 600 python __graft_entry__.py smoke > gpurun_out/r5_final/smoke_after_build.log 2>&1; echo "build+smoke rc $?"
timeout 900 python -m pytest tests/test_e2e_gpu.py tests/test_fp16_gpu.py -q -m gpu -s -k "baseline_sizes or g10 or seeds_at_bs64 or fp32_pose_parity_over_seeds or fp32_train_step_vs_reference" 2>&1 | grep -E "^fp32|^G10|pose rel-err|passed|failed" > gpurun_out/r5_final/figures.txt; tail -3 gpurun_out/r5_final/figures.txt
bash tools/gpu_runs/r5_profiles.sh
