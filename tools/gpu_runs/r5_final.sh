#!/bin/bash
# round 5: the whole GPU suite + smoke as the driver runs them (build() and smoke() in ONE process too), then the profile set of the final sources
mkdir -p gpurun_out/r5_final
export PYTHONUNBUFFERED=1
timeout 1800 python -m pytest tests -q -m gpu > gpurun_out/r5_final/gputests.log 2>&1; echo "rc $?" >> gpurun_out/r5_final/gputests.log; grep -E "passed|failed|^FAILED|^rc|^ERROR" gpurun_out/r5_final/gputests.log | tail -8
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5_final/smoke.log 2>&1; echo "smoke rc $?"; grep "^smoke" gpurun_out/r5_final/smoke.log
timeout 600 python __graft_entry__.py smoke > gpurun_out/r5_final/smoke_after_build.log 2>&1; echo "build+smoke rc $?"
bash tools/gpu_runs/r5_profiles.sh
