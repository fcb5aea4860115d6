#!/bin/bash
# round 4: the whole GPU suite + smoke as the driver runs them, then the profile set of the final sources
mkdir -p gpurun_out/r4_full
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r4_full/gputest.log 2>&1; echo "rc $?" >> gpurun_out/r4_full/gputest.log; grep -E "passed|failed|^FAILED|^rc" gpurun_out/r4_full/gputest.log | tail -8
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4_full/smoke.log 2>&1; echo "rc $?" >> gpurun_out/r4_full/smoke.log; grep -v "^Randomly" gpurun_out/r4_full/smoke.log | tail -6
bash tools/gpu_runs/r4_profiles.sh > gpurun_out/r4_profiles.log 2>&1
tail -c 600 gpurun_out/r4_profiles.log
