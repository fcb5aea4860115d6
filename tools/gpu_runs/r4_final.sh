#!/bin/bash
# round 4: profile set of the final sources, then the whole GPU suite
bash tools/gpu_runs/r4_profiles.sh > gpurun_out/r4_profiles.log 2>&1
tail -c 1500 gpurun_out/r4_profiles.log
mkdir -p gpurun_out/r4_full
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r4_full/gputest.log 2>&1; echo "rc $?" >> gpurun_out/r4_full/gputest.log; tail -6 gpurun_out/r4_full/gputest.log
