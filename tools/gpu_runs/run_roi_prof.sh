#!/bin/bash
# N3 RoI cropper: rocprofv3 kernel stats of tools/roibench.py + a default bench line.  Usage: gpurun --timeout 900 -- 'bash tools/gpu_runs/run_roi_prof.sh'
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/roi_prof -- python $R/tools/roibench.py 64 > $R/gpurun_out/roi_prof.log 2>&1 )
f=$(find gpurun_out/roi_prof -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && grep -i "roi_\|Name" "$f" | head -12 | tee gpurun_out/roi_kernel_stats.csv
tail -3 gpurun_out/roi_prof.log
timeout 600 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench_default.json
