#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_fwd -o bench -- python $R/bench.py --fwd-only --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $R/gpurun_out/bench_fwd.log 2>&1
cd $R
tail -1 gpurun_out/bench_fwd.log | cut -c1-300
