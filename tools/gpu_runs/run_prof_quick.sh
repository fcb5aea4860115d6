#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
R=$PWD
TAG=${1:-q}
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o bench -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extras > $R/gpurun_out/bench_prof_$TAG.log 2>&1
cd $R
tail -1 gpurun_out/bench_prof_$TAG.log | cut -c1-200
