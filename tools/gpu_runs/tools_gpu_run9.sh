#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
R=$PWD
python tools_convbench.py 2>&1 | grep wgrad
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof9 -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/bench9.log 2>&1
cd $R
ls gpurun_out/prof9 | head
