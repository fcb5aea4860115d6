#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
R=$PWD
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
GDRN_LAYER_TABLE=gpurun_out/layers10.txt timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof10 -o bench -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $R/gpurun_out/bench10.log 2>&1
cd $R
ls gpurun_out/prof10 | head
