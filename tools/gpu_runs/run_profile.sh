#!/bin/bash
# full GPU check of a build: tests, smoke, bench (with live roofline + CPU baseline), rocprofv3 kernel stats
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
R=$PWD
TAG=${1:-run}
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
GDRN_LAYER_TABLE=gpurun_out/layers_$TAG.txt timeout 900 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench_$TAG.json
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o bench -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/bench_prof_$TAG.log 2>&1
cd $R
tail -1 gpurun_out/bench_prof_$TAG.log | cut -c1-300
