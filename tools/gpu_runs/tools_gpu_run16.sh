#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
R=$PWD
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for t in 0 384 512 1024 1100; do
echo "min_wg $t"; GDRN_HALO_MIN_WG=$t timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-200
done
GDRN_LAYER_TABLE=gpurun_out/layers16.txt timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof16 -o bench -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $R/gpurun_out/bench16.log 2>&1
cd $R
