#!/bin/bash
# HBM-side traffic of every kernel of the step: two separate PMC passes (FETCH_SIZE and WRITE_SIZE do not fit one pass)
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
R=$PWD
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -o p -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-roofline --no-extras > $R/gpurun_out/pmc_$c.log 2>&1
done
cd $R
ls gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE | head
