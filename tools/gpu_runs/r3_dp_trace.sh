#!/bin/bash
O=$PWD/gpurun_out/r3_dp_trace
mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
GDRN_DIST_TRACE=0 timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/t -o p -- python $R/bench.py --no-cpu-baseline --no-roofline --no-extras --steps 8 --warmup 3 --dist-force > $O/log.txt 2>&1
cd $R
f=$(ls $O/t/*/p_kernel_trace.csv $O/t/p_kernel_trace.csv 2>/dev/null | head -1)
python tools/trace_steps.py $f 5 "dist-force" > $O/steps.txt 2>&1
python tools/overlap_trace.py $f > $O/overlap.txt 2>&1
rm -rf $O/t
head -40 $O/steps.txt | cut -c1-150
cat $O/overlap.txt | cut -c1-170
