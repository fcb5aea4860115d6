#!/bin/bash
# is the eager launch loop or a hipGraph replay the faster way to issue the 263 launches of the step?
O=$PWD/gpurun_out/r3_graph.txt
: > $O
timeout 120 python tools/ubench/launch_gap.py 2>&1 | grep -v amdgpu.ids >> $O
B="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 8"
for e in "X=1" "GDRN_GRAPH=1" "GDRN_WGRAD_STREAM=0" "GDRN_WGRAD_STREAM=0 GDRN_GRAPH=1" "X=1" "GDRN_GRAPH=1"; do
  echo "== $e" >> $O
  env $e timeout 300 $B 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-330 >> $O
done
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/r3_gap_trace -o p -- python $OLDPWD/tools/ubench/launch_gap.py > /dev/null 2>&1
cd $OLDPWD
f=$(ls gpurun_out/r3_gap_trace/*/p_kernel_stats.csv gpurun_out/r3_gap_trace/p_kernel_stats.csv 2>/dev/null | head -1)
head -5 $f | cut -c1-200 >> $O
rm -rf gpurun_out/r3_gap_trace
cat $O
