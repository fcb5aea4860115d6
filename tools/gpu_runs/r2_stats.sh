#!/bin/bash
mkdir -p gpurun_out/r2_stats
O=gpurun_out/r2_stats
R=$PWD
GDRN_LAYER_TABLE=$O/layers.txt timeout 600 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $O/bench.json 2>/dev/null
grep -i "wgrad" $O/layers.txt | head -20
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -o p -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extras > $R/$O/stats.log 2>&1
cd $R
python tools/summarize_stats.py $O/stats/p_kernel_stats.csv 13 "stats" > $O/kernel_stats.txt
head -24 $O/kernel_stats.txt
