#!/bin/bash
# round 2, run 3: deterministic reduce fix -> bit-exact fused/separate test; wgrad pow2 index math; finalize/coef prefetch
mkdir -p gpurun_out/r2_3
O=gpurun_out/r2_3
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1
tail -30 $O/pytest_all.log
GDRN_LAYER_TABLE=$O/layers.txt timeout 600 python bench.py --no-cpu-baseline --steps 30 --warmup 8 > $O/bench.json 2> $O/bench.err
cat $O/bench.json
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o p -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extras > $R/$O/bench_prof.log 2>&1
cd $R
python tools/summarize_stats.py $(ls $O/prof/*/p_kernel_stats.csv $O/prof/p_kernel_stats.csv 2>/dev/null | head -1) 13 "round 2 run 3" > $O/kernel_stats.txt 2>&1
head -45 $O/kernel_stats.txt
