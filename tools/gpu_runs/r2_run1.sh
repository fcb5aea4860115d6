#!/bin/bash
# round 2, run 1: fused BatchNorm applies (xf modes of the halo conv) -- kernel + e2e tests, bench A/B, kernel stats
mkdir -p gpurun_out/r2_1
O=gpurun_out/r2_1
export PYTHONUNBUFFERED=1
R=$PWD
timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_e2e_gpu.py::test_other_baseline_configs_full_size -k "operand_transform or transform_with_fused or bn_bwd_coef or bn_finalize or batchnorm" > $O/pytest_new.log 2>&1
tail -15 $O/pytest_new.log
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1
tail -25 $O/pytest_all.log
GDRN_LAYER_TABLE=$O/layers_fused.txt timeout 600 python bench.py --no-cpu-baseline > $O/bench_fused.json 2> $O/bench_fused.err
cat $O/bench_fused.json
GDRN_FUSE_XF=0 GDRN_LAYER_TABLE=$O/layers_unfused.txt timeout 600 python bench.py --no-cpu-baseline > $O/bench_unfused.json 2> $O/bench_unfused.err
cat $O/bench_unfused.json
GDRN_GRAPH=1 timeout 600 python bench.py --no-cpu-baseline --no-roofline --no-extras > $O/bench_graph.json 2> $O/bench_graph.err
cat $O/bench_graph.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o p -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extras > $R/$O/bench_prof.log 2>&1
cd $R
python tools/summarize_stats.py $(ls $O/prof/*/p_kernel_stats.csv $O/prof/p_kernel_stats.csv 2>/dev/null | head -1) 13 "round 2 run 1 (fused BN applies)" > $O/kernel_stats.txt 2>&1
head -60 $O/kernel_stats.txt
