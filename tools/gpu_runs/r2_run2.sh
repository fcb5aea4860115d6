#!/bin/bash
# round 2, run 2: deterministic BN backward (rows + coef), bit-exact fused-vs-separate e2e test, XF policy sweep
mkdir -p gpurun_out/r2_2
O=gpurun_out/r2_2
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1
tail -40 $O/pytest_all.log
B="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 8"
run() { name=$1; shift; env "$@" timeout 300 $B > $O/bench_$name.json 2> $O/bench_$name.err; python -c "
import json,sys
try:
    d=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1]); print('%-28s %.3f ms/step %8.1f RoI/s' % ('$name', d['ms_per_step'], d['value']))
except Exception as e: print('$name', 'FAILED', e)
"; }
run unfused GDRN_FUSE_XF=0
run all GDRN_XF_MASK=15
run m1 GDRN_XF_MASK=1
run m12 GDRN_XF_MASK=3
run m1_hw32 GDRN_XF_MASK=1 GDRN_XF_MAXHW=32
run m12_hw32 GDRN_XF_MASK=3 GDRN_XF_MAXHW=32
run m123_hw32 GDRN_XF_MASK=7 GDRN_XF_MAXHW=32
run all_hw32 GDRN_XF_MASK=15 GDRN_XF_MAXHW=32
run all_hw16 GDRN_XF_MASK=15 GDRN_XF_MAXHW=16
run m13 GDRN_XF_MASK=5
run unfused_graph GDRN_FUSE_XF=0 GDRN_GRAPH=1
run all_graph GDRN_XF_MASK=15 GDRN_GRAPH=1
run unfused_wstream GDRN_FUSE_XF=0 GDRN_WGRAD_STREAM=1
