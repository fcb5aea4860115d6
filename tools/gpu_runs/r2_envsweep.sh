#!/bin/bash
# one-variable sweeps of engine knobs on the current build (bench.py --no-extras, 40 timed steps each)
B="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 40 --warmup 10"
run() { name=$1; shift; env "$@" timeout 300 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s %.3f ms/step' % ('$name', d['ms_per_step']))"; }
for r in 1 2; do
run default X=1
run wgrad_blocks_512 GDRN_WGRAD_BLOCKS=512
run wgrad_blocks_1536 GDRN_WGRAD_BLOCKS=1536
run wgrad_blocks_2048 GDRN_WGRAD_BLOCKS=2048
run gemm_bnb_off GDRN_GEMM_BNB=0
run graph GDRN_GRAPH=1
done
