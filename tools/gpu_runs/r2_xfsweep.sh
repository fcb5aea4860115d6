#!/bin/bash
mkdir -p gpurun_out/r2_xfsweep
O=gpurun_out/r2_xfsweep
B="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 40 --warmup 10"
run() { name=$1; shift; env "$@" timeout 300 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-24s %.3f ms/step' % ('$name', d['ms_per_step']))"; }
for r in 1 2; do
run all GDRN_XF_MASK=15
run unfused GDRN_FUSE_XF=0
run minc128 GDRN_XF_MINC=128
run minc256 GDRN_XF_MINC=256
run m1_only GDRN_XF_MASK=1
run m13 GDRN_XF_MASK=5
run m134 GDRN_XF_MASK=13
run m13_minc128 GDRN_XF_MASK=5 GDRN_XF_MINC=128
run m1_minc128 GDRN_XF_MASK=1 GDRN_XF_MINC=128
done
