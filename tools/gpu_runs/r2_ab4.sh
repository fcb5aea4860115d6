#!/bin/bash
# halo-kernel change: transform tests, halotime (xf modes 1 / 3) of both trees, step A/B
mkdir -p gpurun_out/r2_ab4
O=gpurun_out/r2_ab4
timeout 900 python -m pytest tests -m gpu -q -x -k "halo or fused_batchnorm" > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log
echo "--- base"; (cd _ab/base && timeout 300 python tools/halotime.py 0,1,3 2>&1 | grep "C=")
echo "--- new"; timeout 300 python tools/halotime.py 0,1,3 2>&1 | grep "C="
B="bench.py --no-cpu-baseline --no-roofline --no-extras --steps 40 --warmup 10"
for r in 1 2 3; do
  for t in base new; do
    if [ $t = base ]; then d=_ab/base; else d=.; fi
    (cd $d && timeout 300 python $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t round $r: %.3f ms/step' % d['ms_per_step'])")
  done
done
