#!/bin/bash
# A/B of the working tree against _ab/base (previous commit, built separately) + kernel parity tests selected by $1
mkdir -p gpurun_out/r2_ab3
O=gpurun_out/r2_ab3
K=${1:-wgrad}
timeout 900 python -m pytest tests -m gpu -q -x -k "$K" > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log
B="bench.py --no-cpu-baseline --no-roofline --no-extras --steps 40 --warmup 10"
for r in 1 2 3; do
  for t in base new; do
    if [ $t = base ]; then d=_ab/base; else d=.; fi
    (cd $d && timeout 300 python $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t round $r: %.3f ms/step' % d['ms_per_step'])")
  done
done
GDRN_LAYER_TABLE=$O/layers.txt timeout 600 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $O/bench.json 2>/dev/null
grep -i "wgrad" $O/layers.txt | head -8
