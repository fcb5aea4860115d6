#!/bin/bash
# same-box A/B: the tree exported under _ab/base (previous commit) against the working tree, alternating, 3 rounds
mkdir -p gpurun_out/r2_ab
O=gpurun_out/r2_ab
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q -x -k "wgrad or fused_batchnorm or vs_oracle or baseline_sizes or train_step_vs_reference or checkpoint" > $O/pytest_sel.log 2>&1
tail -5 $O/pytest_sel.log
B="bench.py --no-cpu-baseline --no-roofline --no-extras --steps 40 --warmup 10"
for r in 1 2 3; do
  for t in base new; do
    if [ $t = base ]; then d=_ab/base; else d=.; fi
    (cd $d && timeout 300 python $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t round $r: %.3f ms/step' % d['ms_per_step'])")
  done
done
