#!/bin/bash
# A/B + layer table of the wgrad kernels
mkdir -p gpurun_out/r2_ab2
O=gpurun_out/r2_ab2
timeout 600 python -m pytest tests -m gpu -q -x -k "wgrad" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
B="bench.py --no-cpu-baseline --no-roofline --no-extras --steps 40 --warmup 10"
for r in 1 2 3; do
  for t in base new; do
    if [ $t = base ]; then d=_ab/base; else d=.; fi
    (cd $d && timeout 300 python $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t round $r: %.3f ms/step' % d['ms_per_step'])")
  done
done
GDRN_LAYER_TABLE=$O/layers.txt timeout 600 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $O/bench.json 2>/dev/null
grep -i "wgrad" $O/layers.txt | head -8
(cd _ab/base && GDRN_LAYER_TABLE=/root/repo/$O/layers_base.txt timeout 600 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 > /dev/null 2>&1)
grep -i "wgrad" $O/layers_base.txt | head -14
