#!/bin/bash
# round 4: the returned loss vector formed in front of the backward chain instead of behind the final stream join
O=gpurun_out/r4_lossmul
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_e2e_gpu.py -q -m gpu -x -k "train_step or reference or graph or optimizer or hooks or fused" > $O/tests.log 2>&1; echo "rc $?" >> $O/tests.log; grep -E "passed|failed|^FAILED|^rc" $O/tests.log | tail -4
b() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 6 "$@" 2>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
for rep in 1 2 3; do echo "train: $(b)"; done | tee $O/ab.txt
