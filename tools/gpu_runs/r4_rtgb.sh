#!/bin/bash
# round 4: fc_r / fc_t bias gradients written in place by the merged layer's one bias_grad launch (shared slot of the flat gradient buffer)
O=gpurun_out/r4_rtgb
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_e2e_gpu.py tests/test_teacher_forced_gpu.py tests/test_fp16_gpu.py -q -m gpu > $O/tests.log 2>&1; echo "rc $?" >> $O/tests.log; grep -E "passed|failed|^FAILED|^rc" $O/tests.log | tail -5
b() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 6 "$@" 2>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
for rep in 1 2 3; do echo "train: $(b)"; done | tee $O/ab.txt
