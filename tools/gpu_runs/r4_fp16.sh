#!/bin/bash
# round 4: the fp16 library build -- kernel tests, end-to-end tests, stage-by-stage test, smoke, step time against bf16
O=gpurun_out/r4_fp16
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_kernels_fp16_gpu.py -x -q -m gpu > $O/kernels.log 2>&1; echo "rc $?" >> $O/kernels.log; tail -4 $O/kernels.log
timeout 900 python -m pytest tests/test_fp16_gpu.py -x -q -m gpu -s > $O/e2e.log 2>&1; echo "rc $?" >> $O/e2e.log; grep -v "^Randomly" $O/e2e.log | tail -25
timeout 900 python -m pytest tests/test_teacher_forced_gpu.py -x -q -m gpu -s -k "fp16" > $O/tf.log 2>&1; echo "rc $?" >> $O/tf.log; grep -v "^Randomly" $O/tf.log | tail -30
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc $?" >> $O/smoke.log; grep -v "^Randomly" $O/smoke.log | tail -8
b() { python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 6 "$@" 2>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
for rep in 1 2; do echo "bf16: $(b --dtype bf16)   fp16: $(b --dtype fp16)"; done | tee $O/ab.txt
