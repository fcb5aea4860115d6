#!/bin/bash
# round 6: quick check of a plan / kernel change -- step and inference timing (3 runs each), then the suites that walk the step
O=$PWD/gpurun_out/r6_quick
mkdir -p $O
export PYTHONUNBUFFERED=1
b() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 6 "$@" 2>>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
i() { timeout 300 python bench.py --fwd-only --no-cpu-baseline --no-roofline --no-extras --steps 100 --warmup 10 "$@" 2>>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
echo "train: $(b) $(b) $(b)   inference: $(i) $(i) $(i)" | tee $O/ab.txt
timeout 1800 python -m pytest tests/test_teacher_forced_gpu.py tests/test_e2e_gpu.py tests/test_fp16_gpu.py -q -m gpu -x ${QUICK_K:+-k "$QUICK_K"} > $O/e2e.log 2>&1; echo "rc $?" >> $O/e2e.log; grep -E "passed|failed|^FAILED|^rc|^E  " $O/e2e.log | tail -8
