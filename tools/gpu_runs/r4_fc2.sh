#!/bin/bash
O=gpurun_out/r4_fc2
mkdir -p $O
export PYTHONUNBUFFERED=1
b() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 6 "$@" 2>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
for rep in 1 2 3; do echo "fc2 gather: $(GDRN_FC2_SPLITK=0 b)  fc2 split-K: $(GDRN_FC2_SPLITK=1 b)   inference: $(GDRN_FC2_SPLITK=0 b --fwd-only) / $(GDRN_FC2_SPLITK=1 b --fwd-only)"; done | tee $O/ab.txt
timeout 1500 python -m pytest tests/test_e2e_gpu.py tests/test_teacher_forced_gpu.py tests/test_fp16_gpu.py -q -m gpu > $O/tests.log 2>&1; echo "rc $?" >> $O/tests.log; grep -E "passed|failed|^FAILED|^rc" $O/tests.log | tail -5
