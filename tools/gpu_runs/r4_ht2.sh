#!/bin/bash
# round 4: head tail kernels with every load hoisted to the top of the iteration + map-loss sums as per-workgroup partial rows (GDRN_ACC_ROWS)
O=gpurun_out/r4_ht2
mkdir -p $O
export PYTHONUNBUFFERED=1
python tools/htbench.py 2>&1 | grep -v "amdgpu.ids" | tee $O/ht.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_kernels_fp16_gpu.py -q -m gpu -k "head_tail or loss or pose" > $O/ktests.log 2>&1; echo "rc $?" >> $O/ktests.log; grep -E "passed|failed|^FAILED|^rc" $O/ktests.log | tail -5
timeout 1500 python -m pytest tests/test_e2e_gpu.py tests/test_teacher_forced_gpu.py tests/test_fp16_gpu.py -q -m gpu -x > $O/tests.log 2>&1; echo "rc $?" >> $O/tests.log; grep -E "passed|failed|^FAILED|^rc" $O/tests.log | tail -5
b() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 6 "$@" 2>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
for rep in 1 2 3; do echo "train: atomics $(GDRN_LOSS_ROWS=0 b)  rows $(b)   inference: $(b --fwd-only)"; done | tee $O/ab.txt
