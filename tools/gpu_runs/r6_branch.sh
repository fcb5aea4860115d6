#!/bin/bash
# round 6: the residual blocks' downsample branch on a third stream beside the main branch (Plan._branch): A/B of the step, then the suites that
# walk every stage of the step (teacher-forced), the fused-vs-separate bit-equality and the end-to-end parity tests
O=gpurun_out/r6_branch
mkdir -p $O
export PYTHONUNBUFFERED=1
b() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 6 "$@" 2>>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
echo "branch stream on:  $(b) $(b) $(b)" | tee $O/ab.txt
echo "branch stream off: $(GDRN_BRANCH_STREAM=0 b) $(GDRN_BRANCH_STREAM=0 b) $(GDRN_BRANCH_STREAM=0 b)" | tee -a $O/ab.txt
echo "branch stream on:  $(b) $(b)" | tee -a $O/ab.txt
timeout 1500 python -m pytest tests/test_teacher_forced_gpu.py tests/test_e2e_gpu.py -q -m gpu -x -k "bs64-default or bs8-unfused or fused_batchnorm or bf16_train_step or conditioned or reduces_the_loss or two_runs or bit" > $O/e2e.log 2>&1; echo "rc $?" >> $O/e2e.log; grep -E "passed|failed|^FAILED|^rc|^E  " $O/e2e.log | tail -8
