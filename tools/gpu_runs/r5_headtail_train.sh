#!/bin/bash
# round 5: train mode: 1x1 output conv + head tail + map-loss sums as one kernel (gdrn_head_conv_tail_loss_fwd) -- kernel tests, step A/B against the
# two-launch path (plan.py patched in this scratch copy for the B leg), teacher-forced + e2e suites
O=gpurun_out/r5_headtail_train
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "head_conv_tail or head_tail" > $O/ktests.log 2>&1; echo "rc $?" >> $O/ktests.log; grep -E "passed|failed|^FAILED|^rc|^E  " $O/ktests.log | tail -8
b() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 6 "$@" 2>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
echo "train, fused head tail + losses: $(b) $(b) $(b)" | tee $O/ab.txt
cp gdr-net_amd/plan.py $O/plan.py.bak
sed -i 's/fused_tail = e.h16 and nreg == 64/fused_tail = e.h16 and nreg == 64 and not WL/' gdr-net_amd/plan.py
echo "train, two launches:             $(b) $(b) $(b)" | tee -a $O/ab.txt
cp $O/plan.py.bak gdr-net_amd/plan.py; rm $O/plan.py.bak
timeout 1500 python -m pytest tests/test_teacher_forced_gpu.py tests/test_e2e_gpu.py -q -m gpu -x -k "bs64-default or bs8-unfused or fused_batchnorm or fp32_train_step or bf16_train_step or conditioned or reduces_the_loss or vs_oracle_other" > $O/e2e.log 2>&1; echo "rc $?" >> $O/e2e.log; grep -E "passed|failed|^FAILED|^rc" $O/e2e.log | tail -6
