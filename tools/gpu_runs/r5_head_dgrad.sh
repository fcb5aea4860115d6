#!/bin/bash
# round 5: the 1x1 output conv's data gradient on its own kernel (gdrn_head_out_dgrad) -- kernel test, step A/B against the generic kernel (plan.py
# patched in this scratch copy for the B leg), teacher-forced + e2e suites
O=gpurun_out/r5_head_dgrad
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "head_out_dgrad or head_conv_tail" > $O/ktests.log 2>&1; echo "rc $?" >> $O/ktests.log; grep -E "passed|failed|^FAILED|^rc|^E  " $O/ktests.log | tail -8
b() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 6 "$@" 2>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
echo "train, own dgrad kernel: $(b) $(b) $(b)" | tee $O/ab.txt
cp gdr-net_amd/plan.py $O/plan.py.bak
sed -i 's/            if e.h16 and e.gemm_bnb:$/            if False:/' gdr-net_amd/plan.py
echo "train, generic kernel:   $(b) $(b) $(b)" | tee -a $O/ab.txt
cp $O/plan.py.bak gdr-net_amd/plan.py; rm $O/plan.py.bak
timeout 1500 python -m pytest tests/test_teacher_forced_gpu.py tests/test_e2e_gpu.py -q -m gpu -x -k "bs64-default or bs8-unfused or bs8-no-gemm or fused_batchnorm or fp32_train_step or bf16_train_step or reduces_the_loss or vs_oracle_other" > $O/e2e.log 2>&1; echo "rc $?" >> $O/e2e.log; grep -E "passed|failed|^FAILED|^rc" $O/e2e.log | tail -6
