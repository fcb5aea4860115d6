#!/bin/bash
# round 6: data gradient of the stride-2 3x3 convs (+ the shortcut's, + the BatchNorm-backward epilogue) on the parity-class kernel: kernel test, A/B, suites
O=gpurun_out/r6_s2d
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_kernels_fp16_gpu.py -q -m gpu -x -k "stride2" > $O/ktests.log 2>&1; echo "rc $?" >> $O/ktests.log; grep -E "passed|failed|^FAILED|^rc|^E  " $O/ktests.log | tail -8
b() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 6 "$@" 2>>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
echo "train, s2 kernels (fwd + dgrad): $(b) $(b) $(b)" | tee $O/ab.txt
echo "train, generic kernel:           $(GDRN_S2_HALO=0 b) $(GDRN_S2_HALO=0 b) $(GDRN_S2_HALO=0 b)" | tee -a $O/ab.txt
echo "train, s2 kernels (fwd + dgrad): $(b) $(b)" | tee -a $O/ab.txt
timeout 1800 python -m pytest tests/test_teacher_forced_gpu.py tests/test_e2e_gpu.py -q -m gpu -x -k "bs64-default or bs8-unfused or bs64-fp16 or fused_batchnorm or bf16_train_step or conditioned or reduces_the_loss or two_rank or graph" > $O/e2e.log 2>&1; echo "rc $?" >> $O/e2e.log; grep -E "passed|failed|^FAILED|^rc|^E  " $O/e2e.log | tail -8
