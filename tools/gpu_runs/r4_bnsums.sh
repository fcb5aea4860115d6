#!/bin/bash
# NOTE: measures switches (GDRN_BN_TAIL / GDRN_BN_SUMS) of the BatchNorm-table experiments that lived in commits 012ec22..e111f10 and were
# removed again (all slower: profiles/r04_bn_statistics_variants.txt); kept as the record of what was run.  Check out e111f10 to re-run.
# round 4: BatchNorm statistics through fixed-point tables, coefficient vectors in the consumer conv's prologue (csrc/bn_sums.h)
O=gpurun_out/r4_bnsums
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_kernels_fp16_gpu.py -x -q -m gpu -k "fixed_point or halo or conv_gemm" > $O/kernels.log 2>&1; echo "rc $?" >> $O/kernels.log; tail -6 $O/kernels.log
b() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 6 "$@" 2>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
for rep in 1 2 3; do echo "sums=0: $(GDRN_BN_SUMS=0 b)   sums=7 (tables + finish launches): $(GDRN_BN_SUMS=7 b)   sums=1: $(GDRN_BN_SUMS=1 b)   sums=2: $(GDRN_BN_SUMS=2 b)   sums=3: $(GDRN_BN_SUMS=3 b)"; done | tee $O/ab.txt
timeout 1500 python -m pytest tests/test_teacher_forced_gpu.py tests/test_fp16_gpu.py -x -q -m gpu -s > $O/tf.log 2>&1; echo "rc $?" >> $O/tf.log; grep -v "^Randomly" $O/tf.log | tail -12
timeout 1500 python -m pytest tests/test_e2e_gpu.py -x -q -m gpu > $O/e2e.log 2>&1; echo "rc $?" >> $O/e2e.log; tail -5 $O/e2e.log
