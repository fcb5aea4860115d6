#!/bin/bash
# round 6: the two-images-per-tile form of the stride-2 kernels (8-wide maps: layer4.0 + its shortcut, Patch-PnP's third conv, forward and backward)
# and the ConvTranspose's forward pass on the parity-class kernel: kernel tests, A/B against the previous build (GDRN_HIP_LIB), the suites
O=$PWD/gpurun_out/r6_tw8
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_kernels_fp16_gpu.py -q -m gpu -x -k "stride2 or conv_transpose" > $O/ktests.log 2>&1; echo "rc $?" >> $O/ktests.log; grep -E "passed|failed|^FAILED|^rc|^E  " $O/ktests.log | tail -8
b() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 6 "$@" 2>>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
i() { timeout 300 python bench.py --fwd-only --no-cpu-baseline --no-roofline --no-extras --steps 100 --warmup 10 "$@" 2>>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
{
for r in 1 2 3; do
echo "8-wide stride-2 kernels: train $(b)  inference $(i)      generic kernel there (GDRN_S2_TW8=0): train $(GDRN_S2_TW8=0 b)  inference $(GDRN_S2_TW8=0 i)"
done
} | tee $O/ab.txt
timeout 2400 python -m pytest tests/test_teacher_forced_gpu.py tests/test_e2e_gpu.py tests/test_fp16_gpu.py -q -m gpu -x > $O/e2e.log 2>&1; echo "rc $?" >> $O/e2e.log; grep -E "passed|failed|^FAILED|^rc|^E  " $O/e2e.log | tail -8
