#!/bin/bash
# round 6: this build against gdr-net_amd/lib/libgdrn_hip_prev.so (tools/build_prev.sh) on one box, interleaved: step and inference; optional kernel tests first (KT="-k expr")
O=$PWD/gpurun_out/r6_ablib
mkdir -p $O
export PYTHONUNBUFFERED=1
if [ -n "$KT" ]; then timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_kernels_fp16_gpu.py -q -m gpu -x -k "$KT" > $O/ktests.log 2>&1; echo "rc $?" >> $O/ktests.log; grep -E "passed|failed|^FAILED|^rc|^E  " $O/ktests.log | tail -6; fi
PREV=$PWD/gdr-net_amd/lib/libgdrn_hip_prev.so
b() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 6 "$@" 2>>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
i() { timeout 300 python bench.py --fwd-only --no-cpu-baseline --no-roofline --no-extras --steps 100 --warmup 10 "$@" 2>>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
{
for r in 1 2 3; do
echo "this build: train $(b)  inference $(i)      previous library: train $(GDRN_HIP_LIB=$PREV b)  inference $(GDRN_HIP_LIB=$PREV i)"
done
} | tee $O/ab.txt
