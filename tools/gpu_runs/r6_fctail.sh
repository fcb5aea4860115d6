#!/bin/bash
# round 6: fc2 finish + fc_r | fc_t + pose decode as one launch; pose-loss rows + weighted losses in the finalize launch; dL/dfc from the pose kernel
# (seeded train step); inference outputs without copy launches.  A/B: GDRN_FC_TAIL=0
O=$PWD/gpurun_out/r6_fctail
mkdir -p $O
export PYTHONUNBUFFERED=1
b() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 6 "$@" 2>>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
i() { timeout 300 python bench.py --fwd-only --no-cpu-baseline --no-roofline --no-extras --steps 100 --warmup 10 "$@" 2>>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
{
for r in 1 2 3; do
echo "fused seam: train $(b)  inference $(i)      GDRN_FC_TAIL=0: train $(GDRN_FC_TAIL=0 b)  inference $(GDRN_FC_TAIL=0 i)"
done
} | tee $O/ab.txt
timeout 2400 python -m pytest tests/test_teacher_forced_gpu.py tests/test_e2e_gpu.py tests/test_fp16_gpu.py -q -m gpu -x > $O/e2e.log 2>&1; echo "rc $?" >> $O/e2e.log; grep -E "passed|failed|^FAILED|^rc|^E  " $O/e2e.log | tail -8
