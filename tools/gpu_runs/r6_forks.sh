#!/bin/bash
# round 6: small side-stream ops deferred to their bucket's end (11 -> 6 switches of the backward pass to the side stream), A/B on one box
O=$PWD/gpurun_out/r6_forks
mkdir -p $O
export PYTHONUNBUFFERED=1
b() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 6 "$@" 2>>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
{
for r in 1 2 3; do
echo "merged forks (6): $(b)    per-op forks (11): $(GDRN_MERGE_FORKS=0 b)"
done
} | tee $O/ab.txt
timeout 1800 python -m pytest tests/test_teacher_forced_gpu.py tests/test_e2e_gpu.py -q -m gpu -x -k "bs64 or fused_batchnorm or bucket or train_step" > $O/e2e.log 2>&1; echo "rc $?" >> $O/e2e.log; grep -E "passed|failed|^FAILED|^rc|^E  " $O/e2e.log | tail -8
