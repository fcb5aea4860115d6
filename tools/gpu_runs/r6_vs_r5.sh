#!/bin/bash
# round 6 end against round 5 end ON ONE BOX: _r5/ = the tree of commit 232ee8b (git --work-tree=_r5 checkout 232ee8b -- . ; built in place; not tracked),
# interleaved runs of both trees' bench.py: bf16 training step and eval-mode inference at bs = 64
O=$PWD/gpurun_out/r6_vs_r5
mkdir -p $O
export PYTHONUNBUFFERED=1
b() { ( cd $1 && timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 6 2>>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])" ); }
i() { ( cd $1 && timeout 300 python bench.py --fwd-only --no-cpu-baseline --no-roofline --no-extras --steps 100 --warmup 10 2>>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])" ); }
{
for r in 1 2 3 4; do
echo "round 6 (this tree): train $(b .)  inference $(i .)      round 5 (commit 232ee8b): train $(b _r5)  inference $(i _r5)"
done
} | tee $O/ab.txt
