#!/bin/bash
# round 4: bucket tails (reduce / unpack / Ranger / re-pack) on a third stream -- same-box A/B, overlap trace, the tests that run whole steps
O=gpurun_out/r4_post
mkdir -p $O
R=$PWD
export PYTHONUNBUFFERED=1
b() { python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 6 2>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
for rep in 1 2 3; do
echo "post=0: $(GDRN_POST_STREAM=0 b)   post=1: $(GDRN_POST_STREAM=1 b)"
done 2>&1 | tee $O/ab.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace -o p -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-extras > $R/$O/trace.log 2>&1
cd $R
f=$(ls $O/trace/*/p_kernel_trace.csv $O/trace/p_kernel_trace.csv 2>/dev/null | head -1)
python tools/overlap_trace.py $f > $O/overlap.txt 2>&1
python tools/trace_steps.py $f 5 > $O/steps.txt 2>&1
rm -rf $O/trace
cat $O/overlap.txt | head -60
timeout 1200 python -m pytest tests/test_e2e_gpu.py tests/test_teacher_forced_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
tail -5 $O/tests.log
