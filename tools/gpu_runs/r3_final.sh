#!/bin/bash
# every gpu test + smoke + the data-parallel A/B + bucket timeline of the data-parallel step
O=$PWD/gpurun_out/r3_final
mkdir -p $O
R=$PWD
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gputests.log 2>&1; grep -E "passed|failed|error" $O/gputests.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; grep smoke $O/smoke.log
B="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 8"
run() { r=$(env $1 timeout 300 $B $2 2>/dev/null | grep '^{"metric' | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"); echo "$2 $1 -> $r ms/step" | tee -a $O/ab.txt; }
run "X=1" ""
run "X=1" "--dist-force"
run "GDRN_EARLY_OPT=0" "--dist-force"
run "GDRN_SIDE_PRIO=normal GDRN_RED_PRIO=normal GPU_MAX_HW_QUEUES=4" "--dist-force"
run "GDRN_COMM_DTYPE=bf16" "--dist-force"
cd /tmp && export TMPDIR=/tmp
GDRN_BUCKETS=5 timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/t -o p -- python $R/bench.py --no-cpu-baseline --no-roofline --no-extras --steps 8 --warmup 3 --dist-force > $O/trace.log 2>&1
cd $R
f=$(ls $O/t/*/p_kernel_trace.csv $O/t/p_kernel_trace.csv 2>/dev/null | head -1)
python tools/bucket_timeline.py $f > $O/r03_bucket_timeline_bs64_bf16.txt 2>&1
rm -rf $O/t
cat $O/r03_bucket_timeline_bs64_bf16.txt
