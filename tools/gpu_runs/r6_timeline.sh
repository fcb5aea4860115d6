#!/bin/bash
# round 6: kernel-by-kernel timeline of one training step and one inference forward (tools/step_timeline.py)
O=$PWD/gpurun_out/r6_timeline
mkdir -p $O
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o p -- python /root/repo/bench.py --no-cpu-baseline --no-roofline --no-extras --steps 8 --warmup 3 > $O/tr.log 2>&1
f=$(find $O/tr -name "p_kernel_trace.csv" | head -1)
python /root/repo/tools/step_timeline.py $f 2 > $O/step_timeline.txt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tri -o p -- python /root/repo/bench.py --fwd-only --no-cpu-baseline --no-roofline --no-extras --steps 8 --warmup 3 > $O/tri.log 2>&1
f=$(find $O/tri -name "p_kernel_trace.csv" | head -1)
python /root/repo/tools/step_timeline.py $f 2 > $O/inference_timeline.txt
rm -rf $O/tr $O/tri
head -3 $O/step_timeline.txt $O/inference_timeline.txt
