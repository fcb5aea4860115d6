#!/bin/bash
# stream priority / hardware queue experiments: engine side stream and reducer stream at low priority, more hardware queues
O=$PWD/gpurun_out/r3_prio.txt
: > $O
B="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 8"
run() { r=$(env $1 timeout 300 $B $2 2>/dev/null | grep '^{"metric' | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"); echo "$2 $1 -> $r ms/step" >> $O; }
for rep in 1 2; do
run "X=1" ""
run "GDRN_SIDE_PRIO=low" ""
run "GDRN_SIDE_PRIO=low GDRN_WGRAD_SIDE_LDS=0" ""
run "GPU_MAX_HW_QUEUES=8" ""
run "X=1" "--dist-force"
run "GPU_MAX_HW_QUEUES=8" "--dist-force"
run "GDRN_SIDE_PRIO=low" "--dist-force"
run "GDRN_SIDE_PRIO=low GDRN_RED_PRIO=low" "--dist-force"
run "GDRN_SIDE_PRIO=low GPU_MAX_HW_QUEUES=8" "--dist-force"
run "GDRN_SIDE_PRIO=high" "--dist-force"
done
cat $O
