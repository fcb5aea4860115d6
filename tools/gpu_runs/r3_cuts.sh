#!/bin/bash
# extra cuts of the grouped weight-gradient launches (GDRN_WGRAD_CUTS: backward group indices, forward order)
O=$PWD/gpurun_out/r3_cuts.txt
: > $O
timeout 600 python -m pytest tests/test_teacher_forced_gpu.py -x -q -k "bs64" > gpurun_out/r3_cuts_tf.log 2>&1; grep -E "passed|failed" gpurun_out/r3_cuts_tf.log | tail -1 >> $O
B="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 8"
run() { r=$(env "$1" timeout 300 $B $2 2>/dev/null | grep '^{"metric' | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"); echo "$2 $1 -> $r ms/step" >> $O; }
for rep in 1 2; do
run "GDRN_WGRAD_CUTS=" ""
run "GDRN_WGRAD_CUTS=22" ""
run "GDRN_WGRAD_CUTS=23,22" ""
run "GDRN_WGRAD_CUTS=22,20" ""
run "GDRN_WGRAD_CUTS=22,14" ""
run "GDRN_WGRAD_CUTS=22,11" ""
run "GDRN_WGRAD_CUTS=22,20,14,11,4" ""
done
run "GDRN_WGRAD_CUTS=" "--dist-force"
run "GDRN_WGRAD_CUTS=22" "--dist-force"
cat $O
