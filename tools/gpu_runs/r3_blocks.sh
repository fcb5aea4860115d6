#!/bin/bash
# workgroups per grouped weight-gradient launch: fewer = fewer partial tiles (HBM traffic, workspace), coarser balance
O=$PWD/gpurun_out/r3_blocks.txt
: > $O
B="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 8"
run() { r=$(env $1 timeout 300 $B $2 2>/dev/null | grep '^{"metric' | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"); echo "$2 $1 -> $r ms/step" >> $O; }
for rep in 1 2; do
for n in 1536 1280 1024 768 512; do run "GDRN_WGRAD_BLOCKS=$n" ""; done
done
cat $O
