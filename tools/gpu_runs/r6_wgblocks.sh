#!/bin/bash
# round 6: workgroup target of the grouped weight gradient PER BUCKET (pnp, head, layer4, layer3, rest); 768 everywhere is the shipped value.
# tools/wgrad_shape_probe.py: the layer4 + layer3 launch stand-alone is 25 % faster at ~128 k-steps per workgroup (1024 workgroups) than at ~190 (720).
O=$PWD/gpurun_out/r6_wgblocks
mkdir -p $O
export PYTHONUNBUFFERED=1
b() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 6 "$@" 2>>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
{
for r in 1 2; do
for m in "" "0,0,900" "0,0,960" "0,0,1024" "0,0,1100" "0,0,1150" "0,0,1200"; do echo "blocks '$m': $(GDRN_WGRAD_BLOCKS_B=$m b)"; done
done
} | tee $O/ab.txt
