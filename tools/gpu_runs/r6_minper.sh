#!/bin/bash
# round 6: fewest k-steps per workgroup of the grouped weight gradient (only the Patch-PnP bucket sits at the minimum: 672 workgroups of 16 k-steps,
# 128 pixel-range splits on its first conv -> a 99 MB reduction by 192 workgroups, 254 us in the step under the head's first data gradient)
O=$PWD/gpurun_out/r6_minper
mkdir -p $O
export PYTHONUNBUFFERED=1
b() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 6 "$@" 2>>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
{
for r in 1 2; do
for m in 16 32 64 128 256; do echo "min k-steps $m: $(GDRN_WGRAD_MIN_PER=$m b)"; done
done
} | tee $O/ab.txt
python tools/wgrad_shape_probe.py 2>&1 | tee $O/shape_probe.txt
