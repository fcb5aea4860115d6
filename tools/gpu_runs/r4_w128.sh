#!/bin/bash
# round 4, first contact of the 128 x 64 weight-gradient tile: kernel tests, isolated timings, whole-step A/B over the resident-grid size
O=gpurun_out/r4_w128
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "wgrad" > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
tail -5 $O/tests.log
timeout 300 python tools/wgrad_tiles.py > $O/tiles.txt 2>&1
cat $O/tiles.txt
b() { python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 6 2>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
for rep in 1 2; do
echo "base            : $(GDRN_WGRAD_W128=0 b)"
for g in 64 96 128 160 256; do
echo "w128=1 grid $g  : $(GDRN_WGRAD_W128=1 GDRN_W128_GRID=$g b)"
done
echo "w128=2 grid 128 : $(GDRN_WGRAD_W128=2 GDRN_W128_GRID=128 b)"
echo "w128=1 grid 128 blocks 512: $(GDRN_WGRAD_W128=1 GDRN_W128_GRID=128 GDRN_W128_BLOCKS=512 b)"
echo "w128=1 one stream: $(GDRN_WGRAD_W128=1 GDRN_WGRAD_STREAM=0 b)   base one stream: $(GDRN_WGRAD_W128=0 GDRN_WGRAD_STREAM=0 b)"
done 2>&1 | tee $O/ab.txt
