#!/bin/bash
# round 4: launch the LAST bucket's weight gradients in pieces (GDRN_WGRAD_CUTS: backward-group indices in forward order: stem 0, layer1 1-3, layer2 4-7)
O=gpurun_out/r4_cuts
mkdir -p $O
export PYTHONUNBUFFERED=1
b() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 6 "$@" 2>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
for rep in 1 2; do
for c in "" "4" "2" "4,2" "6,4,2" "8,4" "1"; do echo "cuts='$c': $(GDRN_WGRAD_CUTS=$c b)"; done
echo "tail_overlap=0: $(GDRN_TAIL_OVERLAP=0 b)   blocks 640: $(GDRN_WGRAD_BLOCKS=640 b)   blocks 1024: $(GDRN_WGRAD_BLOCKS=1024 b)   w128 last bucket only: $(GDRN_WGRAD_W128=1 GDRN_W128_ONLY_LAST=1 b)"
done | tee $O/ab.txt
