#!/bin/bash
# round 4 (last): workgroups per grouped weight-gradient launch after the chain got ~0.1 ms shorter
# result: 640: 7.51-7.54   704: 7.28-7.31   768 (default): 7.276-7.281   832: 7.44-7.46   896: 7.33 ms -- the default stays
O=gpurun_out/r4_blocks2
mkdir -p $O
b() { timeout 200 python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 6 "$@" 2>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
for rep in 1 2; do echo "blocks 640: $(GDRN_WGRAD_BLOCKS=640 b)  704: $(GDRN_WGRAD_BLOCKS=704 b)  768: $(b)  832: $(GDRN_WGRAD_BLOCKS=832 b)  896: $(GDRN_WGRAD_BLOCKS=896 b)"; done | tee $O/ab.txt
