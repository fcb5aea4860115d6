#!/bin/bash
# round 3 A/B (same box): single stream | side-stream bucket ends | + per-bucket optimizer updates
for rep in 1 2; do
for cfg in "GDRN_WGRAD_STREAM=0" "GDRN_WGRAD_STREAM=1 GDRN_EARLY_OPT=0 GDRN_BUCKETS=4" "GDRN_WGRAD_STREAM=1 GDRN_EARLY_OPT=0" "GDRN_WGRAD_STREAM=1 GDRN_EARLY_OPT=1"; do
  r=$(env $cfg python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 40 --warmup 8 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])")
  echo "$cfg -> $r ms/step"
done
done
