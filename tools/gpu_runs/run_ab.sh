#!/bin/bash
# tests, then bench with an A/B environment switch:  run_ab.sh VAR
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for v in 0 1; do
echo "$1=$v"; env $1=$v timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-200
done
