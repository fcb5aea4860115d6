#!/bin/bash
# bench under several values of one environment switch:  run_env_sweep.sh VAR v1 v2 ...
export PYTHONUNBUFFERED=1
VAR=$1; shift
for v in "$@"; do
echo "$VAR=$v"; env $VAR=$v timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d.get('roofline',{}).get('kernel'), d.get('roofline',{}).get('achieved'))"
done
