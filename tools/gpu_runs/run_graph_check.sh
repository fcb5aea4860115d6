#!/bin/bash
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_e2e_gpu.py -m gpu -x -q -k "graph" 2>&1 | grep -E "^E  |passed|failed|Error" | head -20
GDRN_GRAPH=0 timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-200
GDRN_GRAPH=1 timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-200
