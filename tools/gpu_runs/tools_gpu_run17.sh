#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
R=$PWD
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
GDRN_LAYER_TABLE=gpurun_out/layers20.txt timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1
