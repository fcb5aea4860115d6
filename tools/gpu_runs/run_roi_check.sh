#!/bin/bash
# N3 RoI cropper: parity tests + timing.  Usage: gpurun --timeout 600 -- 'bash tools/gpu_runs/run_roi_check.sh'
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_roi_gpu.py -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/roi_tests.log
timeout 120 python tools/roibench.py 64 2>&1 | tail -5 | tee gpurun_out/roi_bench.log
