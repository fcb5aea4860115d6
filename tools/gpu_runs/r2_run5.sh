#!/bin/bash
mkdir -p gpurun_out/r2_5
O=gpurun_out/r2_5
timeout 900 python tools/bf16_parity_probe.py > $O/bf16_probe.log 2>&1
grep -v "Randomly" $O/bf16_probe.log | tail -40
timeout 900 python -m pytest tests/test_e2e_gpu.py -m gpu -q -s -k "baseline_sizes or over_seeds or golden_g4 or fused_batchnorm" > $O/pytest_new.log 2>&1
grep -v "Randomly" $O/pytest_new.log | tail -60
