#!/bin/bash
mkdir -p gpurun_out/r2_6
O=gpurun_out/r2_6
timeout 900 python -m pytest tests/test_e2e_gpu.py -m gpu -q -s -k "conditioned_network" > $O/pytest_new.log 2>&1
grep -v "Randomly" $O/pytest_new.log | grep "bf16 train\|bf16 eval\|passed\|failed\|Error\|assert" | head -30
