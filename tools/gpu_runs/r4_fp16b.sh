#!/bin/bash
O=gpurun_out/r4_fp16
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_fp16_gpu.py -q -m gpu -s > $O/e2e.log 2>&1; echo "rc $?" >> $O/e2e.log; grep -v "^Randomly" $O/e2e.log | tail -25
timeout 900 python -m pytest tests/test_teacher_forced_gpu.py -q -m gpu -s -k "fp16" > $O/tf.log 2>&1; echo "rc $?" >> $O/tf.log; grep -v "^Randomly" $O/tf.log | tail -40
