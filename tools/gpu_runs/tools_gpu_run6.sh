#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
R=$PWD
rm -f gpurun_out/t6.log
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "test_conv3x3_halo" --timeout=300 -p no:cacheprovider 2>&1 | tail -30 >> gpurun_out/t6.log
timeout 1200 python -m pytest tests/test_e2e_gpu.py -m gpu -q -s -k "fp32_train_step or bf16_train or vs_oracle" --timeout=600 -p no:cacheprovider 2>&1 | tail -30 >> gpurun_out/t6.log
GDRN_LAYER_TABLE=$R/gpurun_out/layers6.txt timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench6.log 2>&1
GDRN_HALO=0 timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/bench6_nohalo.log 2>&1
grep -E "passed|failed|error|rel-err" gpurun_out/t6.log | tail
tail -1 gpurun_out/bench6.log | cut -c1-1100
tail -1 gpurun_out/bench6_nohalo.log | cut -c1-250
head -24 gpurun_out/layers6.txt
