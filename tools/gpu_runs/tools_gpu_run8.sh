#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
R=$PWD
rm -f gpurun_out/t8.log
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "test_conv3x3_wgrad_halo or test_conv3x3_halo" --timeout=300 -p no:cacheprovider 2>&1 | tail -30 >> gpurun_out/t8.log
timeout 1200 python -m pytest tests/test_e2e_gpu.py -m gpu -q -s --timeout=600 -p no:cacheprovider 2>&1 | tail -30 >> gpurun_out/t8.log
GDRN_LAYER_TABLE=$R/gpurun_out/layers8.txt timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench8.log 2>&1
grep -E "passed|failed|error|rel-err" gpurun_out/t8.log | tail
tail -1 gpurun_out/bench8.log | cut -c1-1100
grep wgrad gpurun_out/layers8.txt | head -12
