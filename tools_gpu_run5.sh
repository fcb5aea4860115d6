#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
R=$PWD
timeout 600 python tools_debug.py 2>&1 | grep -E "post|pre " 
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
rm -f gpurun_out/t5.log
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "test_conv_wgrad or test_conv_transpose or test_stem" --timeout=300 -p no:cacheprovider 2>&1 | tail -15 >> gpurun_out/t5.log
timeout 1200 python -m pytest tests/test_e2e_gpu.py -m gpu -q -s --timeout=600 -p no:cacheprovider 2>&1 | tail -40 >> gpurun_out/t5.log
GDRN_LAYER_TABLE=$R/gpurun_out/layers5.txt timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/bench5.log 2>&1
grep -E "passed|failed|error|rel-err" gpurun_out/t5.log | tail
tail -1 gpurun_out/bench5.log | cut -c1-1200
grep wgrad gpurun_out/layers5.txt | head -8
