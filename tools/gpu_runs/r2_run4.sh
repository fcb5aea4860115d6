#!/bin/bash
mkdir -p gpurun_out/r2_4
timeout 600 python tools/xf_diff.py > gpurun_out/r2_4/xf_diff.log 2>&1
grep "##\|==" gpurun_out/r2_4/xf_diff.log
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r2_4/pytest_all.log 2>&1
tail -12 gpurun_out/r2_4/pytest_all.log
for b in 4 5; do GDRN_BUCKETS=$b timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('buckets $b', d['ms_per_step'])"; done
