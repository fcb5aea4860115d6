#!/bin/bash
# round 4: partial-tile reduction with eight loads in flight per thread instead of four (same sums, same order) against the previous build (_ab/)
# (no difference: 7.339-7.375 vs 7.354-7.359 ms; the change was dropped)
O=gpurun_out/r4_red8
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "wgrad" > $O/ktests.log 2>&1; echo "rc $?" >> $O/ktests.log; grep -E "passed|failed|^FAILED|^rc" $O/ktests.log | tail -4
b() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 6 "$@" 2>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
base() { GDRN_HIP_LIB=$PWD/_ab/libgdrn_hip_base.so b "$@"; }
for rep in 1 2 3; do echo "train: four in flight $(base)   eight $(b)"; done | tee $O/ab.txt
