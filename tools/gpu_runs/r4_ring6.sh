#!/bin/bash
# round 4: six-stage weight ring of the 8 x 8-pixel halo tile (layer4) against the committed three-stage one (_ab/ = libraries built from HEAD)
# (the kernel change it measured -- RING = 6 in conv3x3_halo.hip -- was dropped: DESIGN.md section 4 (e); _ab/ held libraries built from HEAD)
O=gpurun_out/r4_ring6
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_kernels_fp16_gpu.py -q -m gpu -x -k "halo or conv3x3 or fused or xf or bnb" > $O/ktests.log 2>&1; echo "rc $?" >> $O/ktests.log; grep -E "passed|failed|^FAILED|^rc" $O/ktests.log | tail -5
b() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 6 "$@" 2>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
base() { GDRN_HIP_LIB=$PWD/_ab/libgdrn_hip_base.so GDRN_HIP_LIB_F16=$PWD/_ab/libgdrn_hip_f16_base.so b "$@"; }
for rep in 1 2 3; do echo "train base: $(base)  ring6: $(b)   inference base: $(base --fwd-only)  ring6: $(b --fwd-only)"; done | tee $O/ab.txt
