#!/bin/bash
# round 5: workgroups per launch of head_conv_tail64_kernel (512 / 1024 / 2048; libraries built with a temporary -DHCT_BLOCKS macro in the git-ignored _ab/).
# Measured: inference 1.851-1.852 / 1.850-1.852 / 1.867-1.877 ms, train 7.208-7.220 / 7.225-7.236 / 7.219-7.235 ms: 1024 stays (the macro was not kept).
O=gpurun_out/r5_hct
mkdir -p $O
export PYTHONUNBUFFERED=1
b() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 40 --warmup 8 "$@" 2>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
for rep in 1 2; do
for nb in 512 1024 2048; do
  if [ $nb = 1024 ]; then L=""; else L="GDRN_HIP_LIB=$PWD/_ab/libgdrn_hip_hct$nb.so"; fi
  echo "blocks $nb: inference $(env $L bash -c "$(declare -f b); O=$O; b --fwd-only")  train $(env $L bash -c "$(declare -f b); O=$O; b")"
done; done | tee $O/ab.txt
