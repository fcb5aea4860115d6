#!/bin/bash
# round 4: why is the train variant of the head tail forward 68 us against 31 us for the eval variant?  debug build (_ab/, not in the tree):
# GDRN_HT_DBG=2 no atomics (block 0 stores), 4 = per-block partial rows instead of atomics; GDRN_HT_BLOCKS = grid cap
O=gpurun_out/r4_ht
mkdir -p $O
export GDRN_HIP_LIB=$PWD/_ab/libgdrn_hip_htdbg.so
for cfg in "0 4096" "2 4096" "4 4096" "0 2048" "0 1024" "0 512" "4 2048" "4 1024"; do set -- $cfg; GDRN_HT_DBG=$1 GDRN_HT_BLOCKS=$2 python tools/htbench.py; done 2>&1 | grep -v Warning | tee $O/ht.txt
