#!/bin/bash
# per-launch cost of a dependent tiny kernel under HIP runtime settings (tools/ubench/launch_gap.py)
O=$PWD/gpurun_out/r3_gap.txt
: > $O
run() { env "$@" timeout 120 python tools/ubench/launch_gap.py 2>&1 | grep -v amdgpu.ids >> $O; }
run X=1
run AMD_OPT_FLUSH=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run HIP_FORCE_DEV_KERNARG=0
run HIP_FORCE_DEV_KERNARG=1
run ROC_SYSTEM_SCOPE_SIGNAL=0
run DEBUG_HIP_GRAPH_BATCH_SIZE=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=1000
run ROC_USE_FGS_KERNARG=0
run GPU_FLUSH_ON_EXECUTION=1
run ROC_SKIP_KERNEL_ARG_COPY=1
cat $O
