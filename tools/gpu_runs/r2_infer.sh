#!/bin/bash
# kernel-time table of the eval-mode forward (bs=64 bf16)
mkdir -p gpurun_out/r2_infer
O=$PWD/gpurun_out/r2_infer
R=$PWD
timeout 300 python tools/inferbench.py 50 2>/dev/null | tail -1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python $R/tools/inferbench.py 20 > $O/stats.log 2>&1
cd $R
python tools/summarize_stats.py $O/stats/p_kernel_stats.csv 25 "rocprofv3 --kernel-trace --stats -- python tools/inferbench.py 20 (eval-mode forward, bs=64 bf16, 25 profiled forwards)" > $O/kernel_stats.txt
head -30 $O/kernel_stats.txt | cut -c1-150
