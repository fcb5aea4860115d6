#!/bin/bash
# NOTE: measures switches (GDRN_BN_TAIL / GDRN_BN_SUMS) of the BatchNorm-table experiments that lived in commits 012ec22..e111f10 and were
# removed again (all slower: profiles/r04_bn_statistics_variants.txt); kept as the record of what was run.  Check out e111f10 to re-run.
O=gpurun_out/r4_sums_trace
mkdir -p $O
R=$PWD
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
for v in 0 7 3; do
GDRN_BN_SUMS=$v timeout 400 rocprofv3 --kernel-trace --output-format csv -d $R/$O/t$v -o p -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-extras > $R/$O/t$v.log 2>&1
f=$(ls $R/$O/t$v/*/p_kernel_trace.csv $R/$O/t$v/p_kernel_trace.csv 2>/dev/null | head -1)
python $R/tools/trace_steps.py $f 5 > $R/$O/steps_$v.txt 2>&1
GDRN_WGRAD_STREAM=0 GDRN_BN_SUMS=$v timeout 400 rocprofv3 --kernel-trace --output-format csv -d $R/$O/s$v -o p -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-extras > $R/$O/s$v.log 2>&1
f=$(ls $R/$O/s$v/*/p_kernel_trace.csv $R/$O/s$v/p_kernel_trace.csv 2>/dev/null | head -1)
python $R/tools/trace_steps.py $f 5 > $R/$O/serial_$v.txt 2>&1
rm -rf $R/$O/t$v $R/$O/s$v
done
head -3 $R/$O/steps_*.txt $R/$O/serial_*.txt
