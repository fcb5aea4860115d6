#!/bin/bash
# round 6: the PMC half of the profile set alone (MFMA utilisation / HBM traffic per kernel, train step + inference), with its errors visible
O=$PWD/gpurun_out/r6_prof
mkdir -p $O
R=$PWD
export PYTHONUNBUFFERED=1
f() { if ls $O/$1/*/p_$2.csv >/dev/null 2>&1; then ls $O/$1/*/p_$2.csv | head -1; else echo $O/$1/p_$2.csv; fi; }
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-roofline --no-extras"
B2="$B --steps 2 --warmup 2"
GDRN_WGRAD_STREAM=serial timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -o p -- $B --steps 8 --warmup 3 > $O/serial.log 2>&1
GDRN_WGRAD_STREAM=serial timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq -o p -- $B2 > $O/sq.log 2>&1
GDRN_WGRAD_STREAM=serial timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o p -- $B2 > $O/fetch.log 2>&1
GDRN_WGRAD_STREAM=serial timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o p -- $B2 > $O/write.log 2>&1
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/cal -o p -- python $R/tools/ubench/mfma_rate.py > $O/cal.log 2>&1
cd $R
ls $O/sq $O/fetch $O/write $O/cal $O/serial 2>&1 | head -30
python tools/pmc_util.py $(f sq counter_collection) $(f fetch counter_collection) $(f write counter_collection) $(f serial kernel_trace) $(f cal counter_collection) $O/r06_mfma_util_hbm_bs64_bf16 2>&1 | tail -15
tail -3 $O/sq.log
cd /tmp
BI="python $R/bench.py --no-cpu-baseline --no-roofline --no-extras --fwd-only"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/iplain -o p -- $BI --steps 12 --warmup 3 > $O/iplain.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/isq -o p -- $BI --steps 3 --warmup 2 > $O/isq.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/ifetch -o p -- $BI --steps 3 --warmup 2 > $O/ifetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/iwrite -o p -- $BI --steps 3 --warmup 2 > $O/iwrite.log 2>&1
cd $R
python tools/pmc_util.py $(f isq counter_collection) $(f ifetch counter_collection) $(f iwrite counter_collection) $(f iplain kernel_trace) $(f cal counter_collection) $O/r06_mfma_util_hbm_inference_bs64_bf16 "bs=64 bf16 eval-mode forward (python bench.py --fwd-only)" 2>&1 | tail -5
rm -rf $O/serial $O/sq $O/fetch $O/write $O/cal $O/iplain $O/isq $O/ifetch $O/iwrite
ls $O | grep mfma_util
