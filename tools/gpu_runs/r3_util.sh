#!/bin/bash
# round 3: MFMA utilisation + HBM bandwidth per kernel (tools/pmc_util.py) -- SQ/GRBM pass, FETCH pass, WRITE pass, plain trace, calibration pass
O=$PWD/gpurun_out/r3_util
mkdir -p $O
R=$PWD
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-roofline --no-extras"
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq -o p -- $B > $O/sq.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o p -- $B > $O/fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o p -- $B > $O/write.log 2>&1
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/plain -o p -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-extras > $O/plain.log 2>&1
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/cal -o p -- python $R/tools/ubench/mfma_rate.py > $O/cal.log 2>&1
cd $R
f() { if ls $O/$1/*/p_$2.csv >/dev/null 2>&1; then ls $O/$1/*/p_$2.csv | head -1; else echo $O/$1/p_$2.csv; fi; }
python tools/pmc_util.py $(f sq counter_collection) $(f fetch counter_collection) $(f write counter_collection) $(f plain kernel_trace) $(f cal counter_collection) $O/r03_mfma_util_hbm_bs64_bf16 2>&1 | tail -40
python tools/trace_steps.py $(f plain kernel_trace) 4 > $O/steps.txt 2>&1
rm -rf $O/sq $O/fetch $O/write $O/plain $O/cal
