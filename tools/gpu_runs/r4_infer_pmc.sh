#!/bin/bash
# round 4: MFMA utilisation / HBM traffic per kernel of the eval-mode forward (bs = 64, bf16), the counterpart of r04_mfma_util_hbm_bs64_bf16 for
# inference: three rocprofv3 --pmc passes (each with --kernel-trace only), a plain kernel trace for the durations, the pure-MFMA calibration loop
O=$PWD/gpurun_out/r4_infer_pmc
mkdir -p $O
R=$PWD
export PYTHONUNBUFFERED=1
f() { if ls $O/$1/*/p_$2.csv >/dev/null 2>&1; then ls $O/$1/*/p_$2.csv | head -1; else echo $O/$1/p_$2.csv; fi; }
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-roofline --no-extras --fwd-only"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/plain -o p -- $B --steps 12 --warmup 3 > $O/plain.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq -o p -- $B --steps 3 --warmup 2 > $O/sq.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o p -- $B --steps 3 --warmup 2 > $O/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o p -- $B --steps 3 --warmup 2 > $O/write.log 2>&1
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/cal -o p -- python $R/tools/ubench/mfma_rate.py > $O/cal.log 2>&1
cd $R
python tools/pmc_util.py $(f sq counter_collection) $(f fetch counter_collection) $(f write counter_collection) $(f plain kernel_trace) $(f cal counter_collection) $O/r04_mfma_util_hbm_inference_bs64_bf16 "bs=64 bf16 eval-mode forward (python bench.py --fwd-only)" 2>&1 | tail -30
rm -rf $O/plain $O/sq $O/fetch $O/write $O/cal
