#!/bin/bash
# round 4: generic kernel, fused BatchNorm-backward epilogue with the loads of half the fragment rows in flight together: tests, one-stream kernel table, step
# (the change it measured -- epilogue loads of conv_gemm.hip batched per half tile -- made no difference and was dropped: DESIGN.md section 4)
O=$PWD/gpurun_out/r4_gemm_epi
mkdir -p $O
R=$PWD
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_kernels_fp16_gpu.py -q -m gpu -k "gemm or fused or convT or stride" > $O/ktests.log 2>&1; echo "rc $?" >> $O/ktests.log; grep -E "passed|failed|^FAILED|^rc" $O/ktests.log | tail -5
timeout 1500 python -m pytest tests/test_e2e_gpu.py tests/test_teacher_forced_gpu.py -q -m gpu -x > $O/tests.log 2>&1; echo "rc $?" >> $O/tests.log; grep -E "passed|failed|^FAILED|^rc" $O/tests.log | tail -5
cd /tmp && export TMPDIR=/tmp
GDRN_WGRAD_STREAM=0 GDRN_WGRAD_BLOCKS=768 GDRN_WGRAD_FORCE_LDS=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -o p -- python $R/bench.py --no-cpu-baseline --no-roofline --no-extras --steps 8 --warmup 3 > $O/serial.log 2>&1
cd $R
f=$(ls $O/serial/*/p_kernel_trace.csv 2>/dev/null | head -1); [ -z "$f" ] && f=$O/serial/p_kernel_trace.csv
python tools/trace_steps.py $f 5 "one stream" 2>&1 | grep -E "steady|conv_gemm" | cut -c1-200 | tee $O/kernels.txt
rm -rf $O/serial
b() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 6 "$@" 2>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
for rep in 1 2 3; do echo "train: $(b)"; done | tee $O/ab.txt
