#!/bin/bash
# round 6: row form of the upsampling forward kernels (+ 32-bit index arithmetic in the fused adjoint): kernel tests, A/B against the previous build is
# the profile table; here the suites + step / inference timing
O=gpurun_out/r6_uprows
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_kernels_fp16_gpu.py -q -m gpu -x -k "upsample" > $O/ktests.log 2>&1; echo "rc $?" >> $O/ktests.log; grep -E "passed|failed|^FAILED|^rc|^E  " $O/ktests.log | tail -6
b() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 6 "$@" 2>>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
i() { timeout 300 python bench.py --fwd-only --no-cpu-baseline --no-roofline --no-extras --steps 100 --warmup 10 "$@" 2>>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
echo "train: $(b) $(b) $(b)   inference: $(i) $(i) $(i)" | tee $O/ab.txt
timeout 1800 python -m pytest tests/test_fp16_gpu.py tests/test_teacher_forced_gpu.py tests/test_e2e_gpu.py -q -m gpu -x -k "conditioned or bs64-default or bs64-fp16 or fused_batchnorm or inference or g10" > $O/e2e.log 2>&1; echo "rc $?" >> $O/e2e.log; grep -E "passed|failed|^FAILED|^rc|^E  " $O/e2e.log | tail -8
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -o p -- python /root/repo/bench.py --no-cpu-baseline --no-roofline --no-extras --steps 8 --warmup 3 > $O/tr.log 2>&1
grep -h -i "upsample" $O/tr/*/p_kernel_stats.csv $O/tr/p_kernel_stats.csv 2>/dev/null | cut -c1-200
rm -rf $O/tr
