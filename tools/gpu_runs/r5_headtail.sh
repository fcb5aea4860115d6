#!/bin/bash
# round 5: inference: the head's 1x1 output conv + head tail as one kernel (gdrn_head_conv_tail_fwd) -- kernel test, inference A/B against the
# two-launch path (plan.py patched in this scratch copy for the B leg), the eval-mode tests
O=gpurun_out/r5_headtail
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "head_conv_tail or stem_conv_pool or head_tail" > $O/ktests.log 2>&1; echo "rc $?" >> $O/ktests.log; grep -E "passed|failed|^FAILED|^rc|^E  " $O/ktests.log | tail -8
b() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 40 --warmup 8 --fwd-only "$@" 2>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
echo "inference, fused head tail: $(b) $(b) $(b)   bs 8: $(b --bs 8)" | tee $O/ab.txt
cp gdr-net_amd/plan.py $O/plan.py.bak
sed -i 's/fused_tail = e.h16 and not S and not WL and nreg == 64/fused_tail = False/' gdr-net_amd/plan.py
echo "inference, two launches:    $(b) $(b) $(b)   bs 8: $(b --bs 8)" | tee -a $O/ab.txt
cp $O/plan.py.bak gdr-net_amd/plan.py; rm $O/plan.py.bak
timeout 1500 python -m pytest tests/test_e2e_gpu.py tests/test_fp16_gpu.py tests/test_roi_gpu.py -q -m gpu -k "inference or eval or checkpoint or amp or g10 or conditioned or batch_sizes or reference or postproc or correspond" > $O/e2e.log 2>&1; echo "rc $?" >> $O/e2e.log; grep -E "passed|failed|^FAILED|^rc" $O/e2e.log | tail -6
