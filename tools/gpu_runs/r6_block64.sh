#!/bin/bash
# round 6: layer1's BasicBlocks in eval mode as one launch each (gdrn_block64_eval): kernel test, inference A/B, inference suites
O=gpurun_out/r6_block64
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_kernels_fp16_gpu.py -q -m gpu -x -k "block64" > $O/ktests.log 2>&1; echo "rc $?" >> $O/ktests.log; grep -E "passed|failed|^FAILED|^rc|^E  " $O/ktests.log | tail -6
i() { timeout 300 python bench.py --fwd-only --no-cpu-baseline --no-roofline --no-extras --steps 100 --warmup 10 "$@" 2>>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
echo "inference, fused layer1 blocks: $(i) $(i) $(i)" | tee $O/ab.txt
echo "inference, two launches:        $(GDRN_BLOCK64=0 i) $(GDRN_BLOCK64=0 i) $(GDRN_BLOCK64=0 i)" | tee -a $O/ab.txt
echo "inference, fused layer1 blocks: $(i) $(i)" | tee -a $O/ab.txt
timeout 1500 python -m pytest tests/test_e2e_gpu.py tests/test_fp16_gpu.py -q -m gpu -x -k "inference or eval or g10 or changing_batch or stem_conv_pool or head_conv_tail" > $O/e2e.log 2>&1; echo "rc $?" >> $O/e2e.log; grep -E "passed|failed|^FAILED|^rc|^E  " $O/e2e.log | tail -8
