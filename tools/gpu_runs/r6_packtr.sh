#!/bin/bash
# round 6: tiled-transpose path of gdrn_pack_multi / gdrn_unpack_multi (fc1's operand copies and gradient unpack, the 1x1 / fc data-gradient operands): tests, A/B
O=$PWD/gpurun_out/r6_packtr
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_kernels_fp16_gpu.py -q -m gpu -x -k "pack or unpack" > $O/ktests.log 2>&1; echo "rc $?" >> $O/ktests.log; grep -E "passed|failed|^FAILED|^rc|^E  " $O/ktests.log | tail -6
b() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 6 "$@" 2>>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
for r in 1 2 3; do echo "tiled transposes: $(b)    granule gather (GDRN_PACK_TR=0): $(GDRN_PACK_TR=0 b)"; done | tee $O/ab.txt
timeout 2400 python -m pytest tests/test_teacher_forced_gpu.py tests/test_e2e_gpu.py tests/test_fp16_gpu.py -q -m gpu -x > $O/e2e.log 2>&1; echo "rc $?" >> $O/e2e.log; grep -E "passed|failed|^FAILED|^rc|^E  " $O/e2e.log | tail -6
