#!/bin/bash
# round 6: the fp16 mode's dynamic loss scale decided on the device: tests, then the fp16 step against bf16 on one box
O=gpurun_out/r6_fp16
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_fp16_gpu.py -q -m gpu -x > $O/tests.log 2>&1; echo "rc $?" >> $O/tests.log; grep -E "passed|failed|^FAILED|^rc|^E  " $O/tests.log | tail -8
b() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 6 "$@" 2>>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
echo "bf16:                 $(b) $(b)" | tee $O/ab.txt
echo "fp16 dynamic (device): $(b --dtype fp16) $(b --dtype fp16) $(b --dtype fp16)" | tee -a $O/ab.txt
echo "fp16 static:          $(GDRN_LOSS_SCALE=1024:static b --dtype fp16) $(GDRN_LOSS_SCALE=1024:static b --dtype fp16)" | tee -a $O/ab.txt
tail -3 $O/err.log
