#!/bin/bash
# round 6, session 1: (a) the pipelined head bucket (GDRN_DEFER_HEAD) -- tests that walk the fused train step, same-box A/B of the step;
# (b) the CU-mask probe; (c) the new G11 test
O=gpurun_out/r6_defer
mkdir -p $O
export PYTHONUNBUFFERED=1
b() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 6 "$@" 2>>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
echo "defer=1: $(b) $(b) $(b)" | tee $O/ab.txt
echo "defer=0: $(GDRN_DEFER_HEAD=0 b) $(GDRN_DEFER_HEAD=0 b) $(GDRN_DEFER_HEAD=0 b)" | tee -a $O/ab.txt
echo "defer=1: $(b) $(b)" | tee -a $O/ab.txt
for side in mask:3 mask:2,3 mask:1,2,3; do
  for main in - 0 0,1 0,1,2; do
    GDRN_SIDE_STREAM=$side timeout 200 python tools/cumask_probe.py $main 30 2>>$O/err.log | tee -a $O/cumask.txt
  done
done
GDRN_DEFER_HEAD=0 GDRN_SIDE_STREAM=mask:2,3 timeout 200 python tools/cumask_probe.py 0,1 30 2>>$O/err.log | tee -a $O/cumask.txt
GDRN_DEFER_HEAD=0 timeout 200 python tools/cumask_probe.py - 30 2>>$O/err.log | tee -a $O/cumask.txt
timeout 200 python tools/cumask_probe.py - 30 2>>$O/err.log | tee -a $O/cumask.txt
timeout 1500 python -m pytest tests/test_e2e_gpu.py -q -m gpu -x -k "g11 or train_step or bucket or optimizer or reduces_the_loss or two_rank or graph or stream_of_changing" > $O/e2e.log 2>&1; echo "rc $?" >> $O/e2e.log; grep -E "passed|failed|^FAILED|^rc|^E  " $O/e2e.log | tail -12
tail -5 $O/err.log
