#!/bin/bash
# round 5: every gpu test + smoke + the driver-sized bench line; A/B of the two round-5 defaults (eight-wave halo form in the forward pass,
# fp32 halo tile at bs >= 32)
O=gpurun_out/r5_full
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -q -m gpu -x > $O/gputests.log 2>&1; echo "rc $?" >> $O/gputests.log; grep -E "passed|failed|^FAILED|^rc|Error" $O/gputests.log | tail -8
timeout 600 python -m pytest tests/test_fp16_gpu.py tests/test_e2e_gpu.py -q -m gpu -s -k "g10 or bs64_on_the_halo or overflow" 2>&1 | grep -E "G10|fp32 \(halo|passed|failed|skipped" | tee $O/new_tests.txt
timeout 600 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc $?"; grep smoke $O/smoke.log
b() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 6 "$@" 2>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
for rep in 1 2 3; do echo "train w4: $(GDRN_HALO_WAVES=4 b)  default: $(b)   inference w4: $(GDRN_HALO_WAVES=4 b --fwd-only)  default: $(b --fwd-only)"; done | tee $O/ab.txt
echo "fp32 step: generic $(GDRN_HALO_F32=0 b --dtype fp32 --steps 8 --warmup 3)  auto $(b --dtype fp32 --steps 8 --warmup 3)" | tee -a $O/ab.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-1500 $O/bench.json
