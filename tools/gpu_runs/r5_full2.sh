#!/bin/bash
# round 5: every gpu test (no -x: the whole list of failures), smoke, the parity-mode seeds at bs 64 under both kernel policies
O=gpurun_out/r5_full2
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1800 python -m pytest tests -q -m gpu > $O/gputests.log 2>&1; echo "rc $?" >> $O/gputests.log; grep -E "passed|failed|^FAILED|^rc|^ERROR" $O/gputests.log | tail -12
timeout 900 python -m pytest tests/test_e2e_gpu.py -q -m gpu -s -k "seeds_at_bs64 or bf16_train_step_vs_reference" 2>&1 | grep -E "fp32 \[|bf16 |passed|failed" | tee $O/seeds.txt
timeout 600 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc $?"; grep "smoke\|Error" $O/smoke.log
