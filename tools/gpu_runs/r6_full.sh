#!/bin/bash
# round 6: every gpu test on the current tree (what the driver runs at round end), smoke, then the default bench line
O=gpurun_out/r6_full
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 2400 python -m pytest tests -x -q -m gpu > $O/gputests.log 2>&1; echo "rc $?" >> $O/gputests.log; grep -E "passed|failed|^FAILED|^rc|^ERROR|^E  " $O/gputests.log | tail -8
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -4 $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; cut -c1-300 $O/bench.json
