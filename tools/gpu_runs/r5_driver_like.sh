#!/bin/bash
# round 5: what the driver runs at round end, on the final tree: every gpu test, smoke, the default bench line
O=gpurun_out/r5_driver_like
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1800 python -m pytest tests -x -q -m gpu > $O/gputests.log 2>&1; echo "rc $?" >> $O/gputests.log; grep -E "passed|failed|^FAILED|^rc|^ERROR" $O/gputests.log | tail -5
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; cut -c1-400 $O/bench.json
