#!/bin/bash
# round 4: the whole GPU suite + smoke, as the driver runs them
O=gpurun_out/r4_full
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 2400 python -m pytest tests -x -q -m gpu > $O/gputest.log 2>&1; echo "rc $?" >> $O/gputest.log; tail -6 $O/gputest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc $?" >> $O/smoke.log; grep -v "^Randomly" $O/smoke.log | tail -8
