#!/bin/bash
# every gpu test + smoke + the driver-sized bench line (what the driver runs at round end)
O=$PWD/gpurun_out/r3_full
mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gputests.log 2>&1; grep -E "passed|failed|error" $O/gputests.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; grep smoke $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-400 $O/bench.json
