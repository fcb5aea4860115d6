#!/bin/bash
# full GPU check of the current build: every gpu test, smoke, the driver's default bench line
mkdir -p gpurun_out/r2_full
O=gpurun_out/r2_full
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1
grep -E "passed|failed|FAILED|ERROR" $O/pytest_all.log | tail -15
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
GDRN_LAYER_TABLE=$O/layers.txt timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
cut -c1-1500 $O/bench.json; tail -2 $O/bench.err
