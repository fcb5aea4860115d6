#!/bin/bash
mkdir -p gpurun_out/r2_7
O=gpurun_out/r2_7
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests -m gpu -q -s > $O/pytest_all.log 2>&1
grep -v "Randomly" $O/pytest_all.log | grep "passed\|failed\|FAILED\|bf16 \|fp32 " | tail -40
GDRN_LAYER_TABLE=$O/layers.txt timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
cat $O/bench.json; tail -3 $O/bench.err
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -4 $O/smoke.log
