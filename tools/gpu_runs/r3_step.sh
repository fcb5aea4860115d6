#!/bin/bash
# round 3: whole-step check of a build -- bench line (no CPU baseline), steady-state kernel table from a rocprofv3 kernel trace, per-launch table
O=gpurun_out/r3_step
mkdir -p $O
R=$PWD
export PYTHONUNBUFFERED=1
GDRN_LAYER_TABLE=$R/$O/layers.txt timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace -o p -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-extras > $R/$O/trace.log 2>&1
cd $R
python tools/trace_steps.py $O/trace/*/p_kernel_trace.csv 5 > $O/steps.txt 2>&1 || python tools/trace_steps.py $O/trace/p_kernel_trace.csv 5 > $O/steps.txt 2>&1
rm -rf $O/trace
python - <<'PY'
import json
j = json.loads(open('gpurun_out/r3_step/bench.json').read().strip().splitlines()[-1])
print({k: j[k] for k in ("value", "ms_per_step")}, j.get("also", {}).get("inference_fwd_ms"), j.get("also", {}).get("inference_fwd_tflops"), j["roofline"]["kernel"], j["roofline"]["frac"])
for r in j["roofline"]["conv_kernels"]: print(r)
PY
head -70 $O/steps.txt
