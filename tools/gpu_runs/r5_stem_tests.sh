#!/bin/bash
# round 5: the eval-mode tests behind the fused stem (r5_stem.sh had a wrong file name in its last line)
O=gpurun_out/r5_stem
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_e2e_gpu.py tests/test_fp16_gpu.py -q -m gpu -k "inference or eval or checkpoint or amp or g10 or conditioned or batch_sizes or reference" > $O/e2e.log 2>&1; echo "rc $?" >> $O/e2e.log; grep -E "passed|failed|^FAILED|^rc" $O/e2e.log | tail -6
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -o p -- python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 12 --warmup 3 --fwd-only > $O/tr.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r5_stem/tr/**/p_kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]: print(r["Name"][:70], r["Calls"], r["AverageNs"])
PY
rm -rf $O/tr
