#!/bin/bash
# data-parallel step on one GPU (one-rank RCCL group, force): per-bucket optimizer behind the bucket's all-reduce vs one optimizer launch set at the end
O=$PWD/gpurun_out/r3_dp.txt
: > $O
B="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 8 --dist-force"
for e in "GDRN_EARLY_OPT=0" "GDRN_EARLY_OPT=1" "GDRN_EARLY_OPT=0" "GDRN_EARLY_OPT=1" "GDRN_EARLY_OPT=1 GDRN_COMM_DTYPE=bf16"; do
  r=$(env $e timeout 300 $B 2>/dev/null | grep '^{"metric' | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])")
  echo "--dist-force (5 buckets) $e -> $r ms/step" >> $O
done
cat $O
