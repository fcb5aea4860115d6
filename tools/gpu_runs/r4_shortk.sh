#!/bin/bash
# round 4: 64-row tile of the generic kernel for short reductions (GDRN_GEMM_SHORTK = reduction length below which it applies)
# (the switch it measured -- GDRN_GEMM_SHORTK in gdrn_conv_tile -- was dropped: DESIGN.md section 4 (e))
O=gpurun_out/r4_shortk
mkdir -p $O
export PYTHONUNBUFFERED=1
b() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 6 "$@" 2>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
for rep in 1 2; do echo "train: off $(b)  K<200 $(GDRN_GEMM_SHORTK=200 b)  K<300 $(GDRN_GEMM_SHORTK=300 b)  K<600 $(GDRN_GEMM_SHORTK=600 b)  K<1200 $(GDRN_GEMM_SHORTK=1200 b)  all $(GDRN_GEMM_SHORTK=100000 b)   inference: off $(b --fwd-only)  K<300 $(GDRN_GEMM_SHORTK=300 b --fwd-only) all $(GDRN_GEMM_SHORTK=100000 b --fwd-only)"; done | tee $O/ab.txt
