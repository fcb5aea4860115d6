#!/bin/bash
O=gpurun_out/r6_ablate
mkdir -p $O
export PYTHONUNBUFFERED=1 GDRN_DEFER_HEAD=0
b() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 8 "$@" 2>>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
echo "base:            $(b) $(b) $(b)" | tee $O/ab.txt
echo "ablate rows<=128: $(GDRN_ABLATE_BN=1 b) $(GDRN_ABLATE_BN=1 b) $(GDRN_ABLATE_BN=1 b)" | tee -a $O/ab.txt
echo "ablate rows<=512: $(GDRN_ABLATE_BN=1 GDRN_ABLATE_ROWS=512 b) $(GDRN_ABLATE_BN=1 GDRN_ABLATE_ROWS=512 b)" | tee -a $O/ab.txt
echo "ablate all:      $(GDRN_ABLATE_BN=1 GDRN_ABLATE_ROWS=100000 b) $(GDRN_ABLATE_BN=1 GDRN_ABLATE_ROWS=100000 b)" | tee -a $O/ab.txt
echo "base:            $(b) $(b)" | tee -a $O/ab.txt
tail -3 $O/err.log
