#!/bin/bash
# round 6: what bounds the step?  ablation probes (results are garbage, timings are not): side-stream work removed
O=gpurun_out/r6_ablate2
mkdir -p $O
export PYTHONUNBUFFERED=1
b() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 8 "$@" 2>>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
echo "base:                      $(b) $(b)" | tee $O/ab.txt
echo "no grouped wgrad:          $(GDRN_ABLATE_SIDE=wgrad b) $(GDRN_ABLATE_SIDE=wgrad b)" | tee -a $O/ab.txt
echo "no side ops (but bucket marks):  $(GDRN_ABLATE_SIDE=all b) $(GDRN_ABLATE_SIDE=all b)" | tee -a $O/ab.txt
echo "no side ops, no early opt: $(GDRN_ABLATE_SIDE=all GDRN_EARLY_OPT=0 b) $(GDRN_ABLATE_SIDE=all GDRN_EARLY_OPT=0 b)" | tee -a $O/ab.txt
echo "one stream:                $(GDRN_WGRAD_STREAM=0 b) $(GDRN_WGRAD_STREAM=0 b)" | tee -a $O/ab.txt
echo "one stream no wgrad:       $(GDRN_WGRAD_STREAM=0 GDRN_ABLATE_SIDE=wgrad b) $(GDRN_WGRAD_STREAM=0 GDRN_ABLATE_SIDE=wgrad b)" | tee -a $O/ab.txt
echo "fwd only train (noopt):    " | tee -a $O/ab.txt
tail -3 $O/err.log
