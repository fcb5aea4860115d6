#!/bin/bash
# round 5: inference without the fp32 logits when the caller does not return the maps (cfg.TEST.USE_PNP False): eval tests + inference figure
O=gpurun_out/r5_wantmaps
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_e2e_gpu.py tests/test_fp16_gpu.py tests/test_roi_gpu.py -q -m gpu -k "inference or eval or checkpoint or amp or g10 or conditioned or batch_sizes or reference or postproc or correspond" > $O/e2e.log 2>&1; echo "rc $?" >> $O/e2e.log; grep -E "passed|failed|^FAILED|^rc" $O/e2e.log | tail -4
b() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 40 --warmup 8 --fwd-only "$@" 2>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
echo "inference (no maps requested): $(b) $(b) $(b)" | tee $O/ab.txt
sed -i 's/kctx\["want_maps"\] = bool(cfg.TEST.USE_PNP)/kctx["want_maps"] = True/' gdr-net_amd/GDRN.py
echo "inference (logits written):    $(b) $(b) $(b)" | tee -a $O/ab.txt
