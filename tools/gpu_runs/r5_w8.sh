#!/bin/bash
# round 5: the eight-wave (K-split) form of the first halo kernel's 128-channel tile -- kernel tests, isolated timings 4 vs 8 waves, step / inference A/B
O=gpurun_out/r5_w8
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "halo" > $O/ktests.log 2>&1; echo "rc $?" >> $O/ktests.log; grep -E "passed|failed|^FAILED|^rc|Error" $O/ktests.log | tail -8
for wv in 4 8; do echo "== isolated, halo_waves $wv"; HALO_WAVES=$wv timeout 300 python tools/halotime.py 0,1,3 2>&1 | grep -v amdgpu.ids; done | tee $O/halotime.txt
b() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 6 "$@" 2>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
for rep in 1 2 3; do echo "train w4: $(GDRN_HALO_WAVES=4 b)  auto: $(b)   inference w4: $(GDRN_HALO_WAVES=4 b --fwd-only)  auto: $(b --fwd-only)"; done | tee $O/ab.txt
echo "one stream: w4 $(GDRN_WGRAD_STREAM=0 GDRN_HALO_WAVES=4 b) auto $(GDRN_WGRAD_STREAM=0 b)" | tee -a $O/ab.txt
