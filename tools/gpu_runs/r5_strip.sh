#!/bin/bash
# round 5: a STRIP form of the 64-channel halo tile (layer1: Cin = 64, one chunk per pixel): a workgroup walked up to four adjacent pixel tiles
# and fetched the patch of the next one under the current tile's taps (template argument MT of conv3x3_halo_kernel, gdrn_conv_params.halo_tiles,
# GDRN_HALO_FORM=t1 as the A/B switch).  Measured: nothing -- 64@64x64 34.1 vs 34.3 us plain, 44.3 vs 44.6 us with the BatchNorm-backward transform
# + copy-out, train step 7.54-7.57 vs 7.55-7.60 ms, inference 2.02 vs 2.02 ms (same box) -- so these launches do not wait for their patch
# loads; the change was dropped (DESIGN.md section 4, round-5 experiments); this script records what was run.
O=gpurun_out/r5_strip
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "halo" > $O/ktests.log 2>&1; echo "rc $?" >> $O/ktests.log; grep -E "passed|failed|^FAILED|^rc|Error" $O/ktests.log | tail -8
for tl in 1 0; do echo "== isolated, halo_tiles $tl"; HALO_TILES=$tl timeout 300 python tools/halotime.py 0,1,2,3,4 2>&1 | grep -E "C= 64"; done | tee $O/halotime.txt
b() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 6 "$@" 2>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
for rep in 1 2 3; do echo "train t1: $(GDRN_HALO_FORM=t1 b)  default: $(b)   inference t1: $(GDRN_HALO_FORM=t1 b --fwd-only)  default: $(b --fwd-only)"; done | tee $O/ab.txt
timeout 900 python -m pytest tests/test_teacher_forced_gpu.py tests/test_e2e_gpu.py -q -m gpu -x -k "bs64-default or fused_batchnorm_applies or fp32_train_step" > $O/e2e.log 2>&1; echo "rc $?" >> $O/e2e.log; grep -E "passed|failed|^FAILED|^rc" $O/e2e.log | tail -5
