#!/bin/bash
# round 6: kernel tests of the stride-2 kernels, then per-layer times of their launches (layer table of the bench) and an A/B of step / inference
O=$PWD/gpurun_out/r6_tw8
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_kernels_fp16_gpu.py -q -m gpu -x -k "stride2 or conv_transpose" > $O/ktests.log 2>&1; echo "rc $?" >> $O/ktests.log; grep -E "passed|failed|^FAILED|^rc|^E  " $O/ktests.log | tail -8
GDRN_LAYER_TABLE=$O/layers_tw8.txt timeout 900 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_tw8.json 2> $O/bench.err
grep -h -E "s2|layer4.0|features.0" $O/layers_tw8.txt | cut -c1-150
b() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 6 "$@" 2>>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
i() { timeout 300 python bench.py --fwd-only --no-cpu-baseline --no-roofline --no-extras --steps 100 --warmup 10 "$@" 2>>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
{
for r in 1 2 3; do
echo "this build: train $(b)  inference $(i)      previous commit's library: train $(GDRN_S2_TW8=0 GDRN_HIP_LIB=$PWD/gdr-net_amd/lib/libgdrn_hip_prev.so b)  inference $(GDRN_S2_TW8=0 GDRN_HIP_LIB=$PWD/gdr-net_amd/lib/libgdrn_hip_prev.so i)"
done
} | tee $O/ab2.txt
