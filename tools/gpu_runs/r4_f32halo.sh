#!/bin/bash
# round 4: fp32 (parity mode) 3x3 stride-1 convs on the halo kernel -- kernel tests, the 1e-4 end-to-end tests, step time A/B
O=gpurun_out/r4_f32halo
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "halo_forward_and_dgrad" > $O/kernels.log 2>&1; echo "rc $?" >> $O/kernels.log; tail -4 $O/kernels.log
b() { timeout 400 python bench.py --no-cpu-baseline --no-roofline --no-extras --dtype fp32 --steps 6 --warmup 2 2>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
for rep in 1 2; do echo "fp32 generic: $(GDRN_HALO_F32=0 b)   fp32 halo: $(GDRN_HALO_F32=1 b)"; done | tee $O/ab.txt
timeout 1500 python -m pytest tests/test_e2e_gpu.py -q -m gpu -k "fp32" -s > $O/e2e.log 2>&1; echo "rc $?" >> $O/e2e.log; grep -v "^Randomly" $O/e2e.log | tail -15
GDRN_LAYER_TABLE=$PWD/$O/layers_f32.txt timeout 400 python bench.py --no-cpu-baseline --no-extras --dtype fp32 --steps 4 --warmup 2 > $O/bench_f32.json 2>>$O/err.log; head -30 $O/layers_f32.txt
