#!/bin/bash
# same-box A/B of two library builds: _ab/lib_A.so _ab/lib_B.so ... (GDRN_HIP_LIB)
for rep in 1 2 3; do
for lib in "$@"; do
  r=$(GDRN_HIP_LIB=$PWD/_ab/$lib python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 40 --warmup 8 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])")
  echo "$lib -> $r ms/step"
done
done
