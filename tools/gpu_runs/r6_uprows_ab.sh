#!/bin/bash
# round 6: row form of the upsampling forward kernels against the previous build (gdr-net_amd/lib/libgdrn_hip_prev.so, built from the parent commit)
# on one box, interleaved; plus the per-kernel times of the upsampling kernels in both builds
O=$PWD/gpurun_out/r6_uprows_ab
mkdir -p $O
export PYTHONUNBUFFERED=1
PREV=$PWD/gdr-net_amd/lib/libgdrn_hip_prev.so
b() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 6 "$@" 2>>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
i() { timeout 300 python bench.py --fwd-only --no-cpu-baseline --no-roofline --no-extras --steps 100 --warmup 10 "$@" 2>>$O/err.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; }
{
for r in 1 2 3; do
echo "rows  train $(b)  inference $(i)"
echo "prev  train $(GDRN_HIP_LIB=$PREV b)  inference $(GDRN_HIP_LIB=$PREV i)"
done
} | tee $O/ab.txt
cd /tmp && export TMPDIR=/tmp
for v in rows prev; do
  if [ $v = prev ]; then export GDRN_HIP_LIB=$PREV; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr_$v -o p -- python /root/repo/bench.py --no-cpu-baseline --no-roofline --no-extras --steps 8 --warmup 3 > $O/tr_$v.log 2>&1
  echo "== $v" | tee -a $O/ab.txt
  find $O/tr_$v -name "p_kernel_stats.csv" | head -1 | xargs grep -h -i "upsample" | cut -c1-160 | tee -a $O/ab.txt
  rm -rf $O/tr_$v
done
