#!/bin/bash
# kernel-time tables (rocprofv3 --kernel-trace --stats) of the working tree and of _ab/base, side by side
mkdir -p gpurun_out/r2_abstats
O=$PWD/gpurun_out/r2_abstats
R=$PWD
K=${1:-upsample}
timeout 600 python -m pytest tests -m gpu -q -x -k "$K" > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log
cd /tmp && export TMPDIR=/tmp
for t in new base; do
  if [ $t = base ]; then d=$R/_ab/base; else d=$R; fi
  (cd $d && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$t -o p -- python $d/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extras > $O/$t.log 2>&1)
  python $R/tools/summarize_stats.py $O/$t/p_kernel_stats.csv 13 "$t" > $O/ks_$t.txt
done
cd $R
python - <<'PY'
import re
def load(f):
    d={}
    for l in open(f):
        m=re.match(r'(.{88})\s+(\d+)\s+([\d.]+)\s+([\d.]+)',l)
        if m: d[m.group(1).strip()[:70]]=(int(m.group(2)),float(m.group(3)))
    return d
a,b=load('gpurun_out/r2_abstats/ks_new.txt'),load('gpurun_out/r2_abstats/ks_base.txt')
print("total new %.3f base %.3f"%(sum(v[1] for v in a.values()),sum(v[1] for v in b.values())))
for k in sorted(set(a)|set(b),key=lambda k:-abs(a.get(k,(0,0))[1]-b.get(k,(0,0))[1]))[:22]:
    print("%-70s new %4d %.3f  base %4d %.3f  diff %+.3f"%(k,*a.get(k,(0,0)),*b.get(k,(0,0)),a.get(k,(0,0))[1]-b.get(k,(0,0))[1]))
PY
