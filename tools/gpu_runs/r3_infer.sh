#!/bin/bash
# round 3: eval-mode inference (bs = 64) -- bench line and steady-state kernel table
O=gpurun_out/r3_infer
mkdir -p $O
R=$PWD
export PYTHONUNBUFFERED=1
timeout 300 python bench.py --fwd-only --no-cpu-baseline --steps 30 --warmup 5 > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace -o p -- python $R/bench.py --fwd-only --steps 12 --warmup 3 --no-cpu-baseline > $R/$O/trace.log 2>&1
cd $R
python - <<'PY'
import csv, collections, glob
f = (glob.glob('gpurun_out/r3_infer/trace/*/p_kernel_trace.csv') + glob.glob('gpurun_out/r3_infer/trace/p_kernel_trace.csv'))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "pack_image" in r["Kernel_Name"]]
n = 8
seg = rows[marks[-n - 1]:marks[-1]]
per = collections.OrderedDict()
for r in seg:
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("unsigned short", "bf16")[:90]
    d = per.setdefault(k, [0, 0.0]); d[0] += 1; d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
tot = sum(v[1] for v in per.values()) / n
span = (int(rows[marks[-1]]["Start_Timestamp"]) - int(rows[marks[-n-1]]["Start_Timestamp"])) * 1e-6 / n
print("# eval-mode inference bs=64: %.1f launches, %.3f ms kernel time, %.3f ms wall per forward" % (len(seg) / n, tot, span))
for k, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print("%-92s %6.1f %8.3f %8.1f" % (k, c / n, t / n, t / c * 1e3))
# per-launch list of one forward
one = rows[marks[-2]:marks[-1]]
print("---- one forward, launch by launch")
for r in one:
    print("%8.1f  %s  grid %s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3, r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70], r["Grid_Size"]))
PY
rm -rf $O/trace
cut -c1-300 $O/bench.json
