"""rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; tools/gpu_runs/run_pmc_traffic.sh) -> profiles/r02_hbm_traffic_bs64_bf16.{txt,json}.
HBM-side bytes per launch of every kernel: counter unit KB -> bytes, FETCH_SIZE x2 (gfx950 tallies 128-byte requests at 64 B,
MI355X_MICROARCH.md 'HBM').  Usage: python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE/p_counter_collection.csv
gpurun_out/pmc_WRITE_SIZE/p_counter_collection.csv profiles/r02_hbm_traffic_bs64_bf16"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("unsigned short", "bf16").replace("void ", "")
    name = re.sub(r"\(.*$", "", name)  # drop the argument list
    return name.replace(", ", ",").strip()


def collect(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        tot[k] += float(r["Counter_Value"])
        cnt[k] += 1
    return tot, cnt


ft, fc = collect(sys.argv[1], "FETCH_SIZE")
wt, wc = collect(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(ft) | set(wt)):
    n = max(fc.get(k, 0), wc.get(k, 0))
    f = ft.get(k, 0.0) * 1024 * 2 / max(fc.get(k, 1), 1)
    w = wt.get(k, 0.0) * 1024 / max(wc.get(k, 1), 1)
    out[k] = {"launches": n, "fetch_bytes_per_launch": f, "write_bytes_per_launch": w, "traffic_bytes_per_launch": f + w}
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench  # noqa: E402  (kernel_source_hash: bench.py only trusts a summary measured on the same kernel sources)

out["__kernel_source_sha256_16__"] = bench.kernel_source_hash()
json.dump(out, open(sys.argv[3] + ".json", "w"), indent=1, sort_keys=True)
del out["__kernel_source_sha256_16__"]
rows = sorted(out.items(), key=lambda kv: -kv[1]["traffic_bytes_per_launch"] * kv[1]["launches"])
with open(sys.argv[3] + ".txt", "w") as f:
    f.write("# rocprofv3 --pmc FETCH_SIZE --kernel-trace / rocprofv3 --pmc WRITE_SIZE --kernel-trace (two separate passes) -- python bench.py --steps 2 --warmup 2\n")
    f.write("# per-launch means over the 4 profiled steps, bs=64 bf16.  Counter unit KB -> bytes; FETCH_SIZE x2 (gfx950 counts 128-byte requests at 64 B,\n")
    f.write("# MI355X_MICROARCH.md 'HBM'); Infinity-Cache hits are included in these memory-side counters.  JSON twin: %s.json (read by bench.py).\n" % sys.argv[3].split("/")[-1])
    f.write("%-52s %9s %14s %14s\n" % ("kernel", "launches", "fetch MB", "write MB"))
    for k, v in rows[:40]:
        f.write("%-52s %9d %14.1f %14.1f\n" % (k[:52], v["launches"], v["fetch_bytes_per_launch"] / 1e6, v["write_bytes_per_launch"] / 1e6))
