"""rocprofv3 `*_kernel_stats.csv` -> the per-step summary table kept under profiles/.
Usage: python tools/summarize_stats.py <kernel_stats.csv> <profiled steps> [header line ...] > profiles/<name>.txt"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
for h in sys.argv[3:]:
    print("# " + h)
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"# sum of kernel time: {tot / steps / 1e6:.3f} ms per step over {steps} profiled steps")
print(f"{'kernel':<88}{'calls':>7}{'ms/step':>11}{'avg us':>11}{'%':>7}")
for r in rows:
    t = float(r["TotalDurationNs"])
    if t / tot < 0.0004:
        continue
    name = r["Name"].replace("(anonymous namespace)::", "").replace("unsigned short", "bf16")
    print(f"{name[:86]:<88}{int(r['Calls']):>7}{t / steps / 1e6:>11.3f}{float(r['AverageNs']) / 1e3:>11.1f}{100 * t / tot:>7.2f}")
