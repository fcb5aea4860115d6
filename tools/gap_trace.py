"""rocprofv3 --kernel-trace CSV of two-stream steps -> the main stream's idle gaps per step (which kernel ended, which started, what the side
stream ran meanwhile).  python tools/gap_trace.py <kernel_trace.csv> [steps=4] [min_gap_us=15]"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ming = float(sys.argv[3]) if len(sys.argv) > 3 else 15.0
marks = [i for i, r in enumerate(rows) if "pack_image_kernel" in r["Kernel_Name"]]
nm = lambda r: r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:44]
for k in range(max(1, len(marks) - nsteps), len(marks)):
    seg = rows[marks[k - 1]:marks[k]]
    t0 = int(seg[0]["Start_Timestamp"])
    qs = {}
    for r in seg:
        qs.setdefault(r["Queue_Id"], []).append(r)
    main = max(qs, key=lambda q: len(qs[q]))
    side = [r for q, v in qs.items() if q != main for r in v]
    print("step %d: span %.1f us" % (k, (int(rows[marks[k]]["Start_Timestamp"]) - t0) * 1e-3))
    prev = None
    for m in qs[main]:
        if prev is not None:
            g0, g1 = int(prev["End_Timestamp"]), int(m["Start_Timestamp"])
            if (g1 - g0) * 1e-3 > ming:
                during = [s for s in side if int(s["Start_Timestamp"]) < g1 and int(s["End_Timestamp"]) > g0]
                print("  gap %6.1f us at %7.1f: after %s, before %s | side: %s" % ((g1 - g0) * 1e-3, (g0 - t0) * 1e-3, nm(prev), nm(m),
                      ", ".join("%s [%.1f..%.1f]" % (nm(s)[:28], (int(s["Start_Timestamp"]) - t0) * 1e-3, (int(s["End_Timestamp"]) - t0) * 1e-3) for s in during) or "-"))
        prev = m
