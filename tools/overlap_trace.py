"""rocprofv3 --kernel-trace CSV of a step with side-stream work -> who waits for whom.  python tools/overlap_trace.py <kernel_trace.csv>"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "pack_image_kernel" in r["Kernel_Name"]]
a, b = marks[-2], marks[-1]
seg = rows[a:b]
t0 = int(seg[0]["Start_Timestamp"])
qs = {}
for r in seg:
    qs.setdefault(r["Queue_Id"], []).append(r)
print("step span %.1f us, queues: %s" % ((int(rows[b]["Start_Timestamp"]) - t0) * 1e-3, {q: len(v) for q, v in qs.items()}))
main = max(qs, key=lambda q: len(qs[q]))
for q, v in qs.items():
    if q == main:
        continue
    for r in v:
        s, e = (int(r["Start_Timestamp"]) - t0) * 1e-3, (int(r["End_Timestamp"]) - t0) * 1e-3
        # main-stream kernels running during [s, e]
        ov = [m for m in qs[main] if int(m["Start_Timestamp"]) - t0 < e * 1e3 and int(m["End_Timestamp"]) - t0 > s * 1e3]
        busy = sum(min(e, (int(m["End_Timestamp"]) - t0) * 1e-3) - max(s, (int(m["Start_Timestamp"]) - t0) * 1e-3) for m in ov)
        print("side %-34s %8.1f -> %8.1f us (%6.1f us) | main stream busy %6.1f us of it with %d kernels" % (
            r["Kernel_Name"].replace("(anonymous namespace)::", "")[:34], s, e, e - s, busy, len(ov)))
last_main = max(int(m["End_Timestamp"]) for m in qs[main]) - t0
print("main stream's last kernel ends at %.1f us" % (last_main * 1e-3))
# gaps on the main stream > 5 us
prev = None
for m in qs[main]:
    if prev is not None:
        gap = (int(m["Start_Timestamp"]) - int(prev["End_Timestamp"])) * 1e-3
        if gap > 5:
            print("  main-stream gap %.1f us before %s at %.1f us" % (gap, m["Kernel_Name"].replace("(anonymous namespace)::", "")[:40], (int(m["Start_Timestamp"]) - t0) * 1e-3))
    prev = m
