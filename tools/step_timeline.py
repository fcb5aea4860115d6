"""rocprofv3 --kernel-trace CSV -> every kernel of the last complete training step in start order: queue, start, end, duration, gap to the
previous kernel of the same queue, name.  python tools/step_timeline.py <kernel_trace.csv> [step_from_end=1]"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
marks = [i for i, r in enumerate(rows) if "pack_image_kernel" in r["Kernel_Name"]]
a, b = marks[-1 - back], marks[-back]
seg = rows[a:b]
t0 = int(seg[0]["Start_Timestamp"])
qn = {}
for r in seg:
    qn.setdefault(r["Queue_Id"], len(qn))
last = {}
nm = lambda r: r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("unsigned short", "bf16")[:70]
print("# step span %.1f us (first kernel to the next step's first kernel), %d kernels, %d queues" % ((int(rows[b]["Start_Timestamp"]) - t0) * 1e-3, len(seg), len(qn)))
print("# q   start us     end us    dur us  gap-in-queue us  grid x wg   kernel")
for r in seg:
    q = qn[r["Queue_Id"]]
    s, e = (int(r["Start_Timestamp"]) - t0) * 1e-3, (int(r["End_Timestamp"]) - t0) * 1e-3
    gap = s - last[q] if q in last else 0.0
    last[q] = e
    try:
        grid = "%6d x %-4d" % (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Workgroup_Size_X"]))
    except Exception:
        grid = ""
    print("%s%d %10.1f %10.1f %9.1f %9.1f  %s  %s" % ("    " * q, q, s, e, e - s, gap, grid, nm(r)))
