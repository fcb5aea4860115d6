"""rocprofv3 --kernel-trace CSV -> steady-state per-step kernel table.
A step starts with the image repack of its forward pass (`pack_image_kernel`); the last N complete steps of the trace are averaged, so one-time
work (plan construction, optimizer-state initialisation, warm-up allocations) is not attributed to the step.
Usage: python tools/trace_steps.py <kernel_trace.csv> [N=5] [header ...] > profiles/<name>.txt"""
import collections
import csv
import sys


def short(name):
    return (name.replace("(anonymous namespace)::", "").replace("void ", "").replace("unsigned short", "bf16").replace("at::native::", ""))


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    for h in sys.argv[3:]:
        print("# " + h)
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "pack_image_kernel" in r["Kernel_Name"]]
    if len(marks) < n + 1:
        raise SystemExit("only %d step marks in the trace" % len(marks))
    a, b = marks[-n - 1], marks[-1]
    seg = rows[a:b]
    per = collections.OrderedDict()
    for r in seg:
        k = short(r["Kernel_Name"])
        d = per.setdefault(k, [0, 0.0])
        d[0] += 1
        d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    span = (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) * 1e-6 / n
    tot = sum(v[1] for v in per.values()) / n
    print("# steady state over the last %d steps of the trace: %.1f kernel launches per step, %.3f ms of kernel time per step, %.3f ms per step wall (first launch to first launch)" % (
        n, len(seg) / n, tot, span))
    print("%-96s%10s%11s%10s%7s" % ("kernel", "calls/step", "ms/step", "avg us", "%"))
    for k, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print("%-96s%10.1f%11.3f%10.1f%7.2f" % (k[:94], c / n, t / n, t / c * 1e3, 100 * t / n / tot))


if __name__ == "__main__":
    main()
