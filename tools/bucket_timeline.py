#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of bench.py (GDRN_BUCKETS=5): when, inside the backward pass of one training step, each gradient
bucket (pnp | head | layer4 | layer3 | rest) is complete -- i.e. when its RCCL all-reduce could start -- and the all-reduce time
each bucket may take without being exposed.  Usage: python tools/bucket_timeline.py <p_kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
idx = [i for i, r in enumerate(rows) if name(r).startswith("pack_multi_kernel")]
a, b = idx[-3], idx[-2]
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"])
T = lambda r: (int(r["End_Timestamp"]) - t0) / 1e3
# backward starts at zero_multi_kernel; every bucket ends with its last unpack_multi / wgrad_reduce_multi launch (the engine appends
# grouped weight gradients, their reduction and the gradient unpack to the group that completes the bucket)
bw0 = next(T(r) for r in step if name(r).startswith("zero_multi_kernel"))
ends = [T(r) for r in step if name(r).startswith("wgrad_reduce_multi_kernel") or name(r).startswith("unpack_multi_kernel")]
opt0 = next((int(r["Start_Timestamp"]) - t0) / 1e3 for r in step if name(r).startswith("ranger_multi_kernel"))
end_step = T(step[-1])
# bucket boundaries: every bucket's last launch is its wgrad_reduce_multi (grouped weight gradients -> unpack -> reduce are appended in
# that order to the group that completes the bucket; since the Patch-PnP convs run on the halo weight-gradient kernel, pnp has one too)
names = [name(r) for r in step]
marks = [T(r) for i, r in enumerate(step) if names[i].startswith("wgrad_reduce_multi_kernel")]
assert len(marks) == 5, "expected the 5-bucket layout (GDRN_BUCKETS=5): %d reduce launches" % len(marks)
sizes_mb = {"pnp": 36.1, "head": 19.0, "layer4": 52.4, "layer3": 27.3, "rest": 5.4}
print("one training step (us from its first kernel): backward starts %.0f, optimizer starts %.0f, step ends %.0f" % (bw0, opt0, end_step))
for (k, mb), t in zip(sizes_mb.items(), marks):
    print("  bucket %-7s %5.1f MB fp32 complete at %7.0f us  -> %6.0f us of backward left to hide its all-reduce" % (k, mb, t, opt0 - t))
