#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of `GDRN_BUCKETS=5 python bench.py --dist-force` (the data-parallel step on one GPU: one-rank RCCL group,
every bucket goes through the reducer): when, inside one training step, each gradient bucket (pnp | head | layer4 | layer3 | rest) is
complete -- i.e. when its all-reduce can start --, when its Ranger update and operand re-pack ran (reducer stream, behind the exchange),
and how much of the backward pass is left to hide the exchange.  Usage: python tools/bucket_timeline.py <p_kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
idx = [i for i, r in enumerate(rows) if name(r).startswith("pack_image_kernel")]   # first kernel of a step
a, b = idx[-3], idx[-2]
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"])
S = lambda r: (int(r["Start_Timestamp"]) - t0) / 1e3
E = lambda r: (int(r["End_Timestamp"]) - t0) / 1e3
bw0 = next(S(r) for r in step if name(r).startswith("zero_multi_kernel"))
# a bucket is complete when its partial-tile reduction (the last gradient kernel of the bucket, side stream) has finished
marks = [E(r) for r in step if name(r).startswith("wgrad_reduce_multi_kernel")]
rang = [(S(r), E(r)) for r in step if name(r).startswith("ranger_multi_kernel")]
packs = [E(r) for r in step if name(r).startswith("pack_multi_kernel")]
assert len(marks) == 5, "expected the 5-bucket layout (GDRN_BUCKETS=5): %d reduce launches" % len(marks)
main_q = max(set(r["Queue_Id"] for r in step), key=lambda q: sum(1 for r in step if r["Queue_Id"] == q))
chain_end = max(E(r) for r in step if r["Queue_Id"] == main_q)
end_step = max(E(r) for r in step)
sizes_mb = {"pnp": 36.1, "head": 19.0, "layer4": 52.4, "layer3": 27.3, "rest": 5.4}
print("one training step (us from its first kernel): backward starts %.0f, gradient chain ends %.0f, last kernel ends %.0f; %d hardware queues"
      % (bw0, chain_end, end_step, len(set(r["Queue_Id"] for r in step))))
for i, ((k, mb), t) in enumerate(zip(sizes_mb.items(), marks)):
    rg = rang[i] if i < len(rang) else (float("nan"),) * 2
    pk = packs[i] if i < len(packs) else float("nan")
    print("  bucket %-7s %5.1f MB fp32 complete at %7.0f us -> %6.0f us of chain left to hide its all-reduce | Ranger update %7.0f..%7.0f us, operands re-packed by %7.0f us"
          % (k, mb, t, chain_end - t, rg[0], rg[1], pk))
