#!/usr/bin/env python3
"""Register / scratch / occupancy table of the kernels of one .hip source (hipcc -Rpass-analysis=kernel-resource-usage).
    python tools/resusage.py gdr-net_amd/csrc/conv3x3_halo.hip [filter]"""
import os, re, subprocess, sys, tempfile
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with tempfile.TemporaryDirectory() as td:
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(root, "include"), "-c", src,
                        "-o", os.path.join(td, "o.o"), "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
if r.returncode:
    print(r.stderr[-4000:]); sys.exit(1)
cur = None
rows = {}
for line in r.stderr.splitlines():
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m: continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip(); rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1); rows[cur][k.strip()] = v.strip()
for name, d in rows.items():
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
    dn = re.sub(r"\(anonymous namespace\)::", "", dn)
    if flt and flt not in dn: continue
    print("%-70s VGPR %4s AGPR %3s scratch %4s occ %s spillV %s" % (dn[:70], d.get("VGPRs"), d.get("AGPRs"), d.get("ScratchSize [bytes/lane]"),
                                                                   d.get("Occupancy [waves/SIMD]"), d.get("VGPRs Spill")))
