#!/usr/bin/env python3
"""Where do a kernel's scratch (spill) instructions sit relative to its MFMA loop?  
    python tools/check_scratch.py gdr-net_amd/csrc/conv3x3_v3.hip "conv3x3_v3_kernelILi16ELi256ELi2ELi4ELi1ELi3E"  [more mangled-name substrings]
Compiles the source to gfx950 assembly, and for every matching kernel prints each scratch_load / scratch_store with the number of MFMAs in front of
and behind it in program order, the loop (backward-branch target .. branch) it lies in, if any, and how many MFMAs that loop holds -- a spill inside
a loop that holds MFMAs is a spill in the hot loop; one in a loop-free prologue / epilogue (or in a loop without MFMAs) is not."""
import os, re, subprocess, sys, tempfile
src, pats = sys.argv[1], sys.argv[2:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "k.s")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(root, "include"), "-S", "--cuda-device-only",
                        src, "-o", out] + [a for a in os.environ.get("EXTRA_FLAGS", "").split() if a], capture_output=True, text=True)
    if r.returncode:
        print(r.stderr[-3000:]); sys.exit(1)
    text = open(out).read().splitlines()
kern, cur = {}, None
for ln in text:
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        cur = m.group(1); kern[cur] = []
        continue
    if cur is not None:
        if ln.startswith("\t.end_amdhsa_kernel") or ln.startswith(".Lfunc_end"):
            cur = None
            continue
        kern[cur].append(ln)
for name, body in kern.items():
    if pats and not any(p in name for p in pats):
        continue
    ins, labels = [], {}
    for ln in body:
        s = ln.strip()
        m = re.match(r"^(\.LBB\w+):", s)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if not s or s.startswith((".", ";", "//")):
            continue
        ins.append(s)
    mf = [i for i, s in enumerate(ins) if s.startswith("v_mfma")]
    sc = [i for i, s in enumerate(ins) if s.startswith(("scratch_load", "scratch_store"))]
    loops = []
    for i, s in enumerate(ins):
        m = re.match(r"^s_cbranch\w*\s+(\.LBB\w+)|^s_branch\s+(\.LBB\w+)", s)
        if m:
            t = labels.get(m.group(1) or m.group(2))
            if t is not None and t <= i:
                loops.append((t, i))
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    print(f"{dn[:110]}: {len(ins)} instructions, {len(mf)} MFMAs, {len(sc)} scratch instructions, {len(loops)} loops")
    for i in sc:
        inside = [(a, b) for a, b in loops if a <= i <= b]
        where = "no loop"
        if inside:
            a, b = min(inside, key=lambda ab: ab[1] - ab[0])
            where = f"loop [{a}..{b}] holding {sum(1 for j in mf if a <= j <= b)} MFMAs"
        print(f"   #{i:6d} {ins[i][:60]:60s} MFMAs before {sum(1 for j in mf if j < i):5d} / after {sum(1 for j in mf if j > i):5d}   {where}")
