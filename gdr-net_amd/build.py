"""Build libgdrn_hip.so (gfx950) in-tree with hipcc.  Usage: python -m gdrnet_amd.build [--force]"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "lib", "libgdrn_hip.so")
LIB_F16 = os.path.join(HERE, "lib", "libgdrn_hip_f16.so")   # the same sources with -DGDRN_HALF_F16 (csrc/common.h)
SOURCES = ["conv_gemm.hip", "conv3x3_halo.hip", "block64.hip", "conv3x3s2.hip", "conv3x3s2_dgrad.hip", "conv3x3_v3.hip", "conv_wgrad.hip", "conv3x3_wgrad.hip", "norm.hip", "head_pose_loss.hip", "pack.hip", "multi.hip", "fc.hip", "postproc.hip", "stem.hip", "roi.hip", "workspace.hip"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def source_hash():
    """sha256 over the kernel sources + the C-ABI header (the same digest bench.py stamps its PMC summaries with): identifies the sources a
    built library belongs to"""
    import hashlib

    h = hashlib.sha256()
    for f in sorted(n for n in os.listdir(CSRC) if n.endswith((".hip", ".h")) and not n.startswith("_")) + ["../../include/gdrn_hip.h"]:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


STAMP = os.path.join(HERE, "lib", "source_hash.txt")   # written behind a successful build; git-ignored with the libraries, travels with them


def needs_build():
    """the libraries are rebuilt unless both exist AND were built from exactly these sources (content hash, not mtimes: a checkout, a copy to
    another box or a restored file must not make a stale library look current -- VERDICT r4)"""
    if not os.path.exists(LIB) or not os.path.exists(LIB_F16) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as f:
        return f.read().strip() != source_hash()


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    if os.path.exists(STAMP):
        os.remove(STAMP)   # no stamp while the objects are being replaced
    procs = []
    objs = {LIB: [], LIB_F16: []}
    for lib, tag, flags in ((LIB, "", []), (LIB_F16, "f16_", ["-DGDRN_HALF_F16"])):
        for s in SOURCES:
            o = os.path.join(HERE, "lib", tag + s.replace(".hip", ".o"))
            cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + flags + ["-I", INCLUDE, "-c", os.path.join(CSRC, s), "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((subprocess.Popen(cmd), cmd))
            objs[lib].append(o)
    for p, cmd in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    for lib, ob in objs.items():
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + ob
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(source_hash() + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
