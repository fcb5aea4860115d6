"""Build libgdrn_hip.so (gfx950) in-tree with hipcc.  Usage: python -m gdrnet_amd.build [--force]"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "lib", "libgdrn_hip.so")
LIB_F16 = os.path.join(HERE, "lib", "libgdrn_hip_f16.so")   # the same sources with -DGDRN_HALF_F16 (csrc/common.h)
SOURCES = ["conv_gemm.hip", "conv3x3_halo.hip", "conv3x3_v3.hip", "conv_wgrad.hip", "conv3x3_wgrad.hip", "norm.hip", "head_pose_loss.hip", "pack.hip", "multi.hip", "fc.hip", "postproc.hip", "stem.hip", "roi.hip", "workspace.hip"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def needs_build():
    if not os.path.exists(LIB) or not os.path.exists(LIB_F16):
        return True
    t = min(os.path.getmtime(LIB), os.path.getmtime(LIB_F16))
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "halo_xf.h"), os.path.join(INCLUDE, "gdrn_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
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
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
