"""Cost of one dependent kernel launch inside a captured graph and on a plain stream: N tiny dependent kernels (an in-place add on 256 floats),
wall time / N.  The training step is 263 launches: what each one costs beyond its own work is the floor under the step's short kernels.
    python tools/ubench/launch_gap.py            (run under different HIP runtime settings, see tools/gpu_runs/r3_gap.sh)"""
import os
import time

import torch

x = torch.zeros(256, device="cuda")
N = 2000


def chain():
    for _ in range(N):
        x.add_(1.0)


chain()
torch.cuda.synchronize()
t0 = time.perf_counter()
chain()
torch.cuda.synchronize()
t_stream = (time.perf_counter() - t0) / N
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    chain()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        chain()
g.replay()
torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter()
    g.replay()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / N)
keys = ("AMD_OPT_FLUSH", "DEBUG_CLR_GRAPH_PACKET_CAPTURE", "HIP_FORCE_DEV_KERNARG", "DEBUG_HIP_GRAPH_BATCH_SIZE", "GPU_MAX_HW_QUEUES", "ROC_SYSTEM_SCOPE_SIGNAL",
        "DEBUG_HIP_FORCE_GRAPH_QUEUES", "GPU_FLUSH_ON_EXECUTION", "ROC_USE_FGS_KERNARG", "ROC_SKIP_KERNEL_ARG_COPY")
env = " ".join("%s=%s" % (k, os.environ[k]) for k in keys if k in os.environ) or "(defaults)"
print("%-60s stream %.2f us/launch | graph %.2f us/launch (min of 5; median %.2f)" % (env, t_stream * 1e6, min(ts) * 1e6, sorted(ts)[2] * 1e6), flush=True)

# the same with one of this library's own tiny kernels through the C-ABI (ctypes call: ~1.5 us of host time)
import sys  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
try:
    from gdrnet_amd import cabi  # noqa: E402

    lib = cabi.load()
    a = torch.zeros(3, 64, device="cuda")
    w = torch.ones(3, device="cuda")
    o = torch.zeros(64, device="cuda")

    def chain2():
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(N):
            lib.gdrn_combine3(a.data_ptr(), w.data_ptr(), o.data_ptr(), 64, st)

    chain2()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    chain2()
    t_host = (time.perf_counter() - t0) / N
    torch.cuda.synchronize()
    t_stream = (time.perf_counter() - t0) / N
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g2, stream=s):
            chain2()
    g2.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        g2.replay()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / N)
    print("%-60s C-ABI tiny kernel: host %.2f us/call, stream %.2f us/launch | graph %.2f us/launch" % (env, t_host * 1e6, t_stream * 1e6, min(ts) * 1e6), flush=True)
except Exception as e:  # noqa: BLE001
    print("C-ABI leg skipped:", e)
