"""CPU baseline thread sweep: the oracle's full training step (forward + 8 losses + backward + Ranger) at bs = 64 on this
box's host cores for several torch thread counts -- the measurement behind bench.py's `cpu_baseline.cores`.   python tools/cpu_threads_sweep.py"""
import os
import sys
import time

import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from gdrnet_amd import synth  # noqa: E402
from oracle import gdrn_oracle as O  # noqa: E402
from oracle import ranger_oracle as Rg  # noqa: E402

sd = synth.make_state_dict(0)
names = [k for k, v in sd.items() if v.is_floating_point() and "running" not in k]
for k in names:
    sd[k].requires_grad_(True)
batch = synth.make_batch(64, seed=1)
state = [dict() for _ in names]


def step():
    out = O.gdrn_forward(sd, batch, do_loss=True, training=True, bufs={})
    sum(out["loss_dict"].values()).backward()
    Rg.ranger_step([sd[k] for k in names], [sd[k].grad for k in names], state, lr=1e-4)
    for k in names:
        sd[k].grad = None


print("# host: %d logical CPUs; oracle training step at bs = 64 (1 warm-up, best of 2 timed)" % (os.cpu_count() or 0))
for nt in (8, 16, 32, 48, 64, 96, 128):
    if nt > (os.cpu_count() or 1):
        break
    torch.set_num_threads(nt)
    step()
    ts = []
    for _ in range(2):
        t0 = time.perf_counter()
        step()
        ts.append(time.perf_counter() - t0)
    print("threads %3d: %.2f s per step -> %.2f RoI/s" % (nt, min(ts), 64 / min(ts)), flush=True)
