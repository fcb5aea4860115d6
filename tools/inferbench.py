"""Eval-mode (inference) forward at bs=64, bf16: steady-state loop for rocprofv3 --kernel-trace --stats (BatchNorm folded into the convs).
Usage: python tools/inferbench.py [steps]"""
import os
import sys
import time

import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from gdrnet_amd import GDRN, synth  # noqa: E402
from gdrnet_amd.cfg import lm13_cfg  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cfg = lm13_cfg(device="cuda:0")
model, _ = GDRN.build_model_optimizer(cfg)
model.load_state_dict(synth.make_state_dict(0))
model.eval()
batch = {k: (v.to("cuda:0") if isinstance(v, torch.Tensor) else v) for k, v in synth.make_batch(64, seed=1).items()}
kw = synth.model_kwargs(batch, do_loss=False)
with torch.no_grad():
    for _ in range(5):
        model(batch["roi_img"], **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        model(batch["roi_img"], **kw)
    torch.cuda.synchronize()
print("inference forward: %.3f ms per 64 RoIs" % ((time.perf_counter() - t0) / steps * 1e3))
