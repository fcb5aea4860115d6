"""GPU bring-up debugging helper (not part of the product)."""
import sys

import torch

sys.path.insert(0, ".")
from gdrnet_amd import GDRN, synth  # noqa: E402
from gdrnet_amd.cfg import lm13_cfg  # noqa: E402

dev = "cuda:0"
B = 4
cpu_batch = synth.make_batch(B, seed=1)
batch = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in cpu_batch.items()}


def run(dtype, sync_before):
    cfg = lm13_cfg(device=dev)
    cfg.MODEL.CDPN.HIP_DTYPE = dtype
    model, _ = GDRN.build_model_optimizer(cfg)
    model.load_state_dict(synth.make_state_dict(0))
    model.train()
    _, L = model(batch["roi_img"], **synth.model_kwargs(batch, do_loss=True))
    k0 = "loss_coor_x"
    base = L[k0]._base
    if sync_before:
        torch.cuda.synchronize()
        print(dtype, "pre :", L[k0].item(), base.tolist()[:2], hex(base.data_ptr()), hex(L[k0].data_ptr()), flush=True)
    tot = sum(L.values())
    tot.backward()
    torch.cuda.synchronize()
    print(dtype, "post:", L[k0].item(), base.tolist()[:2], hex(base.data_ptr()), hex(L[k0].data_ptr()), "tot", tot.item(),
          "ver", base._version, L[k0]._version, flush=True)
    g = [p.grad for p in model.parameters()]
    near = [(n, hex(p.grad.data_ptr())) for n, p in model.named_parameters() if abs(p.grad.data_ptr() - base.data_ptr()) < 4096]
    print("   grads allocated within 4 KiB of the loss tensor:", near[:5], flush=True)


run("bf16", False)
run("bf16", True)
run("fp32", False)
