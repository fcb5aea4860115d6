"""Robustness sweep (not product): two fused train steps + one eval forward at odd / small / non-power-of-two batch sizes in both
precision modes; prints whether everything stayed finite.  Run on an MI355X: python tools/sweep_batch_sizes.py"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
from gdrnet_amd import GDRN, synth
from gdrnet_amd.cfg import lm13_cfg
for dtype in ("bf16", "fp32"):
    cfg = lm13_cfg(device="cuda:0"); cfg.MODEL.CDPN.HIP_DTYPE = dtype
    model, opt = GDRN.build_model_optimizer(cfg); model.load_state_dict(synth.make_state_dict(0))
    for B in (1, 2, 3, 5, 7, 16, 33):
        batch = {k: (v.to("cuda:0") if isinstance(v, torch.Tensor) else v) for k, v in synth.make_batch(B, seed=B).items()}
        kw = synth.model_kwargs(batch, do_loss=True); kw.pop("do_loss")
        model.train()
        for _ in range(2):
            l = model.train_step(batch["roi_img"], optimizer=opt, **kw)
        kwi = synth.model_kwargs(batch, do_loss=False)
        model.eval()
        with torch.no_grad():
            od = model(batch["roi_img"], **kwi)
        torch.cuda.synchronize()
        ok = bool(torch.isfinite(l).all()) and bool(torch.isfinite(od["rot"]).all())
        print(dtype, B, "ok" if ok else "NONFINITE", float(l.sum()))
