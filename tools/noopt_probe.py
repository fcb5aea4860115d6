import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from gdrnet_amd import GDRN, synth
from gdrnet_amd.cfg import lm13_cfg
dev = "cuda:0"
cfg = lm13_cfg(device=dev)
model, opt = GDRN.build_model_optimizer(cfg)
model.load_state_dict(synth.make_state_dict(0))
model.train()
batch = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in synth.make_batch(64, seed=1).items()}
kw = synth.model_kwargs(batch, do_loss=True); kw.pop("do_loss")
def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
a = timed(lambda: model.train_step(batch["roi_img"], optimizer=opt, **kw))
b = timed(lambda: model.train_step(batch["roi_img"], optimizer=None, **kw))
c = timed(lambda: model.train_step(batch["roi_img"], optimizer=opt, **kw))
print(os.environ.get("TAG", "default"), "with optimizer %.3f | without %.3f | with again %.3f" % (a, b, c), flush=True)
