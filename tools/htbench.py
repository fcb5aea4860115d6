"""Bring-up aid: the head tail forward kernels alone (bs = 64): eval variant, train variant (map-loss sums), between HIP events.
python tools/htbench.py   (GDRN_HIP_LIB selects the library)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdrnet_amd import cabi  # noqa: E402
from gdrnet_amd.cabi import BF16, PREZEROED, check, ptr  # noqa: E402

lib = cabi.load()
dev = "cuda:0"
B, HW, nreg, hs = 64, 4096, 64, 72
M = B * HW
g = torch.Generator(device=dev).manual_seed(0)
head = torch.randn(M, hs, device=dev, generator=g)
c2d = torch.rand(B, 2, HW, device=dev, generator=g)
ext = torch.rand(B, 3, device=dev, generator=g) * 0.2 + 0.05
pnp = torch.zeros(M, 128, device=dev, dtype=torch.bfloat16)
gxyz = torch.rand(B, 3, HW, device=dev, generator=g)
mv = (torch.rand(B, HW, device=dev, generator=g) > 0.4).float()
mt = (torch.rand(B, HW, device=dev, generator=g) > 0.3).float()
greg = torch.randint(0, 65, (B, HW), device=dev, generator=g)
acc = torch.zeros(8 + 8 * 4096, device=dev, dtype=torch.float64)
st = torch.cuda.current_stream().cuda_stream


def t(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


ev = lambda: check(lib.gdrn_head_tail_fwd(ptr(head), hs, ptr(c2d), ptr(ext), ptr(pnp), 128, B, HW, nreg, BF16 | PREZEROED, st), "fwd")
tr = lambda: check(lib.gdrn_head_tail_loss_fwd(ptr(head), hs, ptr(c2d), ptr(ext), ptr(pnp), 128, ptr(gxyz), ptr(mv), ptr(mt), ptr(greg), ptr(acc), B, HW, nreg,
                                               BF16 | PREZEROED, st), "loss_fwd")
nrows = lib.gdrn_head_tail_loss_rows(B, HW, nreg, hs, 128)
losses = torch.zeros(8, device=dev)
trr = lambda: check(lib.gdrn_head_tail_loss_fwd(ptr(head), hs, ptr(c2d), ptr(ext), ptr(pnp), 128, ptr(gxyz), ptr(mv), ptr(mt), ptr(greg), ptr(acc), B, HW, nreg,
                                                BF16 | PREZEROED | cabi.ACC_ROWS, st), "loss_fwd rows")
fin = lambda: check(lib.gdrn_map_loss_finalize(ptr(acc), B, HW, ptr(losses), st), "fin")
finr = lambda: check(lib.gdrn_map_loss_finalize_rows(ptr(acc), nrows, B, HW, ptr(losses), st), "fin rows")
dh = torch.zeros(M, 128, device=dev, dtype=torch.bfloat16)
dpn = (torch.randn(M, 128, device=dev, generator=g) * 0.01).to(torch.bfloat16)
gw = torch.ones(5, device=dev)
bw = lambda: check(lib.gdrn_head_tail_bwd(ptr(head), hs, ptr(pnp), ptr(dpn), 128, ptr(ext), ptr(gxyz), ptr(mv), ptr(mt), ptr(greg), ptr(acc), ptr(gw), ptr(dh), 128,
                                          B, HW, nreg, BF16 | PREZEROED, st), "bwd")
print("train with partial rows %.1f us | finalize %.1f us, finalize_rows %.1f us | head_tail_bwd %.1f us" % (t(trr), t(fin), t(finr), t(bw)))
print("GDRN_HT_DBG=%s GDRN_HT_BLOCKS=%s: eval %.1f us   train (with the memset) %.1f us" % (os.environ.get("GDRN_HT_DBG"), os.environ.get("GDRN_HT_BLOCKS"), t(ev), t(tr)))
