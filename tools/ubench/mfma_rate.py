"""python tools/ubench/mfma_rate.py -- cycles (s_memtime) per v_mfma_f32_32x32x16_bf16 of one wave, by mode / waves per SIMD / data"""
import ctypes as C
import os
import time

import torch

R = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(R, "mfma_rate.so"))
lib.run.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
out = torch.zeros(2 + 256 * 8, dtype=torch.int64, device="cuda")
for data in ("zeros", "randn"):
    src = (torch.zeros(1024 * 8, device="cuda") if data == "zeros" else torch.randn(1024 * 8, device="cuda")).to(torch.bfloat16)
    for mode in (0, 1, 2):
        for threads in (256, 512):
            iters = 2000
            lib.run(mode, threads, 256, 10, out.data_ptr(), src.data_ptr())
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            lib.run(mode, threads, 256, iters, out.data_ptr(), src.data_ptr())
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            nw = threads // 64
            cyc = out[2:2 + 256 * 8].view(256, 8)[:, :nw].double().mean().item()
            nm = iters * 16
            tf = 256 * nw * nm * 2 * 32 * 32 * 16 / (ms * 1e-3) / 1e12
            print("%-5s mode %d  %d waves/SIMD: %.1f ticks per MFMA per wave (%.1f per SIMD), %.3f ms -> %.0f TF, tick rate %.2f GHz" % (
                data, mode, nw // 4, cyc / nm, cyc / nm / (nw // 4), ms, tf, cyc / (ms * 1e6)))
