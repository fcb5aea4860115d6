#!/usr/bin/env python3
"""bench.py -- RoI crops/s of GDR-Net's per-RoI training step (forward + losses + backward + fused Ranger
step) on MI355X: ResNet-34 + geometric head + Patch-PnP, 256x256 RoIs, bs=64 per GPU (BASELINE.json
configs[1] = LM-13 `a6_cPnP_lm13`, bf16 operands / fp32 accumulate), synthetic seeded RoI batch resident
in HBM, random-init (deterministic Kaiming-scaled) weights.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One JSON line on rank 0.  `roofline`: the conv kernel instantiation with the largest share of the step (picked live;
currently the halo-tiled 3x3 kernel, 8x16 pixels x 128 channels) -- every launch of it in one step is bracketed by HIP
events on the launch stream; achieved = sum(algorithmic FLOPs) / sum(durations - event overhead) against the
2.5 PFLOP/s dense bf16 MFMA peak.
`cpu_baseline`: the CPU oracle (a port of the reference path on stock PyTorch CPU kernels) timed on
this box's host cores on a bounded sample (bs=4, fwd+bwd, 32 threads).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_ROI_TRAIN = 68.16e9  # 34.08 GMAC fwd+bwd per 256x256 RoI
PEAK_BF16 = 2.5e15            # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_F32 = 157.3e12


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--bs", type=int, default=64, help="RoIs per GPU")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra no-optimizer / inference loops")
    ap.add_argument("--fwd-only", action="store_true", help="inference throughput (eval BN, test-mode pose decode)")
    return ap.parse_args()


def cpu_baseline(bs=16, iters=4):
    """Oracle fwd+bwd on the host cores (reported baseline, not a target)."""
    import torch

    from gdrnet_amd import synth
    from oracle import gdrn_oracle as O

    cores = min(os.cpu_count() or 1, 32)  # stock PyTorch CPU conv kernels stop scaling (and regress) beyond ~32 threads
    torch.set_num_threads(cores)
    sd = synth.make_state_dict(0)
    for v in sd.values():
        if v.is_floating_point():
            v.requires_grad_(True)
    batch = synth.make_batch(bs, seed=1)

    def step():
        out = O.gdrn_forward(sd, batch, do_loss=True, training=True, bufs={})
        sum(out["loss_dict"].values()).backward()
        for v in sd.values():
            v.grad = None

    step()
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    dt = (time.perf_counter() - t0) / iters

    def infer():
        with torch.no_grad():
            O.gdrn_forward(sd, batch, do_loss=False, training=False)

    infer()
    t0 = time.perf_counter()
    for _ in range(iters):
        infer()
    dti = (time.perf_counter() - t0) / iters
    return {"value": round(bs / dt, 3), "unit": "RoI/s", "cores": cores, "kind": "port",
            "sample": f"oracle (torch CPU fp32) fwd+bwd, bs={bs}, {iters} timed steps after 1 warm-up, no optimizer step",
            "inference_fwd_roi_s": round(bs / dti, 3)}


def measure_roofline(model, plan, kctx, dtype):
    """Bracket every launch of the dominant conv kernel in one forward+backward with HIP events."""
    import torch

    dom = f"conv_gemm_kernel<{'bf16' if dtype == 'bf16' else 'f32'},128,128>"
    st = plan.e._stream()
    ev = []

    def run(ops):
        for op in ops:
            meta = getattr(op, "meta", None)
            if meta is not None:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                op(st, kctx)
                b.record()
                allev.append((a, b, meta))
            else:
                op(st, kctx)

    # event-pair overhead (two back-to-back records with nothing in between), subtracted from every sample so that the
    # per-launch figure is the kernel's own duration as rocprofv3 --kernel-trace reports it
    cal = []
    for _ in range(64):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        b.record()
        cal.append((a, b))
    torch.cuda.synchronize()
    ovh_ms = sorted(a.elapsed_time(b) for a, b in cal)[len(cal) // 2]
    allev = []
    plan.e.dwp_flat.zero_()
    run(plan.fwd)
    plan.gw.fill_(1.0)
    run(plan.bwd)
    torch.cuda.synchronize()
    # dominant kernel = the conv instantiation with the largest share of this step's conv time
    per_k = {}
    for a, b, m in allev:
        per_k[m["kernel"]] = per_k.get(m["kernel"], 0.0) + a.elapsed_time(b)
    dom = max(per_k, key=per_k.get)
    ev = [x for x in allev if x[2]["kernel"] == dom]
    if os.environ.get("GDRN_LAYER_TABLE"):
        rows = sorted(((a.elapsed_time(b) * 1e3, m) for a, b, m in allev), key=lambda r: -r[0])
        with open(os.environ["GDRN_LAYER_TABLE"], "w") as f:
            for us, m in rows:
                f.write("%9.1f us %8.1f TF  %-40s %s\n" % (us, m["flops"] / us / 1e6, m["kernel"], m["layer"]))
    tot_ms = sum(max(a.elapsed_time(b) - ovh_ms, 1e-6) for a, b, _ in ev)
    flops = sum(m["flops"] for _, _, m in ev)
    achieved = flops / (tot_ms * 1e-3) / 1e12
    peak = (PEAK_BF16 if dtype == "bf16" else PEAK_F32) / 1e12
    # HBM-side bytes per launch of that kernel: rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE in separate runs; FETCH_SIZE doubled
    # as MI355X_MICROARCH.md prescribes for gfx950) cannot be collected from inside this process -- read the committed summary
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r01_hbm_traffic_bs64_bf16.json")
    if dtype == "bf16" and plan.B == 64 and os.path.exists(tpath):
        with open(tpath) as f:
            traffic = json.load(f).get(dom, {}).get("traffic_bytes_per_launch")
        traffic = round(traffic) if traffic else None
    return {"bound": "mfma", "kernel": dom, "launches_per_step": len(ev), "avg_launch_us": round(tot_ms * 1e3 / max(len(ev), 1), 2),
            "algorithmic_gflop_per_launch": round(flops / max(len(ev), 1) / 1e9, 3), "achieved": round(achieved, 2), "peak": peak,
            "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic, "event_overhead_us": round(ovh_ms * 1e3, 2)}


def roi_cropper_extras(B, dev, timed):
    """the GPU RoI cropper / target builder that feeds the path -- train-mode batch of B RoIs cut from synthetic
    640x480 / 720x540 frames resident in HBM (launches only; the per-RoI task table is prepared once, as a loader thread would)."""
    import torch

    from gdrnet_amd import roi_data, synth
    from gdrnet_amd.cfg import lm13_cfg

    d = synth.make_roi_frames(B, seed=9)
    frames = [torch.from_numpy(f).to(dev) for f in d["frames"]]
    rois = []
    for r in d["rois"]:
        q = dict(r)
        q.update(image=frames[r["frame"]], xyz_crop=torch.from_numpy(r["xyz_crop"]).to(dev), segmentation=torch.from_numpy(r["segmentation"]).to(dev),
                 mask_trunc=None if r["mask_trunc"] is None else torch.from_numpy(r["mask_trunc"]).to(dev))
        rois.append(q)
    crop = roi_data.RoiCropper(lm13_cfg(device=dev), extents=d["extents"], fps_points=d["fps_points"], device=dev)
    prep = crop.prepare(rois, train=True)
    t = timed(lambda: crop.launch(prep), 50)
    return {"roi_cropper_train_targets_roi_s": round(B / t, 0), "roi_cropper_us_per_batch": round(t * 1e6, 1)}


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    from gdrnet_amd import GDRN, synth
    from gdrnet_amd import dist as gdist
    from gdrnet_amd.cfg import lm13_cfg
    from gdrnet_amd.engine import LOSS_NAMES

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world == 1 and args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device(dev))

    cfg = lm13_cfg(device=dev)
    cfg.MODEL.CDPN.HIP_DTYPE = args.dtype
    model, opt = GDRN.build_model_optimizer(cfg)
    model.load_state_dict(synth.make_state_dict(0))
    B = args.bs
    batch = synth.make_batch(B, seed=1 + rank)  # each rank its own RoIs (weak scaling)
    batch = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    kw = synth.model_kwargs(batch, do_loss=not args.fwd_only)
    kw.pop("do_loss")
    if world > 1:
        gdist.broadcast_parameters(model)
        model.train()
        gdist.attach(model)

    if args.fwd_only:
        model.eval()

        def step():
            with torch.no_grad():
                return model(batch["roi_img"], do_loss=False, **kw)
    else:
        model.train()

        def step():
            return model.train_step(batch["roi_img"], optimizer=opt, **kw)

    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if not args.fwd_only:
        assert torch.isfinite(out).all(), out

    # SURVEY section 8(d) also asks for the step without the optimizer and for forward-only inference: short extra loops
    # OUTSIDE the timed region above (single GPU only, reported under "also")
    also = None
    if world == 1 and not args.fwd_only and not args.no_extras:
        def timed(fn, n):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / n

        n = max(5, min(args.steps, 20))
        t_noopt = timed(lambda: model.train_step(batch["roi_img"], optimizer=None, **kw), n)
        model.eval()
        kwi = {k: v for k, v in kw.items() if not k.startswith("gt_") and k != "sym_infos"}
        with torch.no_grad():
            t_inf = timed(lambda: model(batch["roi_img"], do_loss=False, **kwi), n)
        model.train()
        also = {"fwd_bwd_without_optimizer_roi_s": round(B / t_noopt, 1), "fwd_bwd_without_optimizer_ms": round(t_noopt * 1e3, 3),
                "inference_fwd_roi_s": round(B / t_inf, 1), "inference_fwd_ms": round(t_inf * 1e3, 3),
                "inference_fwd_tflops": round(B / t_inf * 22.823e9 / 1e12, 1)}
        also.update(roi_cropper_extras(B, dev, timed))

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * B * args.steps / dt
        flop_roi = FLOP_PER_ROI_TRAIN if not args.fwd_only else 22.823e9
        res = {
            "metric": "RoI crops/sec (fwd+bwd) at 256\u00d7256 bs=64" if not args.fwd_only else "RoI crops/sec (fwd, inference) at 256\u00d7256 bs=64",
            "value": round(value, 2), "unit": "RoI/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "LM-13 a6_cPnP_lm13 graph: ResNet-34 + RotWithRegionHead + ConvPnPNet, 256x256 RoIs, "
                                   f"bs={B}/GPU, {'train step: fwd + 8 losses + bwd + fused Ranger optimizer step (all inside the timed region)' if not args.fwd_only else 'inference fwd'}",
                       "global_batch": B * world, "parallelism": f"dp{world}",
                       "whole_step_tflops": round(value * flop_roi / 1e12, 2)},
        }
        if also is not None:
            res["also"] = also
        if not args.no_roofline and not args.fwd_only:
            eng = model.engine()
            plan = eng.plan(B, True, True)
            a = {k: kw.get(k) for k in ("gt_xyz", "gt_mask_trunc", "gt_mask_visib", "gt_region", "gt_ego_rot", "gt_points", "sym_infos",
                                        "gt_trans", "gt_trans_ratio", "roi_coord_2d", "roi_cams", "roi_centers", "roi_whs", "roi_extents",
                                        "resize_ratios")}
            _, plan, kctx = model._prepare(batch["roi_img"], True, a)
            res["roofline"] = measure_roofline(model, plan, kctx, args.dtype)
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline()
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
