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
this box's host cores per SURVEY.md section 8(d): B = 4 and B = 64, forward-only and forward + backward + Ranger, median of 5
after 2 warm-ups, core count and CPU model stated (about 50 s of CPU work).
`also`: the step without the optimizer, eval-mode inference, a sustained (>= 2 s timed) run of the same step, the fp32 parity
mode (the mode the 1e-4 pose bound belongs to) with its own MFMA roofline fraction, the RoI cropper.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # before HIP initialises (gdr-net_amd/__init__.py explains)

FLOP_PER_ROI_TRAIN = 68.16e9  # 34.08 GMAC fwd+bwd per 256x256 RoI
PEAK_BF16 = 2.5e15            # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_F32 = 157.3e12


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--bs", type=int, default=64, help="RoIs per GPU")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-threads-sweep", action="store_true",
                    help="only the cpu_baseline leg, for several torch thread counts (the measurement behind cpu_baseline.cores; no GPU work)")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra no-optimizer / inference loops")
    ap.add_argument("--fwd-only", action="store_true", help="inference throughput (eval BN, test-mode pose decode)")
    ap.add_argument("--dist-force", action="store_true",
                    help="run the data-parallel code path (RCCL process group, parameter broadcast, bucketed side-stream all-reduce) even with one rank")
    ap.add_argument("--comm-dtype", default=os.environ.get("GDRN_COMM_DTYPE", "fp32"), choices=["fp32", "bf16"], help="wire format of the gradient buckets")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend (nccl = RCCL; gloo only with --dry)")
    ap.add_argument("--dry", action="store_true",
                    help="no kernels: the launcher, the process group and the bucketed gradient-exchange protocol on the dry engine's REAL bs=64 "
                         "backward launch list (host tensors); what the CPU test of the N > 1 path runs")
    return ap.parse_args()


def _json_error(args, msg, **extra):
    """ONE JSON line also when the run cannot happen (the driver parses the last line of stdout)"""
    res = {"metric": "RoI crops/sec (fwd+bwd) at 256\u00d7256 bs=64", "value": None, "unit": "RoI/s", "n_gpus": args.gpus, "steps": args.steps,
           "warmup": args.warmup, "error": msg}
    res.update(extra)
    print(json.dumps(res), flush=True)


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no rendezvous in the environment: start the N ranks ourselves (torch.distributed.run, one
    process per GPU, 127.0.0.1) and pass rank 0's JSON line through.  Returns the process exit code."""
    import socket
    import subprocess

    if not args.dry:
        import torch

        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            _json_error(args, f"--gpus {args.gpus} needs {args.gpus} visible GPUs, this box has {have}", gpus_visible=have)
            return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), GDRN_BENCH_LAUNCHED="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    pr = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in pr.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if pr.returncode != 0 or line is None:
        _json_error(args, f"the {args.gpus}-rank launch failed (exit code {pr.returncode}); ranks' stdout tail: " + pr.stdout[-400:].replace("\n", " | "))
        return pr.returncode or 1
    print(line, flush=True)
    return 0


def dry_main(args):
    """--dry: every rank builds the dry engine's REAL backward launch list at bs = args.bs on host tensors and walks it; the reducer fires at the
    plan's own bucket marks (gloo all-reduce per bucket, deferred 1/world), the mean must land in every parameter's gradient view.  Timed like
    the real run (barrier, K steps, max over ranks); `value` counts RoIs per second of protocol time -- it measures the launcher, not a GPU."""
    import torch
    import torch.distributed as dist

    from gdrnet_amd import GDRN as G
    from gdrnet_amd.cfg import lm13_cfg
    from gdrnet_amd.dist import GradReducer
    from gdrnet_amd.engine import Engine

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        dist.init_process_group(args.backend if args.backend == "gloo" else "gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    model, _ = G.build_model_optimizer(lm13_cfg(device="cpu"))
    eng = Engine(dict(model.named_parameters()), dict(model.named_buffers()), dtype="bf16", dry=True)
    plan = eng.plan(args.bs, True, True)
    flat, bounds = eng.grad_flat, eng.bucket_bounds
    red = GradReducer(flat, bounds, average=True, defer_scale=True, comm_dtype=args.comm_dtype)

    def before(i):
        lo, hi = bounds[i]
        flat[lo:hi] = float(rank + 1)

    def step():
        flat.zero_()
        fired = plan.walk_backward(red.on_bucket, before)
        red.finish()
        return fired

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fired = step()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    mean = sum(r + 1 for r in range(world)) / world
    ok = bool(torch.allclose(flat, torch.full_like(flat, mean))) and [b for _, b in fired] == list(range(len(bounds)))
    if world > 1:
        t = torch.tensor([dt, 0.0 if ok else 1.0], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, ok = float(t[0]), float(t[1]) == 0.0
    if rank == 0:
        print(json.dumps({"metric": "RoI crops/sec (fwd+bwd) at 256\u00d7256 bs=64", "value": round(world * args.bs * args.steps / dt, 2), "unit": "RoI/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic", "dry": True,
                          "process_group": {"backend": "gloo", "world_size": dist.get_world_size() if world > 1 else 1},
                          "buckets": len(bounds), "bucket_mb": [round((hi - lo) * 4 / 1e6, 1) for lo, hi in bounds],
                          "backward_launches": len(plan.bwd), "protocol_ok": ok,
                          "config": {"workload": "DRY: no kernels -- launcher + process group + bucketed gradient exchange on the real bs=%d backward "
                                                 "launch list (host tensors)" % args.bs, "global_batch": args.bs * world, "parallelism": f"dp{world}"}}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0 if ok else 1


def kernel_source_hash():
    """sha256 over the kernel sources + the C-ABI header: identifies the build a PMC traffic summary belongs to (the digest
    gdrnet_amd.build stamps a built library with)."""
    from gdrnet_amd import build

    return build.source_hash()


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform

    return platform.processor() or "unknown"


def _pose_outputs(o):
    return {k: o[k].detach().float().cpu() for k in ("rot6d", "t_", "rot", "trans")}


def cpu_baseline(parity_bs=None):
    """the CPU restatement of the reference path (oracle/, stock PyTorch CPU kernels, fp32) on this box's host
    cores, same synthetic batch, B = 4 and B = 64, forward-only and the full training step (forward + 8 losses + backward + Ranger,
    oracle/ranger_oracle.py), median of 5 timed iterations after 2 warm-ups.  A reported baseline, not a target.
    parity_bs (4 | 64): the leg also keeps the oracle's pose outputs of its first train-mode forward at that batch size -- on the plain and on
    the conditioned synthetic weights -- for the `parity` object of the bench line (second return value)."""
    import statistics

    import torch

    from gdrnet_amd import synth
    from oracle import gdrn_oracle as O
    from oracle import ranger_oracle as R

    cores = min(os.cpu_count() or 1, 32)  # stock PyTorch CPU conv kernels stop scaling (and regress) beyond ~32 threads
    torch.set_num_threads(cores)

    def med(fn, warm=2, n=5):
        for _ in range(warm):
            fn()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return statistics.median(ts)

    res = {}
    ref_pose = None
    for bs in (4, 64):
        sd = synth.make_state_dict(0)
        names = [k for k, v in sd.items() if v.is_floating_point() and "running" not in k]
        for k in names:
            sd[k].requires_grad_(True)
        batch = synth.make_batch(bs, seed=1)
        state = [dict() for _ in names]
        if bs == parity_bs:
            with torch.no_grad():
                ref_pose = {"plain": _pose_outputs(O.gdrn_forward(sd, batch, do_loss=True, training=True, bufs={})),
                            "conditioned": _pose_outputs(O.gdrn_forward(synth.conditioned_state_dict(0), batch, do_loss=True, training=True, bufs={}))}

        def train_step(opt=True):
            out = O.gdrn_forward(sd, batch, do_loss=True, training=True, bufs={})
            sum(out["loss_dict"].values()).backward()
            if opt:
                R.ranger_step([sd[k] for k in names], [sd[k].grad for k in names], state, lr=1e-4)
            for k in names:
                sd[k].grad = None

        def infer():
            with torch.no_grad():
                O.gdrn_forward(sd, batch, do_loss=False, training=False)

        t_train = med(train_step)
        t_noopt = med(lambda: train_step(False), warm=1, n=3) if bs == 4 else None
        t_inf = med(infer)
        res[bs] = (t_train, t_noopt, t_inf)
    t64, _, i64 = res[64]
    t4, n4, i4 = res[4]
    return {"value": round(64 / t64, 3), "unit": "RoI/s", "cores": cores, "cpu_model": _cpu_model(), "kind": "port",
            "__ref_pose__": ref_pose,
            "sample": "oracle/ (CPU restatement of the reference path, torch CPU fp32): fwd + 8 losses + bwd + Ranger step at bs=64, median of 5 timed "
                      "steps after 2 warm-ups, %d threads" % cores,
            "train_step_bs4_roi_s": round(4 / t4, 3), "fwd_bwd_without_optimizer_bs4_roi_s": round(4 / n4, 3),
            "inference_fwd_bs64_roi_s": round(64 / i64, 3), "inference_fwd_bs4_roi_s": round(4 / i4, 3)}


def cpu_threads_sweep():
    """the cpu_baseline leg's training step at bs = 64 for several thread counts: one line per count (profiles/r03_cpu_threads_sweep.txt)"""
    import torch

    from gdrnet_amd import synth
    from oracle import gdrn_oracle as O
    from oracle import ranger_oracle as R

    sd = synth.make_state_dict(0)
    names = [k for k, v in sd.items() if v.is_floating_point() and "running" not in k]
    for k in names:
        sd[k].requires_grad_(True)
    batch = synth.make_batch(64, seed=1)
    state = [dict() for _ in names]

    def step():
        out = O.gdrn_forward(sd, batch, do_loss=True, training=True, bufs={})
        sum(out["loss_dict"].values()).backward()
        R.ranger_step([sd[k] for k in names], [sd[k].grad for k in names], state, lr=1e-4)
        for k in names:
            sd[k].grad = None

    ncpu = os.cpu_count() or 1
    print("# %s, %d logical CPUs; oracle training step (fwd + 8 losses + bwd + Ranger) at bs = 64, 1 warm-up, best of 2" % (_cpu_model(), ncpu))
    for nt in (8, 16, 32, 48, 64, 96, 128, 192):
        if nt > ncpu:
            break
        torch.set_num_threads(nt)
        step()
        ts = []
        for _ in range(2):
            t0 = time.perf_counter()
            step()
            ts.append(time.perf_counter() - t0)
        print("threads %3d: %6.2f s per step -> %5.2f RoI/s" % (nt, min(ts), 64 / min(ts)), flush=True)


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
                parts = getattr(op, "parts", None)  # conv launch + the BatchNorm-coefficient launch behind it: time the conv alone
                a.record()
                (parts[0] if parts else op)(st, kctx)
                b.record()
                if parts:
                    parts[1](st, kctx)
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
    run(plan.fwd)
    plan.gw.fill_(1.0)
    plan._gw_key = None
    ztab, zst, znt, znb = plan._zero_tab
    plan.e.lib.gdrn_zero_multi(ztab.data_ptr(), zst.data_ptr(), znt, znb, st)
    run(plan.bwd)
    torch.cuda.synchronize()
    # dominant kernel = the conv instantiation with the largest share of this step's conv time
    per_k = {}
    for a, b, m in allev:
        per_k[m["kernel"]] = per_k.get(m["kernel"], 0.0) + a.elapsed_time(b)
    dom = max(per_k, key=per_k.get)
    ev = [x for x in allev if x[2]["kernel"] == dom]
    if os.environ.get("GDRN_LAYER_TABLE"):
        rows = sorted(((max(a.elapsed_time(b) - ovh_ms, 1e-3) * 1e3, m) for a, b, m in allev), key=lambda r: -r[0])  # event overhead subtracted
        with open(os.environ["GDRN_LAYER_TABLE"], "w") as f:
            for us, m in rows:
                f.write("%9.1f us %8.1f TF  %-40s %s\n" % (us, m["flops"] / us / 1e6, m["kernel"], m["layer"]))
    tot_ms = sum(max(a.elapsed_time(b) - ovh_ms, 1e-6) for a, b, _ in ev)
    flops = sum(m["flops"] for _, _, m in ev)
    achieved = flops / (tot_ms * 1e-3) / 1e12
    peak = (PEAK_BF16 if dtype == "bf16" else PEAK_F32) / 1e12
    # HBM-side bytes per launch, MFMA utilisation and HBM GB/s of that kernel: rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE, SQ/GRBM in
    # separate runs; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950) cannot be collected from inside this process.  The
    # committed summary (tools/pmc_util.py, tools/gpu_runs/r3_util.sh) carries the hash of the kernel sources it was measured on: any
    # other build reports null
    traffic, traffic_src, pmc = None, None, {}
    import glob

    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_mfma_util_hbm_bs64_bf16.json")), reverse=True)   # newest round first
    if dtype == "bf16" and plan.B == 64 and cands:
        traffic_src = "null: the committed PMC summaries belong to other builds of the kernels"
        for tpath in cands:
            with open(tpath) as f:
                tj = json.load(f)
            if tj.get("__kernel_source_sha256_16__") == kernel_source_hash():
                pmc = tj
                traffic = tj.get(dom, {}).get("traffic_bytes_per_launch")
                traffic = round(traffic) if traffic else None
                traffic_src = ("profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE passes of this build)"
                               % os.path.basename(tpath))
                break
    # every conv / GEMM instantiation of the step, largest first (the dominant kernel above is row 0): the operand-transform variants of
    # the halo kernel are separate instantiations, and the per-launch figure of one averages over feature maps from 8x8 to 64x64
    table = []
    for kname in sorted(per_k, key=per_k.get, reverse=True)[:8]:
        kev = [x for x in allev if x[2]["kernel"] == kname]
        kms = sum(max(a.elapsed_time(b) - ovh_ms, 1e-6) for a, b, _ in kev)
        kfl = sum(m["flops"] for _, _, m in kev)
        row = {"kernel": kname, "launches_per_step": len(kev), "ms_per_step": round(kms, 3), "achieved": round(kfl / (kms * 1e-3) / 1e12, 1),
               "frac": round(kfl / (kms * 1e-3) / 1e12 / peak, 3)}
        pk = pmc.get(kname.replace("_kernel<", "_kernel<").replace(" ", ""), {})
        if pk:   # rocprofv3-reported: MFMA-pipe busy cycles / (SIMDs x active cycles), HBM-side GB/s (tools/pmc_util.py)
            row["mfma_util"] = round(pk["mfma_util"], 3) if "mfma_util" in pk else None
            row["hbm_gbps"] = round(pk["hbm_gbps"])
        table.append(row)
    best = max(allev, key=lambda x: x[2]["flops"] / max(x[0].elapsed_time(x[1]) - ovh_ms, 1e-6))
    # the same kernel INSIDE the step as the timed region runs it (two streams: bucket-end work shares the CUs with the gradient chain):
    # events on the stream each launch goes to, over three real backward passes; this is the duration rocprofv3 --kernel-trace --stats
    # of the default command reports for it
    in_step = None
    if getattr(plan.e, "wgrad_stream", False):
        plan.op_events = sink = []
        for _ in range(3):
            plan.run_forward(kctx)
            plan.run_backward(kctx)
        torch.cuda.synchronize()
        plan.op_events = None
        dev_ = [(a.elapsed_time(b) - ovh_ms, m) for a, b, m, _ in sink if m["kernel"] == dom]
        if dev_:
            ms_ = sum(max(t, 1e-6) for t, _ in dev_)
            tf_ = sum(m["flops"] for _, m in dev_) / (ms_ * 1e-3) / 1e12
            in_step = {"avg_launch_us": round(ms_ * 1e3 / len(dev_), 2), "achieved": round(tf_, 1), "frac": round(tf_ / peak, 3),
                       "note": "the launch as the timed step runs it: on the side stream at one workgroup per CU, sharing the CUs with the chain kernels"}
    # the headline figures (avg_launch_us / achieved / frac) are those of the configuration the TIMED REGION runs -- the launch on
    # its own stream while the gradient chain shares the CUs with it (in_step, what rocprofv3 --kernel-trace --stats of the default command
    # reports for the kernel) -- and the stand-alone bracket of the same launches is kept as `isolated`
    nl = max(len(ev), 1)
    iso = {"avg_launch_us": round(tot_ms * 1e3 / nl, 2), "achieved": round(achieved, 2), "frac": round(achieved / peak, 4),
           "bracket": "every launch of the step issued ALONE on one stream, in the step's own launch configuration, between two HIP events"}
    head = in_step if in_step is not None else iso
    abytes = sum(m.get("bytes", 0.0) for _, _, m in ev) / nl
    return {"bound": "mfma", "kernel": dom, "launches_per_step": len(ev), "avg_launch_us": head["avg_launch_us"],
            "algorithmic_gflop_per_launch": round(flops / nl / 1e9, 3), "algorithmic_bytes_per_launch": round(abytes),
            "achieved": head["achieved"], "peak": peak,
            "unit": "TFLOP/s", "frac": head["frac"], "isolated": iso, "traffic": traffic, "traffic_source": traffic_src,
            "mfma_util": (round(pmc[dom]["mfma_util"], 3) if dom in pmc and "mfma_util" in pmc[dom] else None),
            "hbm_gbps": (round(pmc[dom]["hbm_gbps"]) if dom in pmc else None),
            "event_overhead_us": round(ovh_ms * 1e3, 2),
            "bracket": ("HIP events on the stream each launch goes to, inside three real two-stream backward passes of the timed configuration"
                        if in_step is not None else iso["bracket"]),
            "conv_kernels": table,
            "best_launch": {"kernel": best[2]["kernel"], "layer": best[2]["layer"],
                            "achieved": round(best[2]["flops"] / ((best[0].elapsed_time(best[1]) - ovh_ms) * 1e-3) / 1e12, 1)}}


def fp32_seed_probe(m32, dev):
    """the fp32 (parity) engine on the THREE seeded bs = 64 batches of golden G11 (tests/golden/g11_baseline_sizes.npz: the
    reference's own GDRN.forward(do_loss=True) at this size, in fp32 and evaluated in fp64) -- worst-of-seeds pose error against the reference's fp32
    outputs, beside the reference's own fp32-vs-fp64 distance on each batch (the floor two correct fp32 implementations cannot be told apart below:
    1.02e-4 in R on seed 2) and the engine's distance from that fp64 value.  Fixture data only; the oracle is not involved."""
    import numpy as np
    import torch

    from gdrnet_amd import synth

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "g11_baseline_sizes.npz")
    if not os.path.exists(path):
        return None
    g = np.load(path)
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    per, worst, floor, vs64 = {}, 0.0, {}, {}
    for seed in (1, 2, 3):
        tag = f"lm13_b64_s{seed}"
        m32.load_state_dict(synth.make_state_dict(0))
        m32.train()
        b = synth.make_batch(64, seed=seed)
        b = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
        with torch.no_grad():
            m32(b["roi_img"], **synth.model_kwargs(b, do_loss=True))
        pl = m32.engine().plan(64, True, True)
        fc = pl.fc_out.float().cpu()
        got = {"rot6d": fc[:, :6], "t_": fc[:, 6:9], "rot": pl.rot.float().cpu(), "trans": pl.trans.float().cpu()}
        e = {k: rel(got[k], torch.from_numpy(g[f"{tag}/{k}"])) for k in got}
        per[str(seed)] = {k: float("%.3e" % v) for k, v in e.items()}
        worst = max(worst, max(e.values()))
        floor[str(seed)] = float("%.3e" % rel(torch.from_numpy(g[f"{tag}/rot"]), torch.from_numpy(g[f"{tag}/f64/rot"])))
        vs64[str(seed)] = float("%.3e" % rel(got["rot"], torch.from_numpy(g[f"{tag}/f64/rot"])))
    m32.load_state_dict(synth.make_state_dict(0))
    return {"pose_rel_err_worst_of_seeds_1_2_3_vs_reference_g11": float("%.3e" % worst), "pose_rel_err_per_seed_vs_reference_g11": per,
            "R_reference_fp32_vs_its_own_fp64_per_seed": floor, "R_engine_vs_reference_fp64_per_seed": vs64,
            "worst_of_seeds_within_1e-4": worst < 1e-4,
            "rule": "rot6d / t_ / trans within 1e-4 of the reference on every seed; R within 1e-4 of it, or no farther from the reference's fp64 value than "
                    "1.5 x the reference's own fp32 path on that batch (tests/test_e2e_gpu.py::test_fp32_vs_the_reference_itself_at_baseline_sizes_g11)"}


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
    if args.cpu_threads_sweep:
        return cpu_threads_sweep()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)   # `python bench.py --gpus N`: this process becomes the launcher of the N ranks
    if args.dry:
        return dry_main(args)
    if args.backend != "nccl":
        _json_error(args, "--backend gloo only exists for --dry (the kernels need a GPU and RCCL)")
        return 2
    import torch
    import torch.distributed as dist

    from gdrnet_amd import GDRN, synth
    from gdrnet_amd import dist as gdist
    from gdrnet_amd.cfg import lm13_cfg
    from gdrnet_amd.engine import LOSS_NAMES

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            _json_error(args, f"--gpus {args.gpus} but the launcher started {world} ranks")
        return 2
    if not torch.cuda.is_available() or local >= torch.cuda.device_count():
        if rank == 0:
            _json_error(args, f"rank {rank}: no GPU {local} on this box ({torch.cuda.device_count() if torch.cuda.is_available() else 0} visible)")
        return 2
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    use_dist = world > 1 or args.dist_force
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29541"))
        if world > 1:
            os.environ.setdefault("GDRN_BUCKETS", "5")  # (the engine picks this itself once the group exists; explicit for clarity)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))

    cfg = lm13_cfg(device=dev)
    cfg.MODEL.CDPN.HIP_DTYPE = args.dtype
    model, opt = GDRN.build_model_optimizer(cfg)
    model.load_state_dict(synth.make_state_dict(0))
    B = args.bs
    batch = synth.make_batch(B, seed=1 + rank)  # each rank its own RoIs (weak scaling)
    batch = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    kw = synth.model_kwargs(batch, do_loss=not args.fwd_only)
    kw.pop("do_loss")
    if use_dist:
        gdist.broadcast_parameters(model)
        model.train()
        gdist.attach(model, force=args.dist_force, comm_dtype=args.comm_dtype)

    # parity of what is timed: the pose outputs of one train-mode forward of every arithmetic the line reports, on fresh
    # weights, kept on the host until the cpu_baseline leg has the oracle's outputs for the same batch
    want_parity = world == 1 and not args.fwd_only and not args.no_cpu_baseline and B in (4, 64)
    probes = {}

    def pose_probe(m):
        m.train()
        with torch.no_grad():
            m(batch["roi_img"], do_loss=True, **kw)
        pl = m.engine().plan(B, True, True)
        fc = pl.fc_out.float().cpu()
        return {"rot6d": fc[:, :6], "t_": fc[:, 6:9], "rot": pl.rot.float().cpu(), "trans": pl.trans.float().cpu()}

    if want_parity:
        cfgc = lm13_cfg(device=dev)
        cfgc.MODEL.CDPN.HIP_DTYPE = args.dtype
        mc, _ = GDRN.build_model_optimizer(cfgc)
        mc.load_state_dict(synth.conditioned_state_dict(0))
        probes["conditioned"] = pose_probe(mc)
        del mc
        torch.cuda.empty_cache()
        probes["plain"] = pose_probe(model)
        model.load_state_dict(synth.make_state_dict(0))   # (the probe's forward moved the BatchNorm running statistics)

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
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if use_dist:
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
        # sustained figure: the same full step for >= 2 s of timed GPU work (the driver's 20 steps are 0.18 s: DVFS has not settled)
        n_sus = max(50, int(2.2 / max(dt / args.steps, 1e-4)))
        t_sus = timed(step, n_sus)
        also["sustained"] = {"steps": n_sus, "seconds": round(t_sus * n_sus, 2), "ms_per_step": round(t_sus * 1e3, 3), "roi_s": round(B / t_sus, 1)}
        also.update(roi_cropper_extras(B, dev, timed))
        if args.dtype == "bf16":
            # the fp16 arithmetic mode (the same kernels built with IEEE half; what cfg.SOLVER.AMP.ENABLED / cfg.TEST.AMP_TEST select): the full step
            cfg16 = lm13_cfg(device=dev)
            cfg16.MODEL.CDPN.HIP_DTYPE = "fp16"
            m16, o16 = GDRN.build_model_optimizer(cfg16)
            m16.load_state_dict(synth.make_state_dict(0))
            m16.train()
            t16 = timed(lambda: m16.train_step(batch["roi_img"], optimizer=o16, **kw), n)
            also["fp16"] = {"roi_s": round(B / t16, 1), "ms_per_step": round(t16 * 1e3, 3), "loss_scale": m16.engine().loss_scale,
                            "vs_bf16_step": round(t16 / (dt / args.steps), 4)}
            del m16, o16
            torch.cuda.empty_cache()
            # the parity (fp32) mode -- the mode the 1e-4 pose bound is claimed for: generic fp32-MFMA kernels, no halo kernel
            cfg32 = lm13_cfg(device=dev)
            cfg32.MODEL.CDPN.HIP_DTYPE = "fp32"
            m32, o32 = GDRN.build_model_optimizer(cfg32)
            m32.load_state_dict(synth.make_state_dict(0))
            seeds32 = None
            if want_parity:
                probes["fp32"] = pose_probe(m32)
                m32.load_state_dict(synth.make_state_dict(0))
                if B == 64:
                    seeds32 = fp32_seed_probe(m32, dev)
            m32.train()
            t32 = timed(lambda: m32.train_step(batch["roi_img"], optimizer=o32, **kw), 5)
            also["fp32_parity_mode"] = {"roi_s": round(B / t32, 1), "ms_per_step": round(t32 * 1e3, 3),
                                        "whole_step_tflops": round(B / t32 * FLOP_PER_ROI_TRAIN / 1e12, 2),
                                        "frac_of_fp32_mfma_peak": round(B / t32 * FLOP_PER_ROI_TRAIN / PEAK_F32, 4)}
            if not args.no_roofline:
                eng32 = m32.engine()
                _, plan32, kctx32 = m32._prepare(batch["roi_img"], True, {k: kw.get(k) for k in (
                    "gt_xyz", "gt_mask_trunc", "gt_mask_visib", "gt_region", "gt_ego_rot", "gt_points", "sym_infos", "gt_trans", "gt_trans_ratio",
                    "roi_coord_2d", "roi_cams", "roi_centers", "roi_whs", "roi_extents", "resize_ratios")})
                r32 = measure_roofline(m32, plan32, kctx32, "fp32")
                also["fp32_parity_mode"]["roofline"] = {k: r32[k] for k in ("kernel", "launches_per_step", "avg_launch_us", "achieved", "peak", "frac")}
            if seeds32 is not None:
                also["fp32_parity_mode"].update(seeds32)
            del m32, o32
            torch.cuda.empty_cache()

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * B * args.steps / dt
        flop_roi = FLOP_PER_ROI_TRAIN if not args.fwd_only else 22.823e9
        res = {
            "metric": "RoI crops/sec (fwd+bwd) at 256\u00d7256 bs=64" if not args.fwd_only else "RoI crops/sec (fwd, inference) at 256\u00d7256 bs=64",
            "value": round(value, 2), "unit": "RoI/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "process_group": ({"backend": "nccl (RCCL)", "world_size": dist.get_world_size()} if use_dist else None),
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
        if also is not None and "fp32_parity_mode" in also:
            res["fp32_parity_mode"] = also.pop("fp32_parity_mode")   # a sibling of `roofline`: the arithmetic the 1e-4 pose bound belongs to
        if not args.no_cpu_baseline and world == 1:
            cb = cpu_baseline(parity_bs=B if want_parity else None)
            ref = cb.pop("__ref_pose__")
            res["cpu_baseline"] = cb
            if ref is not None and probes:
                def errs(got, want):
                    return {k: float("%.3e" % float((got[k].double() - want[k].double()).norm() / want[k].double().norm())) for k in want}

                e_plain, e_cond = errs(probes["plain"], ref["plain"]), errs(probes["conditioned"], ref["conditioned"])
                f32 = args.dtype == "fp32"
                res["parity"] = {
                    "dtype": args.dtype, "against": "oracle/ (fp32 CPU restatement of the reference path, pinned to the reference's outputs by tests/golden)",
                    "what": f"relative L2 error of the pose outputs of one train-mode forward at bs={B}, same seeded batch and weights as the timed step",
                    f"pose_rel_err_vs_oracle_bs{B}": e_plain, "worst": max(e_plain.values()),
                    "bound": 1e-4 if f32 else None, "within_north_star_1e-4": max(e_plain.values()) < 1e-4,
                    f"conditioned_network_bs{B}": e_cond,
                    "note": ("the mode the 1e-4 pose tolerance of BASELINE.json is claimed for" if f32 else
                             "16-bit storage of ~100 chained tensors is OUTSIDE the 1e-4 tolerance (fp32_parity_mode.pose_rel_err is the mode inside it); on "
                             "the random-init synthetic network a BatchNorm chain with batch statistics amplifies any rounding x700-1600 (the fp32 oracle "
                             "itself sits 4-6e-5 from fp64), so no bound is claimed for this figure; the bounded checks of this arithmetic are the "
                             "stage-by-stage test (tests/test_teacher_forced_gpu.py: every one of the 402 stages of this bs=64 step within 1e-3 of "
                             "the reference's arithmetic on the same stage inputs) and the conditioned network (near-identity residual blocks, x80)")}
                if "fp32" in probes and "fp32_parity_mode" in res:
                    e32 = errs(probes["fp32"], ref["plain"])
                    w3 = res["fp32_parity_mode"].get("pose_rel_err_worst_of_seeds_1_2_3_vs_reference_g11")
                    res["fp32_parity_mode"].update({f"pose_rel_err_vs_oracle_bs{B}_seed1": e32, "pose_rel_err_seed1": max(e32.values()),
                                                    "pose_rel_err": max(max(e32.values()), w3 or 0.0), "bound": 1e-4,
                                                    "within_north_star_1e-4": max(max(e32.values()), w3 or 0.0) < 1e-4})
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
