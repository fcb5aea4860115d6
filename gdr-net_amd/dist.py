"""Data-parallel gradient exchange for the one-process-per-GPU path (RCCL over xGMI).

The reference wraps the model in DDP through Lightning Lite (core/gdrn_modeling/main_gdrn.py:134-142)
and reduces the logged losses with ``comm.reduce_dict`` (core/utils/my_comm.py:8).  Here the engine
owns ONE flat fp32 gradient buffer ordered by backward completion; ``attach`` makes the backward call
``all_reduce`` on each finished bucket (pnp | head | layer4 | layer3 | rest) on a side stream, so the
exchange overlaps with the remaining backward kernels.  xGMI is point-to-point: few large messages
(5 buckets, 140 MB fp32 / 70 MB bf16 in total) rather than DDP's 25 MB default buckets.  The mean over ranks is
not a pass over the buffer: the collective SUMs and the fused optimizer multiplies by 1/world while it reads the
gradients (``GradReducer.grad_scale`` -> ``Ranger.step(grad_scale=)``); callers that read ``.grad`` (autograd path,
stock optimizers) get the scaled buffer from ``finish()``.  Works with any torch.distributed backend
("nccl" == RCCL on ROCm; "gloo" for the CPU tests of the protocol).
"""
import os

import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, flat, bounds, world_size=None, group=None, average=True, force=False, comm_dtype="fp32", defer_scale=False):
        """force: issue the collectives even in a one-rank group (exercises the stream / event protocol on one GPU).
        comm_dtype "bf16": a bucket travels as bf16 (half the xGMI bytes; cast into a staging buffer, all-reduce, cast back
        into the fp32 master gradient buffer on the side stream) -- the optimizer state and update stay fp32.
        defer_scale: leave the 1/world factor to the consumer (``grad_scale``) instead of a multiply pass per bucket."""
        assert comm_dtype in ("fp32", "bf16"), comm_dtype
        self.flat, self.bounds, self.group, self.average, self.force = flat, bounds, group, average, force
        self.world = world_size or (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.cuda = flat.is_cuda
        if self.cuda:
            from .engine import make_stream

            self.stream = make_stream(flat.device, "low")
        else:
            self.stream = None
        self.comm_dtype = comm_dtype
        self.stage = torch.empty(flat.numel(), dtype=torch.bfloat16, device=flat.device) if comm_dtype == "bf16" else None
        self.defer_scale = bool(defer_scale)
        self.active = self.world > 1 or self.force
        self.trace = force   # record a timing event per bucket (tests / tools/bucket_timeline.py run with force=True)
        self.started = []  # per bucket: event recorded on the side stream when its collective was enqueued (trace mode only)

    @property
    def grad_scale(self):
        """factor the consumer of the reduced buffer still has to apply (1.0 unless defer_scale)."""
        return (1.0 / self.world) if (self.active and self.average and self.defer_scale) else 1.0

    def _exchange(self, lo, hi):
        view = self.flat[lo:hi]
        if self.stage is not None:
            st = self.stage[lo:hi]
            st.copy_(view)
            dist.all_reduce(st, op=dist.ReduceOp.SUM, group=self.group)
            view.copy_(st)
        else:
            dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
        if self.average and not self.defer_scale:
            view.mul_(1.0 / self.world)

    def on_bucket(self, i):
        """Called right after the kernels that complete bucket i were enqueued on the compute stream."""
        if not self.active:
            return
        lo, hi = self.bounds[i]
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.flat.device))
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ev)
                if self.trace:
                    if i == 0:
                        self.started = []
                    se = torch.cuda.Event(enable_timing=True)
                    se.record(self.stream)
                    self.started.append(se)
                self._exchange(lo, hi)
        else:
            self._exchange(lo, hi)

    def wait(self):
        """Make the compute stream wait for every outstanding bucket exchange."""
        if self.cuda and self.active:
            torch.cuda.current_stream(self.flat.device).wait_stream(self.stream)

    def finish(self):
        """wait() + apply a deferred 1/world factor to the whole buffer: for consumers that cannot take ``grad_scale``
        (autograd's .grad accumulation, stock torch optimizers)."""
        self.wait()
        gs = self.grad_scale
        if gs != 1.0:
            self.flat.mul_(gs)


def attach(model, group=None, average=True, force=False, comm_dtype="fp32"):
    """Overlap the gradient all-reduce with the model's backward; returns the GradReducer.  The 1/world factor is deferred
    to the consumer: ``GDRN.train_step`` hands it to the fused Ranger step, the autograd path applies it in ``finish()``."""
    eng = model.engine(model.hip_dtype)   # the TRAINING engine's gradient buffer whatever mode the model is in (cfg.TEST.AMP_TEST keeps a second, eval-only engine)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if not eng.buckets_from_env:
        # an engine built before init_process_group picked the one-GPU layout (4 buckets): data-parallel runs want the 5-bucket one
        # (the 52 MB layer4 exchange starts nine blocks before the end of backward); plans are rebuilt on a change
        eng.set_bucket_layout(5 if world > 1 else 4)
    red = GradReducer(eng.grad_flat, eng.bucket_bounds, group=group, average=average, force=force, comm_dtype=comm_dtype, defer_scale=True)
    model._on_bucket = red.on_bucket
    model._reducer = red
    return red


def broadcast_parameters(model, src=0, group=None):
    """Initial parameter / buffer broadcast from rank 0 (what DDP does at wrap time): ONE collective over a flat copy
    instead of one per tensor, written back through ``copy_`` so that the tensors' version counters move (the engine's
    operand repack and eval-mode BatchNorm folding key on them) and the packed operands are rebuilt."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    ts = list(model.parameters()) + list(model.buffers())
    with torch.no_grad():
        by_dtype = {}
        for t in ts:
            by_dtype.setdefault(t.dtype, []).append(t)
        for dt_, lst in by_dtype.items():
            flat = torch.cat([t.detach().reshape(-1) for t in lst])
            dist.broadcast(flat, src=src, group=group)
            off = 0
            for t in lst:
                n = t.numel()
                t.copy_(flat[off:off + n].view_as(t))
                off += n
    for _, eng in getattr(model, "_engs", {}).values():  # engines that already packed the pre-broadcast weights
        eng.repack(force=True)
        eng.bn_epoch += 1


def reduce_loss_dict(loss_dict, group=None):
    """comm.reduce_dict(loss_dict) of core/utils/my_comm.py:8 (mean over ranks), one collective."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return loss_dict
    keys = sorted(loss_dict.keys())
    v = torch.stack([loss_dict[k].detach() for k in keys])
    dist.all_reduce(v, group=group)
    v /= dist.get_world_size(group)
    return {k: v[i] for i, k in enumerate(keys)}
