"""Data-parallel gradient exchange for the one-process-per-GPU path (RCCL over xGMI).

The reference wraps the model in DDP through Lightning Lite (core/gdrn_modeling/main_gdrn.py:134-142)
and reduces the logged losses with ``comm.reduce_dict`` (core/utils/my_comm.py:8).  Here the engine
owns ONE flat fp32 gradient buffer ordered by backward completion; ``attach`` makes the backward call
``all_reduce`` on each finished bucket (pnp | head | layer4+3 | rest) on a side stream, so the
exchange overlaps with the remaining backward kernels.  xGMI is point-to-point: few large messages
(4 buckets, 140 MB total) rather than DDP's 25 MB default buckets.  Works with any
torch.distributed backend ("nccl" == RCCL on ROCm; "gloo" for the CPU tests of the protocol).
"""
import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, flat, bounds, world_size=None, group=None, average=True, force=False):
        """force: issue the collectives even in a one-rank group (exercises the stream / event protocol on one GPU)."""
        self.flat, self.bounds, self.group, self.average, self.force = flat, bounds, group, average, force
        self.world = world_size or (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.cuda = flat.is_cuda
        self.stream = torch.cuda.Stream(device=flat.device) if self.cuda else None
        self.events = []

    def on_bucket(self, i):
        """Called right after the kernels that complete bucket i were enqueued on the compute stream."""
        if self.world == 1 and not self.force:
            return
        lo, hi = self.bounds[i]
        view = self.flat[lo:hi]
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.flat.device))
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ev)
                dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
                if self.average:
                    view.mul_(1.0 / self.world)
        else:
            dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
            if self.average:
                view.mul_(1.0 / self.world)

    def wait(self):
        """Make the compute stream wait for every outstanding bucket exchange."""
        if self.cuda and (self.world > 1 or self.force):
            torch.cuda.current_stream(self.flat.device).wait_stream(self.stream)


def attach(model, group=None, average=True, force=False):
    """Overlap the gradient all-reduce with the model's backward; returns the GradReducer."""
    eng = model.engine()
    red = GradReducer(eng.grad_flat, eng.bucket_bounds, group=group, average=average, force=force)
    model._on_bucket = red.on_bucket
    model._reducer = red
    return red


def broadcast_parameters(model, src=0, group=None):
    """Initial parameter / buffer broadcast from rank 0 (what DDP does at wrap time)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src=src, group=group)


def reduce_loss_dict(loss_dict, group=None):
    """comm.reduce_dict(loss_dict) of core/utils/my_comm.py:8 (mean over ranks), one collective."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return loss_dict
    keys = sorted(loss_dict.keys())
    v = torch.stack([loss_dict[k].detach() for k in keys])
    dist.all_reduce(v, group=group)
    v /= dist.get_world_size(group)
    return {k: v[i] for i, k in enumerate(keys)}
