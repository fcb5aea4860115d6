"""Ranger (RAdam + gradient centralisation + Lookahead) with a fused HIP step.

Semantics of ``lib/torch_utils/solver/ranger.py:29-200`` (version 20.4.11): same constructor
arguments, same ``state`` keys (``step``, ``exp_avg``, ``exp_avg_sq``, ``slow_buffer``) so optimizer
checkpoints interchange.  The per-parameter Python arithmetic of the reference (about 12 ATen
launches per tensor x 148 tensors) is one ``gdrn_ranger_step`` launch per tensor here; the RAdam
rectification scalars (ranger.py:154-186) are computed on the host exactly as the reference does.
"""
import ctypes
import math

import torch
from torch.optim.optimizer import Optimizer

from . import cabi


def _bump_version(t):
    try:
        torch.autograd.graph.increment_version(t)
    except AttributeError:  # older torch
        t.add_(0)


def _global_pre_hooks():
    try:
        from torch.optim.optimizer import _global_optimizer_pre_hooks
        return _global_optimizer_pre_hooks.values()
    except ImportError:   # private name: absent -> no global hooks to fire
        return ()


def _global_post_hooks():
    try:
        from torch.optim.optimizer import _global_optimizer_post_hooks
        return _global_optimizer_post_hooks.values()
    except ImportError:
        return ()


def radam_step_size(step, beta1, beta2, n_sma_threshold):
    """(N_sma, step_size) of ranger.py:160-186."""
    beta2_t = beta2 ** step
    n_sma_max = 2 / (1 - beta2) - 1
    n_sma = n_sma_max - 2 * step * beta2_t / (1 - beta2_t)
    if n_sma > n_sma_threshold:
        step_size = math.sqrt(
            (1 - beta2_t) * (n_sma - 4) / (n_sma_max - 4) * (n_sma - 2) / n_sma * n_sma_max / (n_sma_max - 2)
        ) / (1 - beta1 ** step)
    else:
        step_size = 1.0 / (1 - beta1 ** step)
    return n_sma, step_size


_TASK_WORDS = ctypes.sizeof(cabi.RangerTask) // 4  # gdrn_ranger_task as 32-bit words
_LR_WORD = cabi.RangerTask.lr.offset // 4


class Ranger(Optimizer):
    takes_grad_scale = True  # step(grad_scale=): GDRN.train_step folds the 1/world of a SUM all-reduce into the kernel

    def __init__(self, params, lr=1e-3, alpha=0.5, k=6, N_sma_threshhold=5, betas=(0.95, 0.999), eps=1e-5, weight_decay=0,
                 use_gc=True, gc_conv_only=False):
        if not 0.0 <= alpha <= 1.0:
            raise ValueError(f"Invalid slow update rate: {alpha}")
        if not 1 <= k:
            raise ValueError(f"Invalid lookahead steps: {k}")
        if not lr > 0:
            raise ValueError(f"Invalid Learning Rate: {lr}")
        if not eps > 0:
            raise ValueError(f"Invalid eps: {eps}")
        defaults = dict(lr=lr, alpha=alpha, k=k, step_counter=0, betas=betas, N_sma_threshhold=N_sma_threshhold, eps=eps,
                        weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.N_sma_threshhold = N_sma_threshhold
        self.alpha = alpha
        self.k = k
        self.use_gc = use_gc
        self.gc_gradient_threshold = 3 if gc_conv_only else 1

    def state_dict(self):
        self.sync_dyn()   # (step counters of a run under a device-resident loss-scale state: step_dyn)
        return super().state_dict()

    def load_state_dict(self, state_dict):
        self._dyn = None  # the next step_dyn re-bases the device's step index on the loaded counters
        super().load_state_dict(state_dict)
        # (an aborted per-bucket step -- step_buckets_abort -- left some buckets' PARAMETERS one step ahead too: reloading the optimizer alone does
        #  not make the pair consistent, so the flag stays until the caller says both were restored: reset_after_abort())
        self.__dict__.pop("_multi_cache", None)   # the task tables hold the old state tensors' addresses

    def _init_state(self, p):
        state = self.state[p]
        if len(state) == 0:
            state["step"] = 0
            state["exp_avg"] = torch.zeros_like(p)
            state["exp_avg_sq"] = torch.zeros_like(p)
            state["slow_buffer"] = p.detach().clone()
        return state

    def _gc_shape(self, p, g):
        gc = 1 if (self.use_gc and g.dim() > self.gc_gradient_threshold) else 0
        rows = p.shape[0] if (gc and p.dim() > 1) else 1
        return gc, rows, p.numel() // rows

    def _multi_table(self, gi, items, lr):
        """device task table of one param group for gdrn_ranger_multi, cached on the buffers' addresses."""
        key = tuple((p.data_ptr(), g.data_ptr(), self.state[p]["exp_avg"].data_ptr(), self.state[p]["slow_buffer"].data_ptr()) for p, g in items)
        cache = self.__dict__.setdefault("_multi_cache", {})
        hit = cache.get(gi)
        if hit is not None and hit[0] == key:
            if hit[5] != lr:  # an LR schedule moved the rate: patch the `lr` member of every task on the device (one async fill)
                hit[1].view(torch.float32)[_LR_WORD::_TASK_WORDS].fill_(lr)
                cache[gi] = hit[:5] + (lr,)
            return hit[1:5]
        tasks, starts = [], [0]
        for p, g in items:
            st = self.state[p]
            gc, rows, cols = self._gc_shape(p, g)
            tasks.append(cabi.RangerTask(p=p.data_ptr(), g=g.data_ptr(), m=st["exp_avg"].data_ptr(), v=st["exp_avg_sq"].data_ptr(),
                                         slow=st["slow_buffer"].data_ptr(), rows=rows, cols=cols, gc=gc, lr=lr))
            starts.append(starts[-1] + rows)
        dev = items[0][0].device
        tab = cabi.to_device_table(tasks, dev)
        stt = torch.tensor(starts, dtype=torch.int32, device=dev)
        cache[gi] = (key, tab, stt, len(tasks), starts[-1], lr)
        return cache[gi][1:5]

    # ---- per-bucket stepping: the fused train step (GDRN.train_step) launches the update of a gradient bucket on the side stream as soon
    # as that bucket's gradients are complete, under the rest of the backward pass, instead of three launches behind it
    @torch.no_grad()
    def step_buckets_begin(self, grads, bucket_of, nbuckets, grad_scale=1.0):
        """grads: dict param -> fp32 gradient; bucket_of(param) -> bucket index.  Prepares one gdrn_ranger_multi launch per (param group,
        bucket) for the NEXT step index; the step counters themselves advance in step_buckets_end.  False: the per-bucket path does not
        apply (mixed step counts, CPU parameters, ...) and nothing was changed -- call step()."""
        self._leave_dyn()
        plan = []
        for gi, group in enumerate(self.param_groups):
            items = [(p, grads[p].detach()) for p in group["params"] if grads.get(p) is not None]
            if not items:
                continue
            if any(p.device.type != "cuda" or p.dtype != torch.float32 or g.dtype != torch.float32 or not g.is_contiguous() for p, g in items):
                return False
            states = [self._init_state(p) for p, _ in items]
            if len({s["step"] for s in states}) != 1:
                return False
            plan.append((gi, group, items, states))
        lib = cabi.load()
        self._check_not_poisoned()
        self._bucket_launches = [[] for _ in range(nbuckets)]
        self._bucket_done = 0
        self._bucket_params = []
        self._bucket_states = []   # their step counters advance in step_buckets_end: an exception inside the backward pass leaves them alone
        for hook in list(_global_pre_hooks()) + list(getattr(self, "_optimizer_step_pre_hooks", {}).values()):
            hook(self, (), {})     # the hooks torch fires around optimizer.step()
        for gi, group, items, states in plan:
            self._bucket_states += states
            step = states[0]["step"] + 1
            beta1, beta2 = group["betas"]
            n_sma, step_size = radam_step_size(step, beta1, beta2, self.N_sma_threshhold)
            by_b = {}
            for it in items:
                by_b.setdefault(bucket_of(it[0]), []).append(it)
            for b, its in by_b.items():
                tab, stt, nt, nrows = self._multi_table((gi, b), its, float(group["lr"]))
                self._bucket_launches[b].append((tab, stt, nt, nrows, beta1, beta2, group["eps"], group["weight_decay"], step_size,
                                                 1 if n_sma > self.N_sma_threshhold else 0, 1 if step % group["k"] == 0 else 0, float(grad_scale)))
            self._bucket_params += [p for p, _ in items]
        self._bucket_lib = lib
        return True

    def step_bucket(self, b):
        """launch the updates of bucket b on the current stream"""
        st = torch.cuda.current_stream().cuda_stream
        self._bucket_done += 1
        for (tab, stt, nt, nrows, beta1, beta2, eps, wd, step_size, adaptive, lookahead, gs) in self._bucket_launches[b]:
            cabi.check(self._bucket_lib.gdrn_ranger_multi(tab.data_ptr(), stt.data_ptr(), nt, nrows, beta1, beta2, eps, wd, step_size, adaptive, lookahead,
                                                          self.alpha, gs, st), "ranger_multi")

    def step_buckets_end(self):
        """every bucket's update is enqueued: commit the step counters, tell torch an optimizer step happened (LR schedulers check
        ``_opt_called``; step post-hooks fire as for ``step()``)"""
        for s in self._bucket_states:
            s["step"] += 1
        for p in self._bucket_params:
            _bump_version(p)  # updated in place behind autograd's back: tell repack() the weights changed
        self._bucket_launches, self._bucket_params, self._bucket_states = [], [], []
        self._opt_called = True
        for hook in list(getattr(self, "_optimizer_step_post_hooks", {}).values()) + list(_global_post_hooks()):
            hook(self, (), {})

    def reset_after_abort(self):
        """the caller has restored BOTH the model's parameters and this optimizer's state (load_state_dict on each) after an aborted per-bucket
        step: stepping is allowed again"""
        self._buckets_poisoned = False

    def _check_not_poisoned(self):
        if getattr(self, "_buckets_poisoned", False):
            raise cabi.GdrnHipError("a per-bucket optimizer step was aborted after some buckets had been updated: parameters and moments of those "
                                    "buckets are one step ahead of the rest -- reload the model AND the optimizer state, then optimizer.reset_after_abort()")

    def step_buckets_abort(self):
        """the backward pass raised before every bucket went out: forget the prepared launches; the step counters never moved.  If no bucket
        had been updated yet the optimizer is exactly where it was (a retry is a clean step).  Otherwise some buckets' parameters and moments
        are one step ahead of their counters and of the other buckets: the state is marked invalid and the next per-bucket step raises
        instead of re-applying them.  The step post-hooks fire either way: the pre-hooks of step_buckets_begin have a partner."""
        if getattr(self, "_bucket_done", 0) > 0:
            self._buckets_poisoned = True
        self._bucket_launches, self._bucket_params, self._bucket_states = [], [], []
        for hook in list(getattr(self, "_optimizer_step_post_hooks", {}).values()) + list(_global_post_hooks()):
            hook(self, (), {"aborted": True})   # (the partner of the pre-hook; the marker tells a step-counting hook that no step completed)

    # ---- fp16 arithmetic mode with the dynamic loss scale on the device (r6): no host read between the backward pass and the update
    @torch.no_grad()
    def step_dyn(self, grads, grad_scale, dyn):
        """One fused step under a device-resident loss-scale state `dyn` (engine.LossScaleState = gdrn_loss_scale_state): gdrn_ranger_multi_dyn
        turns into a no-op when the state's overflow flag is raised (GradScaler.step's skip, decided on the device), divides the gradients by the
        state's scale, and evaluates RAdam's step size / rectification / lookahead phase at base_step + applied + 1 -- the count of APPLIED
        steps, kept on the device.  The host's state["step"] counters are NOT advanced here: sync_dyn() (called by state_dict()) reads them back.
        The caller launches gdrn_loss_scale_update behind this.  False: the per-group multi-tensor path does not apply -- nothing was launched."""
        plan = []
        for gi, group in enumerate(self.param_groups):
            items = [(p, grads[p].detach()) for p in group["params"] if grads.get(p) is not None]
            if not items:
                continue
            if any(p.device.type != "cuda" or p.dtype != torch.float32 or g.dtype != torch.float32 or not g.is_contiguous() for p, g in items):
                return False
            states = [self._init_state(p) for p, _ in items]
            plan.append((gi, group, items, states))
        steps = {s["step"] for _, _, _, states in plan for s in states}
        if len(steps) != 1:
            return False
        if getattr(self, "_dyn", None) is not dyn:
            dyn.write(base_step=steps.pop())   # (re)base: device step index = these counters + steps applied from here on
            self._dyn = dyn
        lib = cabi.load()
        st = torch.cuda.current_stream().cuda_stream
        for hook in list(_global_pre_hooks()) + list(getattr(self, "_optimizer_step_pre_hooks", {}).values()):
            hook(self, (), {})
        for gi, group, items, states in plan:
            beta1, beta2 = group["betas"]
            tab, stt, nt, nrows = self._multi_table(("dyn", gi), items, float(group["lr"]))
            cabi.check(lib.gdrn_ranger_multi_dyn(tab.data_ptr(), stt.data_ptr(), nt, nrows, beta1, beta2, group["eps"], group["weight_decay"],
                                                 int(self.N_sma_threshhold), int(group["k"]), self.alpha, float(grad_scale), dyn.ptr, st), "ranger_multi_dyn")
            for p, _ in items:
                _bump_version(p)   # (possibly) updated in place behind autograd's back
        self._opt_called = True
        for hook in list(getattr(self, "_optimizer_step_post_hooks", {}).values()) + list(_global_post_hooks()):
            hook(self, (), {})
        return True

    def sync_dyn(self):
        """host step counters <- the device's count of applied steps (a host synchronisation; no-op without step_dyn)"""
        dyn = getattr(self, "_dyn", None)
        if dyn is None:
            return
        r = dyn.read()
        step = r["base_step"] + r["applied"]
        for group in self.param_groups:
            for p in group["params"]:
                if p in self.state and "step" in self.state[p]:
                    self.state[p]["step"] = step

    def _leave_dyn(self):
        """a step outside step_dyn follows: bring the host counters up to date and forget the device's base"""
        if getattr(self, "_dyn", None) is not None:
            self.sync_dyn()
            self._dyn = None

    @torch.no_grad()
    def step(self, closure=None, grads=None, grad_scale=1.0):
        """grads: optional dict param -> fp32 gradient tensor (used by the fused train step to read the
        engine's flat gradient buffer directly instead of ``p.grad``).  One multi-tensor launch per param group
        when every tensor of the group is at the same step count (the normal case); per-tensor launches otherwise.
        grad_scale: factor applied to every gradient inside the kernel (1/world_size after a SUM all-reduce)."""
        self._check_not_poisoned()   # (a plain step on top of a half-applied per-bucket step would train on inconsistent buckets: ADVICE r5)
        self._leave_dyn()
        lib = cabi.load()
        for gi, group in enumerate(self.param_groups):
            beta1, beta2 = group["betas"]
            items = []
            for p in group["params"]:
                g = grads.get(p) if grads is not None else p.grad
                if g is None:
                    continue
                if p.device.type != "cuda" or p.dtype != torch.float32:
                    raise cabi.GdrnHipError("fused Ranger needs fp32 parameters on the GPU")
                g = g.detach()
                if g.dtype != torch.float32 or not g.is_contiguous():
                    g = g.float().contiguous()
                items.append((p, g))
            if not items:
                continue
            states = [self._init_state(p) for p, _ in items]
            for s in states:
                s["step"] += 1
            st = torch.cuda.current_stream(items[0][0].device).cuda_stream
            steps = {s["step"] for s in states}
            if len(steps) == 1 and len(items) > 1:
                step = steps.pop()
                n_sma, step_size = radam_step_size(step, beta1, beta2, self.N_sma_threshhold)
                tab, stt, nt, nrows = self._multi_table(gi, items, float(group["lr"]))
                self._keep = [g for _, g in items]  # temporaries (non-fp32 grads) must outlive the launch
                cabi.check(
                    lib.gdrn_ranger_multi(tab.data_ptr(), stt.data_ptr(), nt, nrows, beta1, beta2, group["eps"], group["weight_decay"],
                                          step_size, 1 if n_sma > self.N_sma_threshhold else 0, 1 if step % group["k"] == 0 else 0,
                                          self.alpha, float(grad_scale), st),
                    "ranger_multi",
                )
            else:
                for (p, g), state in zip(items, states):
                    if grad_scale != 1.0:
                        g = g * float(grad_scale)
                    step = state["step"]
                    n_sma, step_size = radam_step_size(step, beta1, beta2, self.N_sma_threshhold)
                    gc, rows, cols = self._gc_shape(p, g)
                    cabi.check(
                        lib.gdrn_ranger_step(p.data_ptr(), g.data_ptr(), state["exp_avg"].data_ptr(), state["exp_avg_sq"].data_ptr(),
                                             state["slow_buffer"].data_ptr(), rows, cols, gc, group["lr"], beta1, beta2, group["eps"],
                                             group["weight_decay"], step_size, 1 if n_sma > self.N_sma_threshhold else 0,
                                             1 if step % group["k"] == 0 else 0, self.alpha, st),
                        "ranger_step",
                    )
            for p, _ in items:
                _bump_version(p)  # updated in place behind autograd's back: tell repack() the weights changed
        return None
