"""Learning-rate schedules of the reference trainer ("LR schedule lr_scheduler.py:177-263;
registration solver_utils.py:17-57") for the fused Ranger step.

Host-side mirror of

* ``flat_and_anneal_lr_scheduler``   lib/torch_utils/solver/lr_scheduler.py:177-263 (what every GDR-Net config uses:
  linear warm-up, flat, cosine anneal from ``ANNEAL_POINT``),
* ``WarmupMultiStepLR``              lib/torch_utils/solver/lr_scheduler.py:137-174,
* ``build_lr_scheduler(cfg, optimizer, total_iters)``   core/utils/solver_utils.py:100-140 (``WarmupCosineLR`` is detectron2's
  class: cosine to zero with the same warm-up factor, restated from its published definition).

The schedules are plain ``torch.optim.lr_scheduler`` objects that write ``param_group["lr"]`` once per iteration; the fused
optimizer (``ranger.Ranger``) patches the new value into its cached device task table with one tiny asynchronous fill instead of
rebuilding the table, so a schedule that changes the rate every step costs nothing on the host.  The factor arithmetic is in
Python floats in the reference's operation order: golden G9 (values produced by the reference's own functions) is matched
exactly.
"""
import math
from bisect import bisect_right

import torch

_ANNEAL = ("cosine", "linear", "poly", "exp", "step", "none")


def _warmup_factor(it, warmup_iters, warmup_factor, warmup_method):
    if warmup_method == "constant":
        return warmup_factor
    alpha = float(it) / warmup_iters
    return warmup_factor * (1 - alpha) + alpha


def flat_and_anneal_factor(total_iters, warmup_iters=0, warmup_factor=0.1, warmup_method="linear", anneal_point=0.72, anneal_method="cosine",
                           target_lr_factor=0, poly_power=1.0, step_gamma=0.1, steps=(2 / 3.0, 8 / 9.0)):
    """Returns f(iteration) -> multiplicative LR factor (lr_scheduler.py:177-261)."""
    if warmup_method not in ("constant", "linear"):
        raise ValueError("Only 'constant' or 'linear' warmup_method accepted," "got {}".format(warmup_method))
    if anneal_method not in _ANNEAL:
        raise ValueError("Only 'cosine', 'linear', 'poly', 'exp', 'step' or 'none' anneal_method accepted," "got {}".format(anneal_method))
    steps = list(steps)
    if anneal_method == "step":
        if any(s < warmup_iters / total_iters or s > 1 for s in steps):
            raise ValueError("error in steps: {}. warmup_iters: {} total_iters: {}." "steps should be in ({},1)".format(
                steps, warmup_iters, total_iters, warmup_iters / total_iters))
        if steps != sorted(steps):
            raise ValueError("steps {} is not in ascending order.".format(steps))
        anneal_start = steps[0] * total_iters  # anneal_point is ignored for the step method
    else:
        if anneal_point > 1 or anneal_point < 0:
            raise ValueError("anneal_point should be in [0,1], got {}".format(anneal_point))
        anneal_start = anneal_point * total_iters
    milestones = [s * total_iters for s in steps]

    def factor(x):
        if x < warmup_iters:
            return _warmup_factor(x, warmup_iters, warmup_factor, warmup_method)
        if x < anneal_start or anneal_method == "none":
            return 1
        span = total_iters - anneal_start
        if anneal_method == "step":
            return step_gamma ** bisect_right(milestones, float(x))
        if anneal_method == "cosine":
            return target_lr_factor + 0.5 * (1 - target_lr_factor) * (1 + math.cos(math.pi * ((float(x) - anneal_start) / span)))
        if anneal_method == "linear":
            return target_lr_factor + (1 - target_lr_factor) * (total_iters - float(x)) / span
        if anneal_method == "poly":
            return target_lr_factor + (1 - target_lr_factor) * ((total_iters - float(x)) / span) ** poly_power
        return max(target_lr_factor, 5e-3) ** ((float(x) - anneal_start) / span)  # "exp"; the floor keeps the rate off zero

    return factor


def flat_and_anneal_lr_scheduler(optimizer, total_iters, **kwargs):
    """Same signature and behaviour as the reference function: a ``LambdaLR`` over :func:`flat_and_anneal_factor`."""
    return torch.optim.lr_scheduler.LambdaLR(optimizer, flat_and_anneal_factor(total_iters, **kwargs))


class WarmupMultiStepLR(torch.optim.lr_scheduler.LRScheduler):
    """lr_scheduler.py:137-174: ``base_lr * warmup * gamma ** (#milestones passed)``."""

    def __init__(self, optimizer, milestones, gamma=0.1, warmup_factor=1.0 / 3, warmup_iters=5, warmup_method="linear", last_epoch=-1):
        if not list(milestones) == sorted(milestones):
            raise ValueError("Milestones should be a list of" " increasing integers. Got {}", milestones)
        if warmup_method not in ("constant", "linear"):
            raise ValueError("Only 'constant' or 'linear' warmup_method accepted" "got {}".format(warmup_method))
        self.milestones, self.gamma = list(milestones), gamma
        self.warmup_factor, self.warmup_iters, self.warmup_method = warmup_factor, warmup_iters, warmup_method
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        w = 1
        if self.last_epoch < self.warmup_iters:
            w = _warmup_factor(self.last_epoch, self.warmup_iters, self.warmup_factor, self.warmup_method)
        return [base_lr * w * self.gamma ** bisect_right(self.milestones, self.last_epoch) for base_lr in self.base_lrs]


class WarmupCosineLR(torch.optim.lr_scheduler.LRScheduler):
    """detectron2.solver.WarmupCosineLR (third-party; published definition): ``base_lr * warmup * 0.5 * (1 + cos(pi * it / max_iters))``."""

    def __init__(self, optimizer, max_iters, warmup_factor=0.001, warmup_iters=1000, warmup_method="linear", last_epoch=-1):
        if warmup_method not in ("constant", "linear"):
            raise ValueError("Unknown warmup method: {}".format(warmup_method))
        self.max_iters, self.warmup_factor, self.warmup_iters, self.warmup_method = max_iters, warmup_factor, warmup_iters, warmup_method
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        w = 1.0
        if self.last_epoch < self.warmup_iters:
            w = _warmup_factor(self.last_epoch, self.warmup_iters, self.warmup_factor, self.warmup_method)
        return [base_lr * w * 0.5 * (1.0 + math.cos(math.pi * self.last_epoch / self.max_iters)) for base_lr in self.base_lrs]


def build_lr_scheduler(cfg, optimizer, total_iters):
    """core/utils/solver_utils.py:100-140."""
    s = cfg.SOLVER
    name = s.LR_SCHEDULER_NAME
    if name == "WarmupMultiStepLR":
        return WarmupMultiStepLR(optimizer, [rel * total_iters for rel in s.REL_STEPS], s.GAMMA, warmup_factor=s.WARMUP_FACTOR,
                                 warmup_iters=s.WARMUP_ITERS, warmup_method=s.WARMUP_METHOD)
    if name == "WarmupCosineLR":
        return WarmupCosineLR(optimizer, total_iters, warmup_factor=s.WARMUP_FACTOR, warmup_iters=s.WARMUP_ITERS, warmup_method=s.WARMUP_METHOD)
    if name.lower() == "flat_and_anneal":
        return flat_and_anneal_lr_scheduler(
            optimizer, total_iters=total_iters, warmup_factor=s.WARMUP_FACTOR, warmup_iters=s.WARMUP_ITERS, warmup_method=s.WARMUP_METHOD,
            anneal_method=s.ANNEAL_METHOD, anneal_point=s.ANNEAL_POINT, steps=s.get("REL_STEPS", [2 / 3.0, 8 / 9.0]),
            target_lr_factor=s.get("TARTGET_LR_FACTOR", 0), poly_power=s.get("POLY_POWER", 1.0), step_gamma=s.GAMMA)
    raise ValueError("Unknown LR scheduler: {}".format(name))
