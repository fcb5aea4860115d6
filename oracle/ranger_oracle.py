"""ORACLE -- test infrastructure, NOT product code (see gdrn_oracle.py for the rules).

CPU restatement of the reference optimizer step, ``Ranger.step`` (lib/torch_utils/solver/ranger.py:100-200, version 20.4.11:
RAdam + gradient centralisation + Lookahead), as a function over explicit state dicts, so that bench.py's ``cpu_baseline`` can time
the reference's full training step (forward + backward + optimizer) on the host cores and tests can pin the fused HIP optimizer.
Pinned by golden G6 (7 steps of the reference's own class; tests/test_oracle_golden.py::test_ranger_oracle).
"""
import math

import torch


def radam_step_size(step, beta1, beta2, n_sma_threshold=5):
    """rectification term of ranger.py:160-186."""
    beta2_t = beta2 ** step
    n_sma_max = 2 / (1 - beta2) - 1
    n_sma = n_sma_max - 2 * step * beta2_t / (1 - beta2_t)
    if n_sma > n_sma_threshold:
        step_size = math.sqrt((1 - beta2_t) * (n_sma - 4) / (n_sma_max - 4) * (n_sma - 2) / n_sma * n_sma_max / (n_sma_max - 2)) / (1 - beta1 ** step)
    else:
        step_size = 1.0 / (1 - beta1 ** step)
    return n_sma, step_size


@torch.no_grad()
def ranger_step(params, grads, state, lr=1e-3, alpha=0.5, k=6, n_sma_threshold=5, betas=(0.95, 0.999), eps=1e-5, weight_decay=0.0, use_gc=True):
    """params / grads: lists of fp32 tensors (params updated in place); state: list of dicts (filled on first use)."""
    beta1, beta2 = betas
    for p, g, st in zip(params, grads, state):
        if not st:  # ranger.py:125-136
            st.update(step=0, exp_avg=torch.zeros_like(p), exp_avg_sq=torch.zeros_like(p), slow_buffer=p.detach().clone())
        g = g.detach().clone()
        if use_gc and g.dim() > 1:  # gradient centralisation, ranger.py:144-145
            g.add_(-g.mean(dim=tuple(range(1, g.dim())), keepdim=True))
        st["step"] += 1
        st["exp_avg_sq"].mul_(beta2).addcmul_(g, g, value=1 - beta2)
        st["exp_avg"].mul_(beta1).add_(g, alpha=1 - beta1)
        n_sma, step_size = radam_step_size(st["step"], beta1, beta2, n_sma_threshold)
        if weight_decay != 0:
            p.add_(p, alpha=-weight_decay * lr)
        if n_sma > n_sma_threshold:
            p.addcdiv_(st["exp_avg"], st["exp_avg_sq"].sqrt().add_(eps), value=-step_size * lr)
        else:
            p.add_(st["exp_avg"], alpha=-step_size * lr)
        if st["step"] % k == 0:  # lookahead, ranger.py:192-198
            st["slow_buffer"].add_(p - st["slow_buffer"], alpha=alpha)
            p.copy_(st["slow_buffer"])
