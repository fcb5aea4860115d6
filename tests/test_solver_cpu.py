"""LR schedules (lr_scheduler.py:137-263, solver_utils.py:100-140) against golden G9 = the
learning rates the reference's own schedulers wrote during a 2000-iteration run.  Exact match (Python float arithmetic in
the reference's operation order)."""
import os

import numpy as np
import pytest
import torch

from gdrnet_amd import solver
from gdrnet_amd.cfg import lm13_cfg

HERE = os.path.dirname(os.path.abspath(__file__))
TOTAL = 2000


def _run(make):
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1e-4)
    sch = make(opt)
    lrs = []
    for _ in range(TOTAL):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sch.step()
    return np.array(lrs, np.float64)


@pytest.fixture(scope="module")
def g9():
    return np.load(os.path.join(HERE, "golden", "g9_lr_schedules.npz"))


@pytest.mark.parametrize("method", ["cosine", "linear", "poly", "exp", "step", "none"])
def test_flat_and_anneal_matches_the_reference(g9, method):
    got = _run(lambda o: solver.flat_and_anneal_lr_scheduler(
        o, total_iters=TOTAL, warmup_iters=100, warmup_factor=0.001, warmup_method="linear", anneal_point=0.72, anneal_method=method,
        target_lr_factor=0.05 if method != "cosine" else 0, poly_power=0.9, step_gamma=0.1, steps=[0.5, 0.75]))
    assert np.array_equal(got, g9["flat_" + method])


def test_constant_warmup_and_multistep_match_the_reference(g9):
    got = _run(lambda o: solver.flat_and_anneal_lr_scheduler(o, total_iters=TOTAL, warmup_iters=50, warmup_factor=0.1, warmup_method="constant",
                                                             anneal_point=0.5, anneal_method="cosine"))
    assert np.array_equal(got, g9["flat_cosine_constwarm"])
    got = _run(lambda o: solver.WarmupMultiStepLR(o, [1000.0, 1500.0], 0.1, warmup_factor=0.001, warmup_iters=100, warmup_method="linear"))
    assert np.array_equal(got, g9["multistep"])


def test_build_lr_scheduler_from_the_lm_config(g9):
    cfg = lm13_cfg(device="cpu")
    cfg.SOLVER.WARMUP_ITERS = 100
    got = _run(lambda o: solver.build_lr_scheduler(cfg, o, TOTAL))  # flat_and_anneal / cosine / 0.72 (a6_cPnP_lm13.py:22-32)
    assert np.array_equal(got, g9["flat_cosine"])
    cfg.SOLVER.LR_SCHEDULER_NAME = "WarmupMultiStepLR"
    got = _run(lambda o: solver.build_lr_scheduler(cfg, o, TOTAL))  # REL_STEPS (0.5, 0.75) -> milestones 1000, 1500
    assert np.array_equal(got, g9["multistep"])
    cfg.SOLVER.LR_SCHEDULER_NAME = "WarmupCosineLR"
    got = _run(lambda o: solver.build_lr_scheduler(cfg, o, TOTAL))
    assert got[0] == 1e-4 * 0.001 and abs(got[1000] - 0.5e-4) < 1e-12 and got[-1] < 1e-9
    cfg.SOLVER.LR_SCHEDULER_NAME = "nope"
    with pytest.raises(ValueError):
        solver.build_lr_scheduler(cfg, torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0), TOTAL)


def test_argument_errors_like_the_reference():
    with pytest.raises(ValueError):
        solver.flat_and_anneal_factor(100, warmup_method="cubic")
    with pytest.raises(ValueError):
        solver.flat_and_anneal_factor(100, anneal_method="sqrt")
    with pytest.raises(ValueError):
        solver.flat_and_anneal_factor(100, anneal_point=1.5)
    with pytest.raises(ValueError):
        solver.flat_and_anneal_factor(100, anneal_method="step", steps=[0.8, 0.5])
    with pytest.raises(ValueError):
        solver.flat_and_anneal_factor(100, warmup_iters=50, anneal_method="step", steps=[0.2, 0.9])
