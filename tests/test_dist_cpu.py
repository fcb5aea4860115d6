"""N > 1 path on CPU: two gloo ranks exercise the bucketed gradient exchange (gdrnet_amd.dist.GradReducer), the
initial parameter broadcast and the loss reduction -- the protocol bench.py / a DDP-free trainer runs over RCCL on
the GPUs (reference: Lightning-Lite DDP wrap, core/gdrn_modeling/main_gdrn.py:134-142; comm.reduce_dict,
core/utils/my_comm.py:8)."""
import os
import socket

import pytest

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gdrnet_amd.dist import GradReducer, broadcast_parameters, reduce_loss_dict

    try:
        # flat gradient buffer with 4 buckets in backward-completion order
        n = 1000
        flat = torch.arange(n, dtype=torch.float32) * (rank + 1)
        bounds = [(0, 100), (100, 400), (400, 900), (900, 1000)]
        red = GradReducer(flat, bounds, average=True)
        assert red.world == world
        for i in range(len(bounds)):  # backward calls on_bucket(i) as each bucket's kernels are enqueued
            red.on_bucket(i)
        red.wait()
        expect = torch.arange(n, dtype=torch.float32) * (sum(r + 1 for r in range(world)) / world)
        assert torch.allclose(flat, expect), (rank, (flat - expect).abs().max())
        # bf16 on the wire, fp32 master buffer, 1/world deferred to the consumer (the fused optimizer's grad_scale)
        flat2 = torch.arange(n, dtype=torch.float32) * (rank + 1)
        red2 = GradReducer(flat2, bounds, average=True, comm_dtype="bf16", defer_scale=True)
        for i in range(len(bounds)):
            red2.on_bucket(i)
        red2.wait()
        assert red2.grad_scale == 1.0 / world
        assert torch.allclose(flat2 * red2.grad_scale, expect, rtol=1e-2), (rank, (flat2 * red2.grad_scale - expect).abs().max())
        red2.finish()
        assert torch.allclose(flat2, expect, rtol=1e-2)

        # initial broadcast of parameters and buffers from rank 0
        torch.manual_seed(rank)
        model = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.BatchNorm2d(4))
        broadcast_parameters(model)
        ref = [torch.zeros_like(p) for p in model.parameters()]
        for r, p in zip(ref, model.parameters()):
            r.copy_(p.data)
            dist.broadcast(r, src=0)
            assert torch.equal(r, p.data)

        # loss dict reduction = mean over ranks, one collective
        ld = {"loss_a": torch.tensor(float(rank)), "loss_b": torch.tensor(2.0 * rank + 1)}
        out = reduce_loss_dict(ld)
        assert abs(out["loss_a"].item() - (world - 1) / 2) < 1e-6 and abs(out["loss_b"].item() - world) < 1e-6
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def _worker_autograd(rank, world, port, q):
    """loss.backward() through GDRN's single autograd node with a reducer attached: the gradients autograd accumulates into
    .grad must be the all-reduced MEAN (the node waits for the bucket exchanges and applies the deferred 1/world before it hands
    its views to AccumulateGrad) -- the path the reference trainer takes (loss.backward(); optimizer.step(), engine.py:279).
    The HIP engine is replaced by a stub plan with the same interface (no GPU in this test)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from types import SimpleNamespace as NS

    from gdrnet_amd import GDRN as G
    from gdrnet_amd.dist import GradReducer

    try:
        for comm in ("fp32", "bf16"):
            params = {"a.weight": torch.nn.Parameter(torch.zeros(6, 5)), "b.bias": torch.nn.Parameter(torch.zeros(7))}
            names = list(params)
            flat = torch.zeros(40)
            offs = {"b.bias": 0, "a.weight": 8}  # backward-completion order: b first
            grads = {n: flat[offs[n]: offs[n] + params[n].numel()].view(params[n].shape) for n in names}
            eng = NS(grads=grads, param_names=names, grad_flat=flat, P=params)
            bounds = [(0, 8), (8, 40)]
            red = GradReducer(flat, bounds, average=True, comm_dtype=comm, defer_scale=True)
            model = NS(_on_bucket=red.on_bucket, _reducer=red)
            plan = NS(e=eng, has_backward=True, generation=0, losses=torch.ones(8), gw=torch.zeros(8))

            def run_forward(kctx):
                plan.generation += 1

            def run_backward(kctx, on_bucket=None):
                # "kernels": rank-dependent gradients scaled by the incoming loss gradient, bucket by bucket
                flat[0:8] = (rank + 1) * plan.gw.sum()
                on_bucket(0)
                flat[8:40] = torch.arange(32, dtype=torch.float32) * (rank + 1)
                on_bucket(1)

            plan.run_forward, plan.run_backward = run_forward, run_backward
            losses = G._PathFn.apply(model, plan, {}, *[params[n] for n in names])
            losses.sum().backward()
            mean_scale = sum(r + 1 for r in range(world)) / world
            assert torch.allclose(params["b.bias"].grad, torch.full((7,), 8.0 * mean_scale), rtol=1e-2 if comm == "bf16" else 1e-6)
            exp = (torch.arange(32, dtype=torch.float32) * mean_scale)[:30].view(6, 5)
            assert torch.allclose(params["a.weight"].grad, exp, rtol=1e-2 if comm == "bf16" else 1e-6), (comm, params["a.weight"].grad, exp)
            # a second forward before the backward of the first is refused (one set of activations per plan)
            l1 = G._PathFn.apply(model, plan, {}, *[params[n] for n in names])
            G._PathFn.apply(model, plan, {}, *[params[n] for n in names])
            try:
                l1.sum().backward()
                raise AssertionError("stale backward was not refused")
            except Exception as e:  # GdrnHipError
                assert "overwritten" in str(e), e
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback

        q.put((rank, repr(e) + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_autograd_path_waits_for_the_reducer_two_ranks_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_autograd, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_grad_reducer_two_ranks_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_single_process_reducer_is_a_noop():
    from gdrnet_amd.dist import GradReducer, reduce_loss_dict

    flat = torch.ones(10)
    red = GradReducer(flat, [(0, 10)], world_size=1)
    red.on_bucket(0)
    red.wait()
    assert torch.equal(flat, torch.ones(10))
    ld = {"a": torch.tensor(1.0)}
    assert reduce_loss_dict(ld) is ld


# ---------------------------------------------------------------------------------------------- the REAL plan's bucket protocol
def _worker_real_plan(rank, world, port, q):
    """Two gloo ranks drive the bucket marks of the real backward launch list (dry engine on host tensors, no launches): every
    element of the flat gradient buffer is all-reduced exactly once, the buckets fire in order 0..n-1 behind the last op that
    writes one of their gradients, and the mean lands in every parameter's gradient view."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gdrnet_amd import GDRN as G
        from gdrnet_amd.cfg import lm13_cfg
        from gdrnet_amd.dist import GradReducer
        from gdrnet_amd.engine import Engine

        torch.manual_seed(0)
        model, _ = G.build_model_optimizer(lm13_cfg(device="cpu"))
        eng = Engine(dict(model.named_parameters()), dict(model.named_buffers()), dtype="bf16", dry=True)
        assert len(eng.bucket_bounds) == 5, eng.bucket_bounds   # world > 1 -> the data-parallel layout
        plan = eng.plan(2, True, True)
        flat, bounds = eng.grad_flat, eng.bucket_bounds
        # the bounds partition the buffer
        assert bounds[0][0] == 0 and bounds[-1][1] == flat.numel() and all(bounds[i][1] == bounds[i + 1][0] for i in range(len(bounds) - 1))
        # every parameter's gradient is complete when its bucket fires: the backward group that writes it belongs to the same bucket or to
        # an EARLIER one (the BatchNorm-backward sums of a layer's last block come out of the next layer's stride-2 data gradient, one
        # bucket earlier); groups run in reverse forward order, buckets fire in order 0, 1, ...
        first = eng.bucket_first_group
        bucket_of_group = lambda gi: next(i for i, g0 in enumerate(first) if gi >= g0)
        bucket_of_off = lambda off: next(i for i, (lo, hi) in enumerate(bounds) if lo <= off < hi)
        assert set(plan.grad_group) == set(eng.param_names), set(eng.param_names) ^ set(plan.grad_group)
        for n in eng.param_names:
            assert bucket_of_group(plan.grad_group[n]) <= bucket_of_off(eng.grad_offsets[n]), n
        assert sum(bucket_of_group(plan.grad_group[n]) < bucket_of_off(eng.grad_offsets[n]) for n in eng.param_names) <= 8
        red = GradReducer(flat, bounds, average=True, defer_scale=True)
        seen = []

        def before(i):
            # "the kernels of bucket i have just finished": its gradients appear, later buckets are still untouched
            lo, hi = bounds[i]
            assert float(flat[hi:].abs().sum()) == 0.0, i
            flat[lo:hi] = (rank + 1) * (torch.arange(lo, hi) % 97 + 1).float()
            seen.append(i)

        flat.zero_()
        fired = plan.walk_backward(red.on_bucket, before)
        red.finish()
        assert seen == list(range(len(bounds))) and [b for _, b in fired] == seen
        ops = [i for i, _ in fired]
        assert ops == sorted(ops) and ops[-1] == len(plan.bwd) - 1      # the last bucket closes behind the last backward op
        mean = sum(r + 1 for r in range(world)) / world
        exp = mean * (torch.arange(flat.numel()) % 97 + 1).float()
        assert torch.allclose(flat, exp, rtol=1e-6), float((flat - exp).abs().max())   # reduced exactly once, mean applied once
        for n in eng.param_names:
            o = eng.grad_offsets[n]
            assert torch.equal(eng.grads[n].reshape(-1), flat[o:o + eng.P[n].numel()]), n
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback

        q.put((rank, repr(e) + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_real_plan_bucket_marks_two_ranks_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_real_plan, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_bucket_layout_follows_the_process_group_and_rejects_bad_values(monkeypatch):
    from gdrnet_amd import GDRN as G
    from gdrnet_amd.cfg import lm13_cfg
    from gdrnet_amd.engine import Engine

    model, _ = G.build_model_optimizer(lm13_cfg(device="cpu"))
    eng = Engine(dict(model.named_parameters()), dict(model.named_buffers()), dtype="bf16", dry=True)
    assert len(eng.bucket_bounds) == 4 and not eng.buckets_from_env
    eng.plan(2, True, True)
    assert eng.set_bucket_layout(5) and len(eng.bucket_bounds) == 5 and not eng.plans   # plans are rebuilt for the new grouping
    assert not eng.set_bucket_layout(5)
    monkeypatch.setenv("GDRN_BUCKETS", "3")
    with pytest.raises(ValueError, match="GDRN_BUCKETS"):
        Engine(dict(model.named_parameters()), dict(model.named_buffers()), dtype="bf16", dry=True)


# ---------------------------------------------------------------------------------------------- bench.py starts its own ranks
def _bench(*argv):
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    pr = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    lines = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (pr.stdout[-2000:], pr.stderr[-2000:])   # ONE JSON line, whatever happened
    return pr.returncode, json.loads(lines[0])


def test_bench_gpus2_launches_its_own_ranks_dry_gloo():
    """`python bench.py --gpus 2` (no torch.distributed.run in front, as the driver's single-GPU command looks) starts two ranks itself; --dry
    --backend gloo runs the launcher, the process group and the bucket protocol of the real bs=64 backward launch list without kernels."""
    rc, j = _bench("--gpus", "2", "--backend", "gloo", "--dry", "--steps", "2", "--warmup", "1")
    assert rc == 0, j
    assert j["n_gpus"] == 2 and j["process_group"]["world_size"] == 2 and j["dry"] is True and j["protocol_ok"] is True
    assert j["buckets"] == 5 and abs(sum(j["bucket_mb"]) - 140.2) < 0.5          # the data-parallel layout, 35.05 M fp32 gradients
    assert j["config"]["global_batch"] == 128 and j["value"] > 0 and j["steps"] == 2 and j["warmup"] == 1


@pytest.mark.skipif(torch.cuda.is_available() and torch.cuda.device_count() >= 2, reason="box has two GPUs: the launch would really run")
def test_bench_gpus2_without_two_gpus_is_one_json_error_line():
    rc, j = _bench("--gpus", "2", "--steps", "2", "--warmup", "1")
    assert rc != 0 and j["value"] is None and j["n_gpus"] == 2 and "needs 2 visible GPUs" in j["error"]


# ---------------------------------------------------------------------------------------------- divergence safety net of the data-parallel step
def _worker_divergence(rank, world, port, q):
    """GDRN._dp_divergence_check (the per-bucket optimizer behind the all-reduce has only run with one real rank): ranks holding
    bit-identical parameters pass, a rank whose parameters differ in ONE element makes every rank raise; after GDRN_DP_CHECK_STEPS checks
    the function is a no-op (no host synchronisation in steady state)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from types import SimpleNamespace as NS

    from gdrnet_amd import GDRN as G
    from gdrnet_amd.cabi import GdrnHipError

    try:
        torch.manual_seed(0)
        P = {"a.weight": torch.nn.Parameter(torch.randn(6, 5)), "b.bias": torch.nn.Parameter(torch.randn(7))}
        eng = NS(P=P, param_names=list(P))
        m = NS(_reducer=NS(world=world, group=None, active=True))
        m.__dict__["_dp_checked"] = 0
        G.GDRN._dp_divergence_check(m, eng)          # identical replicas: passes, one check consumed
        assert m.__dict__["_dp_checked"] == 1
        with torch.no_grad():
            if rank == 1:
                P["a.weight"][2, 3] += 1e-6           # one element on one rank
        try:
            G.GDRN._dp_divergence_check(m, eng)
            raised = False
        except GdrnHipError as ex:
            raised = "diverged" in str(ex)
        assert raised                                  # ... on EVERY rank (all of them see the gathered checksums)
        assert m.__dict__["_dp_checked"] == 2
        G.GDRN._dp_divergence_check(m, eng)          # the budget of checks (2) is used up: no-op even though the replicas still differ
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback

        q.put((rank, repr(e) + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_data_parallel_divergence_check_two_ranks_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_divergence, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
