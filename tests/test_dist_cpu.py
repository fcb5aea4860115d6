"""N > 1 path on CPU: two gloo ranks exercise the bucketed gradient exchange (gdrnet_amd.dist.GradReducer), the
initial parameter broadcast and the loss reduction -- the protocol bench.py / a DDP-free trainer runs over RCCL on
the GPUs (reference: Lightning-Lite DDP wrap, core/gdrn_modeling/main_gdrn.py:134-142; comm.reduce_dict,
core/utils/my_comm.py:8)."""
import os
import socket

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
