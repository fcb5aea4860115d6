"""Checkpoint import / export: the reference's file format and MyCheckpointer protocol
(core/utils/my_checkpoint.py:9-54, engine.py:190-210,328-333) on the drop-in model -- host logic, no GPU needed."""
import os

import numpy as np
import pytest
import torch

from gdrnet_amd import GDRN, synth
from gdrnet_amd.cfg import lm13_cfg
from gdrnet_amd.checkpoint import MyCheckpointer, PeriodicCheckpointer


@pytest.fixture(scope="module")
def model_opt():
    model, opt = GDRN.build_model_optimizer(lm13_cfg(device="cpu"))
    model.load_state_dict(synth.make_state_dict(0))
    return model, opt


def _perturb(model):
    with torch.no_grad():
        for p in model.parameters():
            p.add_(1.0)


def test_save_load_round_trip_in_the_reference_format(tmp_path, model_opt):
    model, opt = model_opt
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda it: 1.0)
    ck = MyCheckpointer(model, str(tmp_path), save_to_disk=True, optimizer=opt, scheduler=sched)
    assert not ck.has_checkpoint() and ck.resume_or_load("", resume=True) == {}
    want = {k: v.clone() for k, v in model.state_dict().items()}
    f = ck.save("model_0000009", iteration=9)
    data = torch.load(f, map_location="cpu", weights_only=False)
    assert sorted(data) == ["iteration", "model", "optimizer", "scheduler"]
    assert list(data["model"]) == list(synth.param_schema())  # the reference's state_dict keys, in its order
    assert open(os.path.join(tmp_path, "last_checkpoint")).read() == "model_0000009.pth"
    _perturb(model)
    rest = MyCheckpointer(model, str(tmp_path), optimizer=opt, scheduler=sched).resume_or_load("ignored.pth", resume=True)
    assert rest["iteration"] == 9 and rest["__incompatible__"].missing_keys == []
    for k, v in model.state_dict().items():
        assert torch.equal(v, want[k]), k


def test_released_weight_layouts_load(tmp_path, model_opt):
    model, _ = model_opt
    sd = synth.make_state_dict(3)
    # (a) bare state dict, (b) DDP "module." prefix inside {"model": ...}, (c) numpy arrays
    torch.save(sd, tmp_path / "bare.pth")
    torch.save({"model": {"module." + k: v for k, v in sd.items()}, "iteration": 5}, tmp_path / "ddp.pth")
    torch.save({"model": {k: v.numpy() for k, v in sd.items()}}, tmp_path / "np.pth")
    for name in ("bare.pth", "ddp.pth", "np.pth"):
        _perturb(model)
        rest = MyCheckpointer(model).resume_or_load(str(tmp_path / name), resume=False)
        inc = rest["__incompatible__"]
        assert inc.missing_keys == [] and inc.unexpected_keys == [] and inc.incorrect_shapes == []
        for k, v in model.state_dict().items():
            assert torch.equal(v, sd[k].reshape(v.shape)), (name, k)
    assert MyCheckpointer(model).resume_or_load(str(tmp_path / "ddp.pth"), resume=False)["iteration"] == 5


def test_shape_mismatch_is_skipped_and_reported(tmp_path, model_opt):
    model, _ = model_opt
    sd = synth.make_state_dict(0)
    sd["pnp_net.fc_r.weight"] = torch.zeros(4, 256)  # another rotation parametrisation
    sd["extra.weight"] = torch.zeros(1)
    sd.pop("backbone.conv1.weight")
    torch.save({"model": sd}, tmp_path / "other.pth")
    before = model.state_dict()["pnp_net.fc_r.weight"].clone()
    inc = MyCheckpointer(model).load(str(tmp_path / "other.pth"))["__incompatible__"]
    assert inc.incorrect_shapes == [("pnp_net.fc_r.weight", (4, 256), (6, 256))]
    assert inc.missing_keys == ["backbone.conv1.weight"] and inc.unexpected_keys == ["extra.weight"]
    assert torch.equal(model.state_dict()["pnp_net.fc_r.weight"], before)
    with pytest.raises(FileNotFoundError):
        MyCheckpointer(model).load(str(tmp_path / "nope.pth"))
    with pytest.raises(NotImplementedError):
        MyCheckpointer(model).load("torchvision://resnet34")


def test_periodic_checkpointer_and_rank_gate(tmp_path, model_opt):
    model, opt = model_opt
    ck = MyCheckpointer(model, str(tmp_path / "run"), save_to_disk=True, optimizer=opt)
    pc = PeriodicCheckpointer(ck, period=2, max_iter=7, max_to_keep=2)
    for it in range(7):
        pc.step(it, epoch=0)
    files = sorted(os.path.basename(f) for f in ck.get_all_checkpoint_files())
    assert files == ["model_0000003.pth", "model_0000005.pth", "model_final.pth"]
    assert os.path.basename(ck.get_checkpoint_file()) == "model_final.pth"
    assert ck.resume_or_load("", resume=True)["iteration"] == 6
    silent = MyCheckpointer(model, str(tmp_path / "rank1"), save_to_disk=False)
    assert silent.save("model_x") is None and not os.path.exists(tmp_path / "rank1")


def test_optimizer_state_round_trip(tmp_path):
    from gdrnet_amd.ranger import Ranger

    p = [torch.nn.Parameter(torch.from_numpy(np.arange(6, dtype=np.float32).reshape(2, 3)))]
    opt = Ranger(p, lr=1e-3)
    opt.state[p[0]].update(step=3, exp_avg=torch.ones(2, 3), exp_avg_sq=torch.full((2, 3), 2.0), slow_buffer=torch.zeros(2, 3))
    m = torch.nn.Linear(1, 1)
    MyCheckpointer(m, str(tmp_path), save_to_disk=True, optimizer=opt).save("m")
    opt2 = Ranger([torch.nn.Parameter(torch.zeros(2, 3))], lr=5e-2)
    MyCheckpointer(m, str(tmp_path), optimizer=opt2).resume_or_load("", resume=True)
    st = next(iter(opt2.state.values()))
    assert st["step"] == 3 and torch.equal(st["exp_avg_sq"], torch.full((2, 3), 2.0)) and opt2.param_groups[0]["lr"] == 1e-3
