"""Checkpoint interop (§8(b) "state_dict keys must be unchanged").
CPU: ordered state_dict keys / shapes / parameter order of every network against tests/golden/keys.json (captured
from the reference's classes by oracle/make_golden.py gen_keys); `.pth.tar` round trips against torch.optim.Adam,
which is what the reference saves and loads (torch_implementation.py:915-934, util/utilTorch_loadweight.py).
GPU: a TrainStep resumed from a file continues exactly like the one that wrote it."""
import copy
import json
import os
import types

import pytest
import torch

from oracle import ref_models as R

KEYS = os.path.join(os.path.dirname(__file__), "golden", "keys.json")


def _nets():
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import nn as N, psmnet as P
    return {"mini_a0": lambda: N.minidsnetExt(R.CFG(aspp=0), labels=2, patch_type='1dcorr'),
            "mini_a1": lambda: N.minidsnetExt(R.CFG(aspp=1), labels=2, patch_type='1dcorr'),
            "mini_a2_hanet": lambda: N.minidsnetExt(R.CFG(aspp=2, hanet=1), labels=19, patch_type='1dcorr'),
            "dsnet": lambda: N.dsnet(R.CFG(), labels=2),
            "psmnet192": lambda: P.PSMNet(192),
            "minidsnet": lambda: N.minidsnet(R.CFG(), labels=2, patch_type='1dcorr'),
            "mini_a0_edges": lambda: N.minidsnetExt(R.CFG(aspp=0), labels=2, patch_type='1dcorr', include_edges=True)}


@pytest.mark.parametrize("name", ["mini_a0", "mini_a1", "mini_a2_hanet", "dsnet", "psmnet192", "minidsnet", "mini_a0_edges"])
def test_state_dict_surface_equals_reference(name):
    gold = json.load(open(KEYS))[name]
    m = _nets()[name]()
    assert [[k, list(v.shape)] for k, v in m.state_dict().items()] == gold["state_dict"]
    assert [k for k, _ in m.named_parameters()] == gold["parameters"]
    for k in ("Conv2DownUp5.c1.0.layers.0.c2d.weight", "resnet_features.resnet_features.denseblock.0.denselayer1.conv2.weight"):
        if name.startswith("mini"):
            assert k in m.state_dict()


class _Tiny(torch.nn.Module):
    """Shapes with numel not divisible by 4 exercise the 16-byte aligned slices of the flat buffer."""

    def __init__(self):
        super().__init__()
        self.a = torch.nn.Conv2d(3, 5, 3)
        self.bn = torch.nn.BatchNorm2d(5)
        self.b = torch.nn.Linear(7, 3)


def _fake_step(model, device="cpu"):
    from pmt_learning_for_semantic_segmentation_and_disparity_amd.train import flatten_parameters
    flat_p, flat_g = flatten_parameters(model)
    return types.SimpleNamespace(model=model, flat_p=flat_p, flat_g=flat_g, exp_avg=torch.zeros_like(flat_p),
                                 exp_avg_sq=torch.zeros_like(flat_p), beta_pow=torch.ones(2), steps_done=0,
                                 lr=0.0015, betas=(0.9, 0.999), eps=1e-7)


def _reference_checkpoint(tmp_path, steps=3):
    """What the reference writes: DDP-prefixed state_dict + torch.optim.Adam state + histories."""
    torch.manual_seed(1)
    net = _Tiny()
    opt = torch.optim.Adam(net.parameters(), lr=0.0015, eps=1e-7)
    for _ in range(steps):
        for p in net.parameters():
            p.grad = torch.randn_like(p)
        opt.step()
    state = {"epoch": 7, "state_dict": {"module." + k: v for k, v in net.state_dict().items()}, "optimizer": opt.state_dict(),
             "train_cm": None, "test_cm": None, "best_metric": [0.5, 0.9], "epoch_history": [1, 2], "IoU_history_val": [[0.1, 0.2]],
             "disp_history_val": [[1.0, 2.0]], "loss_history_val": [[3.0, 1.0, 2.0]], "IoU_history_train": [[0.3, 0.4]],
             "disp_history_train": [[1.5, 2.5]], "loss_history_train": [[4.0, 2.0, 2.0]]}
    path = str(tmp_path / "ref.pth.tar")
    torch.save(state, path)
    return net, opt, path


def test_load_reference_checkpoint_into_flat_buffers(tmp_path):
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import checkpoint as C
    net, opt, path = _reference_checkpoint(tmp_path)
    step = _fake_step(_Tiny())
    out = C.load_checkpoint_and_params(path, step)
    assert out[0] == 7 and out[1] == [0.5, 0.9] and out[2] == [1, 2] and out[8] == [[4.0, 2.0, 2.0]]
    for (k, v), (k2, v2) in zip(net.state_dict().items(), step.model.state_dict().items()):
        assert k == k2 and torch.equal(v, v2)
    off = 0
    for i, p in enumerate(step.model.parameters()):
        assert p.data_ptr() == step.flat_p[off:].data_ptr()         # still aliases the flat buffer
        st = opt.state_dict()["state"][i]
        assert torch.equal(step.exp_avg[off:off + p.numel()].view(p.shape), st["exp_avg"])
        assert torch.equal(step.exp_avg_sq[off:off + p.numel()].view(p.shape), st["exp_avg_sq"])
        off += ((p.numel() + 3) // 4) * 4
    assert step.steps_done == 3
    assert torch.allclose(step.beta_pow, torch.tensor([0.9 ** 3, 0.999 ** 3]))


def test_saved_checkpoint_loads_in_torch_adam(tmp_path):
    """make_state + save_checkpoint produce what `net.load_state_dict` / `optimizer.load_state_dict` of the reference accept."""
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import checkpoint as C
    net, opt, path = _reference_checkpoint(tmp_path)
    step = _fake_step(_Tiny())
    C.load_checkpoint_and_params(path, step)
    state = C.make_state(step, epoch=8, histories={"epoch_history": [1, 2, 3]}, best_metric=(0.5, 0.9))
    base = str(tmp_path / "w")
    C.save_checkpoint(state, 0.9, 0.95, 0.5, 0.4, base)
    assert os.path.exists(base + ".pth.tar") and os.path.exists(base + "_model_best_IOU0.95_Derr0.4.pth.tar")
    C.save_checkpoint(state, 0.95, 0.97, 0.4, 0.3, base)              # a better score replaces the previous best copy
    assert not os.path.exists(base + "_model_best_IOU0.95_Derr0.4.pth.tar") and os.path.exists(base + "_model_best_IOU0.97_Derr0.3.pth.tar")
    C.save_checkpoint(state, 0.97, 0.5, 0.3, 0.9, base)               # a worse one only rewrites the epoch file
    assert os.path.exists(base + "_model_best_IOU0.97_Derr0.3.pth.tar")
    ck = torch.load(base + ".pth.tar", weights_only=False)
    assert ck["epoch"] == 8 and ck["epoch_history"] == [1, 2, 3] and all(k.startswith("module.") for k in ck["state_dict"])
    wrapped = torch.nn.Sequential()
    wrapped.add_module("module", _Tiny())                              # the DDP wrapper's naming
    wrapped.load_state_dict(ck["state_dict"])                          # strict
    opt2 = torch.optim.Adam(wrapped.parameters(), lr=1.0)
    opt2.load_state_dict(ck["optimizer"])
    assert opt2.param_groups[0]["lr"] == 0.0015 and opt2.param_groups[0]["eps"] == 1e-7
    # both optimizers continue identically
    for p, q in zip(net.parameters(), wrapped.parameters()):
        g = torch.randn_like(p)
        p.grad, q.grad = g, g.clone()
    opt.step()
    opt2.step()
    for p, q in zip(net.parameters(), wrapped.parameters()):
        assert torch.equal(p, q)


def test_load_by_name_follows_reference_rules(tmp_path):
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import checkpoint as C

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.convOutput = torch.nn.Module()
            self.convOutput.ct2d = torch.nn.ConvTranspose2d(2, 2, 3)
            self.keep = torch.nn.Linear(2, 2)
    m = Net()
    before = copy.deepcopy(m.state_dict())
    src = {"module.Conv2DownUp11.1.ct2d.weight": torch.full((2, 2, 3, 3), 7.0), "module.keep.weight": torch.full((2, 2), 3.0),
           "module.unknown.weight": torch.zeros(1)}
    copied = C.load_model_state(m, src, by_name=True)
    assert sorted(copied) == ["module.convOutput.ct2d.weight", "module.keep.weight"]
    assert torch.all(m.convOutput.ct2d.weight == 7) and torch.all(m.keep.weight == 3)
    assert torch.equal(m.keep.bias, before["keep.bias"])
    with pytest.raises(RuntimeError):
        C.load_model_state(m, src, by_name=False)                      # strict load refuses the unknown / missing keys
    step = _fake_step(_Tiny())
    out = C.load_checkpoint_and_params("", step)
    assert out[0] == 0 and out[1] == [1, 0] and all(o == [] for o in out[2:])


@pytest.mark.gpu
def test_resumed_train_step_continues_identically(tmp_path):
    from oracle.detweights import fill_state_dict
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import checkpoint as C, nn as N, ops
    from pmt_learning_for_semantic_segmentation_and_disparity_amd.train import TrainStep, synthetic_batch
    batch = synthetic_batch(2, 256, 256)
    mk = lambda seed: fill_state_dict(N.minidsnetExt(R.CFG(), labels=2, patch_type='1dcorr'), seed).cuda().train()
    a = TrainStep(mk(5), dtype=torch.float32, use_graph=False)
    for _ in range(2):
        a(*batch)
    base = str(tmp_path / "ck")
    C.save_checkpoint(C.make_state(a, epoch=1), 0.0, 0.5, 1.0, 0.5, base)
    b = TrainStep(mk(6), dtype=torch.float32, use_graph=False)      # different weights until the file is loaded
    start = C.load_checkpoint_and_params(base + ".pth.tar", b, map_location="cuda:0")[0]
    assert start == 1 and b.steps_done == 2
    assert torch.equal(a.flat_p, b.flat_p) and torch.equal(a.exp_avg, b.exp_avg) and torch.equal(a.exp_avg_sq, b.exp_avg_sq)
    assert torch.allclose(a.beta_pow, b.beta_pow, rtol=1e-6)
    la, lb = a(*batch), b(*batch)
    ops.set_step_context(None)
    assert abs(float(la) - float(lb)) <= 1e-4 * abs(float(la))
    rel = float((a.flat_p - b.flat_p).norm() / a.flat_p.norm())
    assert rel < 1e-5, rel


@pytest.mark.gpu
def test_graph_replays_are_counted_as_optimizer_steps(tmp_path):
    """Adam's `step` written to the checkpoint = warm-up steps + graph replays (the recording pass of the capture runs
    nothing and is not a step); a TrainStep resumed from that file continues like the one that wrote it."""
    from oracle.detweights import fill_state_dict
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import checkpoint as C, nn as N, ops
    from pmt_learning_for_semantic_segmentation_and_disparity_amd.train import TrainStep, synthetic_batch
    batch = synthetic_batch(2, 256, 256)
    mk = lambda seed: fill_state_dict(N.minidsnetExt(R.CFG(), labels=2, patch_type='1dcorr'), seed).cuda().train()
    a = TrainStep(mk(5), dtype=torch.float32, use_graph=True)
    n = 3
    for _ in range(n):
        a(*batch)                                  # first call: 2 eager warm-up steps + capture + 1 replay
    assert a.steps_done == 2 + n
    want_pow = torch.tensor([0.9 ** (2 + n), 0.999 ** (2 + n)])
    assert torch.allclose(a.beta_pow.cpu(), want_pow, rtol=1e-5)
    base = str(tmp_path / "ckg")
    state = C.make_state(a, epoch=1)
    assert all(int(v["step"]) == 2 + n for v in state["optimizer"]["state"].values())
    C.save_checkpoint(state, 0.0, 0.5, 1.0, 0.5, base)
    b = TrainStep(mk(6), dtype=torch.float32, use_graph=False)
    C.load_checkpoint_and_params(base + ".pth.tar", b, map_location="cuda:0")
    assert b.steps_done == 2 + n and torch.allclose(a.beta_pow, b.beta_pow, rtol=1e-6)
    la, lb = a(*batch), b(*batch)
    ops.set_step_context(None)
    assert abs(float(la) - float(lb)) <= 2e-3 * max(1.0, abs(float(la)))
