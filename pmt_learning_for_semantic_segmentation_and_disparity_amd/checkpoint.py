"""Checkpoint interop with the reference's `.pth.tar` files.

The reference writes `{'epoch', 'state_dict', 'optimizer', 'train_cm', 'test_cm', 'best_metric', 'epoch_history',
'IoU_history_val', 'disp_history_val', 'loss_history_val', 'IoU_history_train', 'disp_history_train',
'loss_history_train'[, 'amp']}` with `torch.save` (torch_implementation.py:915-934) and reads it back in
`load_checkpoint_and_params` (util/utilTorch_loadweight.py:6-115).  `state_dict` is the DistributedDataParallel
wrapper's, so every key carries a `module.` prefix; `optimizer` is `torch.optim.Adam.state_dict()`, whose per-parameter
state is indexed by the position of the parameter in `net.parameters()`.

Here the parameters live in ONE flat f32 buffer with flat Adam moments beside it (train.TrainStep), so this module
maps between the two layouts.  The networks of this package register their parameters in the reference's order
(tests/test_checkpoint.py checks that against a key list captured from the reference), which makes the index mapping
the identity.  Files written here load in the reference and vice versa.
"""
import os
import shutil

import torch

HISTORY_KEYS = ("epoch_history", "IoU_history_val", "disp_history_val", "loss_history_val", "IoU_history_train",
                "disp_history_train", "loss_history_train")
# util/utilTorch_loadweight.py:33-40: one tensor is re-homed and two are skipped when loading "by name"
RENAME_BY_NAME = {"module.Conv2DownUp11.1.ct2d.weight": "module.convOutput.ct2d.weight"}
SKIP_BY_NAME = ("module.segNet.Conv2DownUp2.1.ct2d.weight", "module.Conv2DownUp11.1.ct2d.weight")


def _param_slices(model):
    """[(name, param, offset, numel)] in `model.parameters()` order; offsets as train.flatten_parameters lays them out
    (16-byte aligned slices)."""
    out, off = [], 0
    for name, p in model.named_parameters():
        out.append((name, p, off, p.numel()))
        off += ((p.numel() + 3) // 4) * 4
    return out


def add_prefix(state_dict, prefix="module."):
    return {prefix + k: v for k, v in state_dict.items()}


def strip_prefix(state_dict, prefix="module."):
    return {(k[len(prefix):] if k.startswith(prefix) else k): v for k, v in state_dict.items()}


def _match_prefix(state_dict, own_keys):
    """Bring checkpoint keys to the naming of `own_keys` (with or without the DDP `module.` prefix)."""
    own_pref = all(k.startswith("module.") for k in own_keys) if own_keys else False
    ck_pref = all(k.startswith("module.") for k in state_dict) if state_dict else False
    if ck_pref and not own_pref:
        return strip_prefix(state_dict)
    if own_pref and not ck_pref:
        return add_prefix(state_dict)
    return state_dict


def load_model_state(model, state_dict, by_name=False):
    """`net.load_state_dict(state_dict)` (strict) or the reference's by-name copy (utilTorch_loadweight.py:30-46,83-101):
    tensors whose name is unknown are ignored, `Conv2DownUp11.1.ct2d.weight` is re-homed to `convOutput.ct2d.weight`
    when that exists, and the two skipped names stay untouched.  Parameters are written in place, so they keep
    aliasing the flat buffer of a TrainStep.  Returns the list of names that were copied."""
    own = model.state_dict()
    own_pref = bool(own) and all(k.startswith("module.") for k in own)
    if not by_name:
        model.load_state_dict(_match_prefix(state_dict, list(own.keys())))
        _invalidate()
        return list(own.keys())
    src = add_prefix(strip_prefix(state_dict))                     # canonical "module."-prefixed names, as the reference sees them
    own_c = own if own_pref else add_prefix(own)
    copied = []
    with torch.no_grad():
        for name, param in src.items():
            if name in RENAME_BY_NAME and RENAME_BY_NAME[name] in own_c:
                own_c[RENAME_BY_NAME[name]].copy_(param)
                copied.append(RENAME_BY_NAME[name])
            if name not in own_c or name in SKIP_BY_NAME:
                continue
            own_c[name].copy_(param)
            copied.append(name)
    _invalidate()
    return copied


def _invalidate():
    try:
        from . import ops
    except ImportError:         # CPU-only tooling (no libsdhip.so): nothing is cached
        return
    ops.invalidate_packed_weights()


def optimizer_state_dict(step):
    """`torch.optim.Adam.state_dict()` of a TrainStep-like object (attributes model, exp_avg, exp_avg_sq, steps_done, lr,
    betas, eps): the flat moments are cut back into per-parameter tensors."""
    state = {}
    slices = _param_slices(step.model)
    t = float(step.steps_done)
    for i, (_, p, off, n) in enumerate(slices):
        if t == 0:
            continue            # torch creates the per-parameter state lazily at the first step
        state[i] = {"step": torch.tensor(t), "exp_avg": step.exp_avg[off:off + n].view(p.shape).clone(),
                    "exp_avg_sq": step.exp_avg_sq[off:off + n].view(p.shape).clone()}
    group = {"lr": step.lr, "betas": tuple(step.betas), "eps": step.eps, "weight_decay": 0, "amsgrad": False,
             "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
             "params": list(range(len(slices)))}
    return {"state": state, "param_groups": [group]}


def load_optimizer_state(step, opt_state):
    """Inverse of optimizer_state_dict: fills the flat moments, the step count and beta1^t / beta2^t."""
    slices = _param_slices(step.model)
    groups = opt_state["param_groups"]
    order = [i for g in groups for i in g["params"]]
    if len(order) != len(slices):
        raise ValueError("optimizer state holds %d parameters, the model has %d" % (len(order), len(slices)))
    g0 = groups[0]
    step.lr, step.betas, step.eps = float(g0["lr"]), tuple(g0["betas"]), float(g0["eps"])
    t = 0.0
    with torch.no_grad():
        step.exp_avg.zero_()
        step.exp_avg_sq.zero_()
        for pos, idx in enumerate(order):
            st = opt_state["state"].get(idx)
            if st is None:
                continue
            _, p, off, n = slices[pos]
            if tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise ValueError("optimizer state %d has shape %s, parameter %s has %s" %
                                 (idx, tuple(st["exp_avg"].shape), slices[pos][0], tuple(p.shape)))
            step.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
            step.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
            t = max(t, float(st["step"]))
        step.steps_done = int(t)
        step.beta_pow.copy_(torch.tensor([step.betas[0] ** t, step.betas[1] ** t], dtype=torch.float64).to(step.beta_pow.dtype))


def make_state(step, epoch, histories=None, best_metric=(1, 0), train_cm=None, test_cm=None, ddp_prefix=True):
    """The dict the reference saves (torch_implementation.py:915-928).  `epoch` is the number of finished epochs."""
    sd = {k: v.detach().clone() for k, v in step.model.state_dict().items()}
    state = {"epoch": epoch, "state_dict": add_prefix(sd) if ddp_prefix else sd, "optimizer": optimizer_state_dict(step),
             "train_cm": train_cm, "test_cm": test_cm, "best_metric": list(best_metric)}
    histories = histories or {}
    for k in HISTORY_KEYS:
        state[k] = list(histories.get(k, []))
    return state


def save_checkpoint(state, old_loss, new_loss, old_D_error, new_D_error, filename="checkpoint"):
    """util/utilTorch_loadweight.py:117-132, same file names: `<filename>.pth.tar` always, plus a
    `<filename>_model_best_IOU{acc}_Derr{err}.pth.tar` copy (replacing the previous best) when the score improved."""
    path = filename + ".pth.tar"
    if new_loss > old_loss:
        new_loss, old_loss = round(new_loss, 4), round(old_loss, 4)
        old_D_error, new_D_error = round(old_D_error, 4), round(new_D_error, 4)
        state["best_metric"] = [new_D_error, new_loss]
        torch.save(state, path)
        prev = filename + "_model_best_IOU{}_Derr{}.pth.tar".format(old_loss, old_D_error)
        if os.path.exists(prev):
            os.remove(prev)
        shutil.copyfile(path, filename + "_model_best_IOU{}_Derr{}.pth.tar".format(new_loss, new_D_error))
    else:
        torch.save(state, path)
    return path


def load_checkpoint_and_params(path, step, load_weights_by_name=False, map_location="cpu"):
    """Counterpart of util/utilTorch_loadweight.py:6-115 for a TrainStep: returns the same 9-tuple
    (start_e, best_metric, epoch_history, IoU_history_val, disp_history_val, loss_history_val, IoU_history_train,
    disp_history_train, loss_history_train).  A `.tar` file is the full dict; anything else is a bare state_dict
    (start epoch 0, optimizer untouched).  The optimizer state is restored unless loading by name (:66-68)."""
    start_e, best_metric = 0, [1, 0]
    hist = {k: [] for k in HISTORY_KEYS}
    if path:
        ck = torch.load(path, map_location=map_location, weights_only=False)
        if path.rsplit(".", 1)[-1] == "tar":
            load_model_state(step.model, ck["state_dict"], by_name=load_weights_by_name)
            start_e = ck["epoch"]
            if not load_weights_by_name:
                load_optimizer_state(step, ck["optimizer"])
            if "IoU_history_val" in ck:
                best_metric = ck["best_metric"]
                for k in HISTORY_KEYS:
                    hist[k] = ck[k]
        else:
            load_model_state(step.model, ck, by_name=load_weights_by_name)
    return (start_e, best_metric) + tuple(hist[k] for k in HISTORY_KEYS)
