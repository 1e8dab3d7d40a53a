"""Checkpoint helpers with the reference's names and file format (uformerWM/utils/model_utils.py:10-56):
``{"epoch", "state_dict", "optimizer", ...}`` written with ``torch.save``.  A checkpoint trained by the reference loads
into the MI355X model and its fused optimizer, and the other way round:

* parameters: the ``state_dict`` keys and shapes are the reference's own (``module.`` prefixes of DataParallel runs are
  stripped as model_utils.py:27-36 does);
* optimizer: ``torch.optim.AdamW`` keeps ``exp_avg / exp_avg_sq / step`` per parameter index, the fused optimizer keeps
  two flat arenas.  Indices follow the order of the checkpoint's own ``state_dict`` (parameters only), which is the
  registration order of the model that wrote it.

Files are read with ``torch.load(..., weights_only=True)``: nothing in a checkpoint is executed.
"""
import os
from collections import OrderedDict

import torch


def freeze(model):
    for p in model.parameters():
        p.requires_grad = False


def unfreeze(model):
    for p in model.parameters():
        p.requires_grad = True


def is_frozen(model):
    return not all(p.requires_grad for p in model.parameters())


def save_checkpoint(model_dir, state, session):
    """model_utils.py:22-25."""
    path = os.path.join(model_dir, "model_epoch_{}_{}.pth".format(state["epoch"], session))
    torch.save(state, path)
    return path


def _read(weights):
    return torch.load(weights, map_location="cpu", weights_only=True)


def _strip(state_dict):
    return OrderedDict((k[7:] if "module." in k else k, v) for k, v in state_dict.items())


def load_checkpoint(model, weights):
    """model_utils.py:27-36."""
    sd = _read(weights)["state_dict"]
    try:
        model.load_state_dict(sd)
    except RuntimeError:
        model.load_state_dict(_strip(sd))


def load_checkpoint_multigpu(model, weights):
    """model_utils.py:39-46: every key carries the 7-character ``module.`` prefix."""
    sd = _read(weights)["state_dict"]
    model.load_state_dict(OrderedDict((k[7:], v) for k, v in sd.items()))


def load_start_epoch(weights):
    return _read(weights)["epoch"]


def _param_order(ckpt, net):
    names = [k for k in _strip(ckpt["state_dict"]) if k in net._offsets]
    if len(names) != len(net._offsets):
        raise ValueError("the checkpoint's state_dict does not list the model's parameters")
    return names


def load_optim(optimizer, weights):
    """model_utils.py:52-56: restore the optimizer, return its learning rate.  ``optimizer`` is a ``FusedAdamW`` (the
    reference's per-parameter AdamW state or the fused optimizer's own flat state are both accepted) or any
    ``torch.optim`` optimizer."""
    ckpt = _read(weights)
    state = ckpt["optimizer"]
    if not hasattr(optimizer, "exp_avg"):
        optimizer.load_state_dict(state)
        return optimizer.param_groups[-1]["lr"]
    if "exp_avg" in state:                                       # written from FusedAdamW.state_dict()
        optimizer.load_state_dict(state)
        return optimizer.param_groups[0]["lr"]
    net = optimizer.net
    order = _param_order(ckpt, net)
    n = net.optimized_numel
    optimizer.exp_avg.zero_()
    optimizer.exp_avg_sq.zero_()
    step = 0
    for idx, st in state["state"].items():
        name = order[int(idx)]
        off, cnt = net._offsets[name], st["exp_avg"].numel()
        if off + cnt > n:
            continue                                             # a parameter the train step never touches
        optimizer.exp_avg[off:off + cnt].copy_(st["exp_avg"].reshape(-1))
        optimizer.exp_avg_sq[off:off + cnt].copy_(st["exp_avg_sq"].reshape(-1))
        step = max(step, int(st["step"]))
    optimizer.step_t.fill_(step)
    g = state["param_groups"][0]
    pg = optimizer.param_groups[0]
    pg.update(lr=g["lr"], betas=tuple(g["betas"]), eps=g["eps"], weight_decay=g["weight_decay"])
    optimizer.sync_lr()
    return g["lr"]


def reference_optimizer_state(optimizer, order=None):
    """The fused optimizer's state in ``torch.optim.AdamW.state_dict()`` form, so a reference run can resume from a
    checkpoint written here.  ``order``: parameter names in the reference model's registration order (the keys of its
    ``state_dict``); defaults to this model's own ``state_dict`` order."""
    net = optimizer.net
    names = [k for k in (order if order is not None else net.state_dict().keys()) if k in net._offsets]
    n, step = net.optimized_numel, float(optimizer.step_t.item())
    state = {}
    for i, name in enumerate(names):
        off, shape = net._offsets[name], net._views[name].shape
        cnt = net._views[name].numel()
        if off + cnt > n or step == 0:
            continue
        state[i] = {"step": torch.tensor(step), "exp_avg": optimizer.exp_avg[off:off + cnt].view(shape).clone(),
                    "exp_avg_sq": optimizer.exp_avg_sq[off:off + cnt].view(shape).clone()}
    g = optimizer.param_groups[0]
    group = {"lr": g["lr"], "betas": tuple(g["betas"]), "eps": g["eps"], "weight_decay": g["weight_decay"],
             "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
             "fused": None, "decoupled_weight_decay": True, "params": list(range(len(names)))}
    return {"state": state, "param_groups": [group]}
