"""Checkpoint save / load with the reference's call surface.

``load_checkpoint(model, optimizer, path) -> None`` mutates in place (reference train/model_loader.py:35-42);
``save_checkpoint(model, optimizer, path) -> None`` (reference train/rl_nonadversarial.py:62-67).
The reference stores ``{"model": nnx.state(model), "optimizer": nnx.state(optimizer)}`` with orbax in a directory
``path``; here the same two-entry tree (parameter names = Flax attribute paths) is one ``checkpoint.pt`` inside it.
"""
import os

import torch


def save_checkpoint(model, optimizer, path):
    os.makedirs(path, exist_ok=True)
    state = {
        "model": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()},
        "optimizer": optimizer.state_dict(),
    }
    tmp = os.path.join(path, "checkpoint.pt.tmp")
    torch.save(state, tmp)
    os.replace(tmp, os.path.join(path, "checkpoint.pt"))


def load_checkpoint(model, optimizer, path):
    state = torch.load(os.path.join(path, "checkpoint.pt"), map_location="cpu", weights_only=True)
    own = model.state_dict()
    missing = set(own) - set(state["model"])
    extra = set(state["model"]) - set(own)
    if missing or extra:
        raise KeyError(f"checkpoint/model mismatch: missing {sorted(missing)[:5]}, unexpected {sorted(extra)[:5]}")
    with torch.no_grad():
        for k, v in state["model"].items():
            own[k].copy_(v)          # in place: keeps parameters aliased to the optimizer's flat buffer
    optimizer.load_state_dict(state["optimizer"])
