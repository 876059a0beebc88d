"""On-disk artefacts of the reference's DM drivers (SURVEY 5 "Checkpoint / resume", 8(f)-4).

The reference is save-only; file names and tensor layouts are the contract:
  * ``images_{it}.pt`` / ``images_best.pt``  -- ``torch.save(image_syn.cpu())``, (C*ipc,T,3,H,W)
    (distill_baseline.py:324-332);
  * ``dynamic_{it}.pt`` / ``dynamic_best.pt`` -- dynamic memory flattened to (C*dpc,T,1,H,W),
    ``hal_{it}.pt`` / ``weights_best.pt`` -- ``ModuleList[Conv3DNet].state_dict()`` i.e. keys
    ``{i}.encoder.weight`` / ``{i}.encoder.bias`` (distill_s2d_ms.py:362-392);
  * static memory is LOADED from ``torch.load(path)["image"]`` (distill_s2d_ms.py:96-99).
"""
from __future__ import annotations

import os
from typing import Dict, Sequence

import torch


def save_images(save_dir: str, it: int, image_syn: torch.Tensor, best: bool = False) -> str:
    os.makedirs(save_dir, exist_ok=True)
    path = os.path.join(save_dir, "images_%d.pt" % it)
    torch.save(image_syn.detach().cpu(), path)
    if best:
        torch.save(image_syn.detach().cpu(), os.path.join(save_dir, "images_best.pt"))
    return path


def save_s2d(save_dir: str, it: int, dynamic_syn: torch.Tensor, hal_weights: Sequence[torch.Tensor],
             hal_biases: Sequence[torch.Tensor], best: bool = False) -> None:
    """dynamic_syn (C,dpc,T,1,H,W); hallucinator i = (hal_weights[i], hal_biases[i])."""
    os.makedirs(save_dir, exist_ok=True)
    flat = dynamic_syn.detach().cpu().reshape((-1,) + tuple(dynamic_syn.shape[2:]))
    state: Dict[str, torch.Tensor] = {}
    for i, (w, b) in enumerate(zip(hal_weights, hal_biases)):
        state["%d.encoder.weight" % i] = w.detach().cpu()
        state["%d.encoder.bias" % i] = b.detach().cpu()
    torch.save(flat, os.path.join(save_dir, "dynamic_%d.pt" % it))
    torch.save(state, os.path.join(save_dir, "hal_%d.pt" % it))
    if best:
        torch.save(flat, os.path.join(save_dir, "dynamic_best.pt"))
        torch.save(state, os.path.join(save_dir, "weights_best.pt"))


def load_static(path: str) -> torch.Tensor:
    """Static memory as the reference loads it: a dict with key "image" -> (C*spc,3,H,W)."""
    obj = torch.load(path, map_location="cpu")
    if not isinstance(obj, dict) or "image" not in obj:
        raise KeyError('static memory file must be a dict with key "image" (distill_s2d_ms.py:96-99)')
    return obj["image"].float()


def load_hallucinators(path: str):
    """-> list of (weight, bias) from a ``hal_{it}.pt`` / ``weights_best.pt`` state dict."""
    state = torch.load(path, map_location="cpu")
    out, i = [], 0
    while "%d.encoder.weight" % i in state:
        out.append((state["%d.encoder.weight" % i], state["%d.encoder.bias" % i]))
        i += 1
    if not out:
        raise KeyError("no '{i}.encoder.weight' keys in %s" % path)
    return out
