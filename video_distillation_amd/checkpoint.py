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


# ---- expert trajectories for MTT (buffer.py:75-104; distill_baseline.py:113-135, 203-216) -------
def save_expert_buffer(save_dir: str, trajectories) -> str:
    """``replay_buffer_{n}.pt``: list over experts of list over epochs of the 8 parameter tensors
    (CPU, parameters() order); n = first free index, as buffer.py:97-101."""
    os.makedirs(save_dir, exist_ok=True)
    n = 0
    while os.path.exists(os.path.join(save_dir, "replay_buffer_{}.pt".format(n))):
        n += 1
    path = os.path.join(save_dir, "replay_buffer_{}.pt".format(n))
    torch.save([[[p.detach().cpu() for p in ep] for ep in traj] for traj in trajectories], path)
    return path


def load_expert_buffers(buffer_dir: str, max_files: int = None):
    """All ``replay_buffer_*.pt`` of a directory, concatenated (distill_baseline.py:116-133);
    raises AssertionError like the reference when none is found."""
    files, n = [], 0
    while os.path.exists(os.path.join(buffer_dir, "replay_buffer_{}.pt".format(n))):
        files.append(os.path.join(buffer_dir, "replay_buffer_{}.pt".format(n)))
        n += 1
    if n == 0:
        raise AssertionError("No buffers detected at {}".format(buffer_dir))
    if max_files is not None:
        files = files[:max_files]
    out = []
    for f in files:
        out += torch.load(f, map_location="cpu")
    return out


def train_expert_trajectories(net_factory, trainloader, args, num_experts: int, train_epochs: int, lr_teacher: float = 0.01,
                              mom: float = 0.0, l2: float = 0.0, decay: bool = False):
    """The producer loop of buffer.py:64-95 (with ``args.eval_mode`` defined): every
    expert is a fresh network trained with SGD through ``utils.epoch('train')`` -- i.e. the HIP
    train step -- recording the parameters before training and after every epoch."""
    from . import utils
    criterion = torch.nn.CrossEntropyLoss().to(args.device)
    if not hasattr(args, "eval_mode"):
        args.eval_mode = "SS"
    trajectories = []
    for _ in range(num_experts):
        net = net_factory().to(args.device)
        net.train()
        lr = lr_teacher
        opt = torch.optim.SGD(net.parameters(), lr=lr, momentum=mom, weight_decay=l2)
        stamps = [[p.detach().cpu() for p in net.parameters()]]
        for e in range(train_epochs):
            utils.epoch("train", trainloader, net, opt, criterion, args)
            stamps.append([p.detach().cpu() for p in net.parameters()])
            if decay and e == train_epochs // 2 + 1:
                lr *= 0.1
                opt = torch.optim.SGD(net.parameters(), lr=lr, momentum=mom, weight_decay=l2)
        trajectories.append(stamps)
    return trajectories
