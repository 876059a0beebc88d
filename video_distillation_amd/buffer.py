#!/usr/bin/env python3
"""Expert trajectories for trajectory matching: the reference's ``buffer.py`` on the HIP train step.

    python -m video_distillation_amd.buffer --dataset miniUCF101 --data_path D --num_experts 10 --train_epochs 50

Same flags and the same ``replay_buffer_{n}.pt`` files as buffer.py:106-128 / :75-104 (list over experts of list over
epochs of the 8 parameter tensors, ``save_interval`` experts per file).  What differs on purpose: the reference leaves
``args.eval_mode`` undefined, so its first ``epoch()`` raises — it is 'SS' here; the real clips are decoded
once and stay in HBM (``dataset.preload``; the reference's ``--preload`` keeps a host copy and its DataLoader re-uploads
every batch), batches are device-side gathers in the DataLoader's shuffle order (``dataset.DeviceBatches``); there is no
wandb.  Training itself is ``utils.epoch('train')`` = ``ConvNet3D.hip_train_step`` per batch
(``checkpoint.train_expert_trajectories``).
"""
from __future__ import annotations

import argparse
import os

import torch


def build_parser():
    p = argparse.ArgumentParser(description='Parameter Processing')
    p.add_argument('--dataset', type=str, default='miniUCF101')
    p.add_argument('--model', type=str, default='ConvNet3D')
    p.add_argument('--num_experts', type=int, default=100)
    p.add_argument('--lr_teacher', type=float, default=0.001)
    p.add_argument('--batch_train', type=int, default=256)
    p.add_argument('--batch_real', type=int, default=256)
    p.add_argument('--num_workers', type=int, default=8)
    p.add_argument('--data_path', type=str, default='distill_utils/data')
    p.add_argument('--buffer_path', type=str, default='./logs/buffers')
    p.add_argument('--train_epochs', type=int, default=50)
    p.add_argument('--decay', action='store_true')
    p.add_argument('--mom', type=float, default=0)
    p.add_argument('--l2', type=float, default=0)
    p.add_argument('--save_interval', type=int, default=10)
    p.add_argument('--preload', action='store_true', help='accepted for compatibility: the clips are always HBM-resident')
    p.add_argument('--im_size', type=int, default=112)
    p.add_argument('--frames', type=int, default=16)
    return p


def run(args, train=None, num_classes=None, log=print):
    """``train`` = (clips (N,T,3,H,W) on the device, labels) skips the dataset read (tests, synthetic data)."""
    from . import checkpoint, dataset as D, utils
    args.device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(args.device)
    args.eval_mode = 'SS'
    if train is None:
        _, im_size, num_classes, _, _, _, dst_train, _, _ = D.get_dataset(args.dataset, args.data_path, img_size=(args.im_size, args.im_size))
        clips, labels = D.preload(dst_train, args.device, workers=args.num_workers)
    else:
        clips, labels = train
        im_size = tuple(clips.shape[-2:])
    frames = int(clips.shape[1])
    loader = D.DeviceBatches(clips, labels, args.batch_train, shuffle=True)
    files, pending = [], []
    for it in range(args.num_experts):
        traj = checkpoint.train_expert_trajectories(
            lambda: utils.get_network(args.model, 3, num_classes, im_size, frames=frames, dist=False), loader, args,
            num_experts=1, train_epochs=args.train_epochs, lr_teacher=args.lr_teacher, mom=args.mom, l2=args.l2, decay=args.decay)
        pending += traj
        log("expert %d: %d timestamps" % (it, len(traj[0])))
        if len(pending) == args.save_interval:
            files.append(checkpoint.save_expert_buffer(args.buffer_path, pending))
            log("Saving {}".format(files[-1]))
            pending = []
    return files


def main(argv=None):
    run(build_parser().parse_args(argv))


if __name__ == "__main__":
    main()
