#!/usr/bin/env python3
"""DM distillation driver over the HIP hot path (the ``--method DM`` branch of the reference's
distill_baseline.py:292-361, with its flag names for everything that branch reads).

    python -m video_distillation_amd.run_dm --dataset synthetic --ipc 1 --Iteration 100 --eval_it 50
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m video_distillation_amd.run_dm ...

Data: ``--dataset synthetic`` (randn clips); ``--dataset miniUCF101|UCF101|HMDB51|Kinetics400 --data_path D``
(the reference's frame folders, decoded once and kept in HBM: dataset.py); or ``--data_file f.pt`` holding
{"clips": (N,T,3,H,W), "labels": (N,), "test_clips", "test_labels"}.  Logging: JSON lines with the reference's wandb keys
(``Loss``, ``Accuracy/<model>``, ``Max_Accuracy/<model>``, ``Std/<model>``, ``Max_Std/<model>``).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch


def build_parser():
    p = argparse.ArgumentParser(description="DM distillation on MI355X")
    p.add_argument('--dataset', type=str, default='synthetic')
    p.add_argument('--data_file', type=str, default=None)
    p.add_argument('--data_path', type=str, default='distill_utils/data')
    p.add_argument('--num_workers', type=int, default=8, help='decode threads of the preload')
    p.add_argument('--method', type=str, default='DM', choices=['DM'])
    p.add_argument('--model', type=str, default='ConvNet3D')
    p.add_argument('--ipc', type=int, default=1)
    p.add_argument('--eval_mode', type=str, default='SS')
    p.add_argument('--num_eval', type=int, default=5)
    p.add_argument('--eval_it', type=int, default=500)
    p.add_argument('--epoch_eval_train', type=int, default=500)
    p.add_argument('--Iteration', type=int, default=5000)
    p.add_argument('--lr_net', type=float, default=0.01)
    p.add_argument('--lr_img', type=float, default=1.0)
    p.add_argument('--batch_real', type=int, default=64)
    p.add_argument('--batch_train', type=int, default=256)
    p.add_argument('--init', type=str, default='real', choices=['noise', 'real'])
    p.add_argument('--save_path', type=str, default='./logged_files')
    p.add_argument('--frames', type=int, default=16)
    p.add_argument('--im_size', type=int, default=112)
    p.add_argument('--num_classes', type=int, default=50, help='synthetic data only')
    p.add_argument('--pool_per_class', type=int, default=93, help='synthetic data only')
    p.add_argument('--prec_real', type=str, default='f16')
    p.add_argument('--prec_syn', type=str, default='f16x3')
    p.add_argument('--log_file', type=str, default=None)
    p.add_argument('--no_eval', action='store_true')
    return p


def load_data(args, rank, world, geo, device):
    """-> (RealPool with this rank's classes resident on `device`, num_classes, (c_lo, c_hi), testloader)."""
    from . import distill
    if args.data_file is None and args.dataset != 'synthetic':
        from . import dataset as D
        _, im_size, num_classes, _, _, _, dst_train, _, testloader = D.get_dataset(args.dataset, args.data_path,
                                                                                   img_size=(args.im_size, args.im_size))
        c_lo, c_hi = distill.class_range(num_classes, rank, world)
        pool = distill.RealPool.from_dataset(dst_train, num_classes, list(range(c_lo, c_hi)), device, workers=args.num_workers)
        return pool, num_classes, (c_lo, c_hi), testloader
    if args.data_file is None:
        c_lo, c_hi = distill.class_range(args.num_classes, rank, world)
        pool = distill.RealPool.synthetic(args.num_classes, list(range(c_lo, c_hi)), args.pool_per_class, geo, device)
        return pool, args.num_classes, (c_lo, c_hi), None
    blob = torch.load(args.data_file, map_location="cpu")
    clips, labels = blob["clips"].float(), blob["labels"].long()
    num_classes = int(labels.max()) + 1
    c_lo, c_hi = distill.class_range(num_classes, rank, world)
    order = torch.argsort(labels, stable=True)
    clips, labels = clips[order], labels[order]
    counts = torch.bincount(labels, minlength=num_classes).tolist()
    starts = np.concatenate([[0], np.cumsum(counts)]).astype(int)
    a, b = int(starts[c_lo]), int(starts[c_hi])
    offsets = [int(s) - a for s in starts[:-1]]
    pool = distill.RealPool(clips[a:b].to(device), counts, offsets)
    test = None
    if "test_clips" in blob:
        test = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(blob["test_clips"].float(), blob["test_labels"].long()),
                                           batch_size=64, shuffle=False)
    return pool, num_classes, (c_lo, c_hi), test


def run(args, backend=None, log=None):
    from . import checkpoint, distill, plan, utils
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = backend is None
    device = torch.device("cuda", local_rank) if use_cuda else torch.device("cpu")
    if use_cuda:
        torch.cuda.set_device(device)
    if world > 1:
        import torch.distributed as dist
        if not dist.is_initialized():
            dist.init_process_group(backend="nccl" if use_cuda else "gloo")
    geo = plan.NetGeometry(args.frames, args.im_size, args.im_size)
    pool, num_classes, (c_lo, c_hi), testloader = load_data(args, rank, world, geo, device)
    if backend is None:
        backend = distill.HipBackend(geo, device, prec_real=args.prec_real, prec_syn=args.prec_syn)
    image_syn = None
    if args.init == 'noise':
        gen = torch.Generator(device=device); gen.manual_seed(4321 + rank)
        image_syn = torch.randn((c_hi - c_lo) * args.ipc, args.frames, 3, args.im_size, args.im_size, device=device, generator=gen)
    trainer = distill.DMTrainer(backend, pool, num_classes, args.ipc, args.batch_real, args.lr_img, momentum=0.5,
                                rank=rank, world=world, image_syn=image_syn)
    eval_pool = utils.get_eval_pool(args.eval_mode, args.model, args.model)
    best_acc = {m: 0.0 for m in eval_pool}; best_std = {m: 0.0 for m in eval_pool}
    save_dir = os.path.join(args.save_path, "Baseline_DM", "%s_ipc%d_%s" % (args.dataset, args.ipc, args.lr_img))
    out = open(args.log_file, "a") if (args.log_file and rank == 0) else None

    def emit(rec):
        if rank == 0:
            line = json.dumps(rec)
            (log.append(rec) if log is not None else None)
            print(line, flush=True)
            if out:
                out.write(line + "\n"); out.flush()

    eval_its = set(np.arange(0, args.Iteration + 1, args.eval_it).tolist())
    t0 = time.time()
    for it in range(args.Iteration + 1):
        if it in eval_its and not args.no_eval:
            syn_all = trainer.gather_syn()
            trainer.sync()
            save_best = False
            if rank == 0 and testloader is not None:
                label_syn = torch.arange(num_classes).repeat_interleave(args.ipc)
                for model_eval in eval_pool:
                    accs = []
                    for it_eval in range(args.num_eval):
                        net_eval = utils.get_network(model_eval, 3, num_classes, (args.im_size, args.im_size), frames=args.frames, dist=False).to(device)
                        eargs = argparse.Namespace(device=str(device), lr_net=args.lr_net, epoch_eval_train=args.epoch_eval_train,
                                                   batch_train=args.batch_train, model=args.model, eval_mode=args.eval_mode)
                        _, _, acc_test, _ = utils.evaluate_synset(it_eval, net_eval, syn_all.detach().clone(), label_syn, testloader, eargs, mode='none')
                        accs.append(acc_test)
                    mean, std = float(np.mean(accs)), float(np.std(accs))
                    if mean > best_acc[model_eval]:
                        best_acc[model_eval], best_std[model_eval], save_best = mean, std, True
                    emit({"step": it, "Accuracy/%s" % model_eval: mean, "Max_Accuracy/%s" % model_eval: best_acc[model_eval],
                          "Std/%s" % model_eval: std, "Max_Std/%s" % model_eval: best_std[model_eval]})
            if rank == 0 and (save_best or it % 1000 == 0):
                checkpoint.save_images(save_dir, it, syn_all, best=save_best)
        loss = trainer.global_loss(trainer.step(it, overlap=True))
        if it % 10 == 0 or it == args.Iteration:
            trainer.sync()
            emit({"step": it, "Loss": float(loss) / num_classes, "elapsed_s": round(time.time() - t0, 3)})
    trainer.sync()
    if out:
        out.close()
    return trainer


def main(argv=None):
    args = build_parser().parse_args(argv)
    run(args)


if __name__ == "__main__":
    main()
