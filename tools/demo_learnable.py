#!/usr/bin/env python3
"""End-to-end functional check on a LEARNABLE synthetic dataset (no real dataset ships): class c =
fixed random spatio-temporal template + noise.  Distil with DM on the HIP path, then train fresh
ConvNet3Ds on the synthetic clips with evaluate_synset and test on held-out clips.  Prints accuracy
at iteration 0 (synthetic clips = one noisy real clip per class) and after distillation."""
import argparse, json, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from video_distillation_amd import distill, plan, utils

ap = argparse.ArgumentParser()
ap.add_argument("--classes", type=int, default=10)
ap.add_argument("--iters", type=int, default=300)
ap.add_argument("--noise", type=float, default=2.0)
ap.add_argument("--lr_img", type=float, default=5.0)
ap.add_argument("--epochs", type=int, default=150)
ap.add_argument("--num_eval", type=int, default=2)
args = ap.parse_args()
dev = torch.device("cuda:0")
C, T, S = args.classes, 8, 64
g = torch.Generator().manual_seed(0)
templates = torch.randn(C, T, 3, S, S, generator=g)
templates = torch.nn.functional.avg_pool2d(templates.view(-1, 1, S, S), 9, 1, 4).view(C, T, 3, S, S) * 6  # smooth patterns


def sample(n_per):
    x = templates.repeat_interleave(n_per, 0) + args.noise * torch.randn(C * n_per, T, 3, S, S, generator=g)
    y = torch.arange(C).repeat_interleave(n_per)
    return x, y


train_x, train_y = sample(40)
test_x, test_y = sample(20)
geo = plan.NetGeometry(T, S, S)
be = distill.HipBackend(geo, dev)
pool = distill.RealPool(train_x.to(dev), [40] * C, [40 * c for c in range(C)])
tr = distill.DMTrainer(be, pool, C, 1, 32, lr_img=args.lr_img)
testloader = torch.utils.data.DataLoader(utils.TensorDataset(test_x, test_y), batch_size=64)
eargs = types.SimpleNamespace(device="cuda", lr_net=0.01, epoch_eval_train=args.epochs, batch_train=256, model="ConvNet3D", eval_mode="SS")


def evaluate(tag):
    accs = []
    for k in range(args.num_eval):
        net = utils.get_network("ConvNet3D", 3, C, (S, S), frames=T, dist=False).to(dev)
        _, acc_tr, acc_te, _ = utils.evaluate_synset(k, net, tr.image_syn.detach().clone(), torch.arange(C), testloader, eargs, mode="none")
        accs.append(acc_te)
    print(json.dumps({"stage": tag, "test_acc_mean": float(np.mean(accs)), "test_acc": [float(a) for a in accs]}), flush=True)


evaluate("it 0 (init = one noisy real clip per class)")
losses = []
for it in range(args.iters):
    losses.append(tr.step(it, overlap=True))
tr.sync()
print(json.dumps({"loss_first": float(losses[0]) / C, "loss_last": float(losses[-1]) / C}))
evaluate("after %d DM iterations" % args.iters)
