import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import ref_cpu as R
from tests.cpu_backend import OracleMTTOps
from video_distillation_amd import distill, plan
z = np.load("tests/golden/g10_mtt_step.npz")
C, n_syn = int(z["C"]), int(z["n_syn"])
start = R.init_params(int(z["net_seed"]), 3, C)
g = torch.Generator().manual_seed(int(z["data_seed"]))
target = [p + 0.02 * p.abs().mean() * torch.randn(p.shape, generator=g) for p in start]
image_syn = torch.randn(n_syn, 8, 3, 64, 64, generator=g)
labels = torch.tensor(z["labels"])
hop = distill.HipMTTOps(plan.NetGeometry(8, 64, 64), C, "cuda:0", dropout_p=0.0)
oop = OracleMTTOps()
rel = lambda a, b: float((a.cpu().double() - b.double()).norm() / b.double().norm())
for idx in [torch.tensor(i) for i in z["indices"]]:
    for sub in (idx, idx[:1], idx[1:]):
        gh, hh = hop.grads([p.cuda() for p in start], image_syn[sub].cuda(), labels[sub].cuda())
        go, ho = oop.grads(start, image_syn[sub], labels[sub])
        p64 = [p.double().requires_grad_(True) for p in start]
        ce = torch.nn.functional.cross_entropy(R.convnet3d_logits(image_syn[sub].double(), p64), labels[sub])
        g64 = torch.autograd.grad(ce, p64)
        print(sub.tolist(), "hip vs fp32 oracle:", ["%.1e" % rel(a, b) for a, b in zip(gh, go)])
        print("        hip vs fp64:", ["%.1e" % rel(a, b) for a, b in zip(gh, g64)], " fp32 vs fp64:", ["%.1e" % rel(a, b) for a, b in zip(go, g64)])
