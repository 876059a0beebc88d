"""The synthetic-clip side of a DM step by itself (50 clips 112x112x16: hi+lo forward with kept arg-max, DM loss against fixed real
means, hi+lo backward to the pixels, SGD), 20 times and nothing else: for rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_distillation_amd import distill, plan
from video_distillation_amd.networks import _batch_hint
dev = torch.device("cuda:0")
geo = plan.NetGeometry(16, 112, 112)
C = 50
be = distill.HipBackend(geo, dev, chunk=3200, syn_batch_hint=_batch_hint(C))
syn = torch.randn(C, 16, 3, 112, 112, device=dev)
buf = torch.zeros_like(syn)
f_real = torch.randn(C, geo.num_feat, device=dev)
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    w = be.new_network(seed=it)
    be._dither = 8          # as after set_real_weights(w, 64): the real side is dithered, so the synthetic clips take ONE forward
    f_syn, handle = be.embed_syn(syn, w)
    loss_c, g_syn = be.dm_loss(f_real, f_syn, C)
    grad = be.embed_backward(handle, g_syn)
    be.sgd(syn, buf, grad, 1.0, 0.5, first=(it == 0))
torch.cuda.synchronize()
