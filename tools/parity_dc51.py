"""Configuration 4 at its FULL class count: one gradient-matching step of distill.GMTrainer over all 51 classes x (64 real + 5 synthetic)
clips 112x112x16 ('ours' metric, eight class lanes, shipped precisions) against the reference-shaped double backward on the oracle,
class by class (tests/test_gpu_config_geometry.py runs two of the 51 classes; the oracle takes a few seconds per class term).
   python tools/parity_dc51.py [classes]      -> gpurun_out/r06_parity_dc51.json + a summary on stdout"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("VD_GM_LANES", "8")
import numpy as np
import torch
import torch.nn.functional as F

from oracle import ref_cpu as R
from video_distillation_amd import distill, plan

ncls = int(sys.argv[1]) if len(sys.argv) > 1 else 51
K, ipc, B, T, S = 51, 5, 64, 16, 112


def rel(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / b.norm())


t0 = time.time()
geo = plan.NetGeometry(T, S, S)
g = torch.Generator().manual_seed(404)
base = torch.randn(ncls, T, 3, S, S, generator=g)
clips = torch.empty(ncls * (B + ipc), T, 3, S, S)
for c in range(ncls):                                     # class by class: the noise of all classes at once is 34 GB twice
    clips[c * (B + ipc):(c + 1) * (B + ipc)] = base[c][None] + 0.7 * torch.randn(B + ipc, T, 3, S, S, generator=g)
counts = [B + ipc] * ncls + [0] * (K - ncls)
offsets = [c * (B + ipc) for c in range(ncls)] + [0] * (K - ncls)
params = R.init_params(4040, 3, K)
lr_img = 0.1
world = 1 if ncls == K else None
if world is None:                                         # fewer classes: the rank of a world that owns exactly classes [0, ncls)
    world = next(w for w in range(1, K + 1) if distill.class_range(K, 0, w) == (0, ncls))
tr = distill.GMTrainer(distill.HipGMOps("cuda:0", "ours"), distill.RealPool(clips.cuda(), counts, offsets), geo, K, ipc, batch_real=B,
                       lr_img=lr_img, rank=0, world=world, outer_loop=1, inner_loop=1, dropout_p=0.0, net_init=lambda it: params)
syn0 = tr.image_syn.detach().clone()
idx = distill.sample_real_indices(0, counts, offsets, B, list(range(ncls))).reshape(ncls, B)
t1 = time.time()
loss_hip = float(tr.step(0))
torch.cuda.synchronize()
t2 = time.time()
g_hip = ((syn0 - tr.image_syn) / lr_img).cpu()
torch.set_num_threads(min(32, os.cpu_count() or 1))
p = [q.clone().requires_grad_(True) for q in params]
loss_ref, per_clip, per_class_loss = 0.0, [], []
for k in range(ncls):
    xr = clips[torch.as_tensor(idx[k])]
    gw_real = [t.detach() for t in torch.autograd.grad(F.cross_entropy(R.convnet3d_logits(xr, p), torch.full((B,), k)), p)]
    xs = syn0[k * ipc:(k + 1) * ipc].cpu().clone().requires_grad_(True)
    gw_syn = torch.autograd.grad(F.cross_entropy(R.convnet3d_logits(xs, p), torch.full((ipc,), k)), p, create_graph=True)
    loss = R.match_loss(gw_syn, gw_real, "ours")
    (gx,) = torch.autograd.grad(loss, xs)
    loss_ref += float(loss)
    per_class_loss.append(float(loss))
    per_clip += [rel(g_hip[k * ipc + i], gx[i]) for i in range(ipc)]
t3 = time.time()
per = np.asarray(per_clip)
g_ref32 = {}
# the classes holding the worst clips again in fp64: is the outlier the HIP path's or a pooling near-tie that the fp32 oracle resolves its own way too?
worst = []
for k in sorted({int(i) // ipc for i in np.argsort(per)[-4:]}):
    p64 = [q.double().clone().requires_grad_(True) for q in params]
    xr = clips[torch.as_tensor(idx[k])].double()
    gw_real = [t.detach() for t in torch.autograd.grad(F.cross_entropy(R.convnet3d_logits(xr, p64), torch.full((B,), k)), p64)]
    xs = syn0[k * ipc:(k + 1) * ipc].cpu().double().clone().requires_grad_(True)
    gw_syn = torch.autograd.grad(F.cross_entropy(R.convnet3d_logits(xs, p64), torch.full((ipc,), k)), p64, create_graph=True)
    (gx64,) = torch.autograd.grad(R.match_loss(gw_syn, gw_real, "ours"), xs)
    # the fp32 oracle's gradient of this class once more (not kept above)
    p32 = [q.clone().requires_grad_(True) for q in params]
    xr32 = clips[torch.as_tensor(idx[k])]
    gw_real32 = [t.detach() for t in torch.autograd.grad(F.cross_entropy(R.convnet3d_logits(xr32, p32), torch.full((B,), k)), p32)]
    xs32 = syn0[k * ipc:(k + 1) * ipc].cpu().clone().requires_grad_(True)
    gw_syn32 = torch.autograd.grad(F.cross_entropy(R.convnet3d_logits(xs32, p32), torch.full((ipc,), k)), p32, create_graph=True)
    (gx32,) = torch.autograd.grad(R.match_loss(gw_syn32, gw_real32, "ours"), xs32)
    row = {"class": k, "hip_vs_fp32": [rel(g_hip[k * ipc + i], gx32[i]) for i in range(ipc)],
           "hip_vs_fp64": [rel(g_hip[k * ipc + i], gx64[i]) for i in range(ipc)], "fp32_vs_fp64": [rel(gx32[i], gx64[i]) for i in range(ipc)]}
    worst.append(row)
    print("class %d: HIP vs fp32 oracle %s | HIP vs fp64 oracle %s | fp32 vs fp64 oracle %s" % (
        k, ["%.1e" % v for v in row["hip_vs_fp32"]], ["%.1e" % v for v in row["hip_vs_fp64"]], ["%.1e" % v for v in row["fp32_vs_fp64"]]))
# (the test's third bar, every clip within 5e-2, is a flip bound for ITS ten clips; over 255 clips the tail is reported, not asserted)
ok = abs(loss_hip / loss_ref - 1) < 1e-3 and float(np.median(per)) < 3e-3
print("config 4, %d classes x (64 + 5) clips 112x112x16: matching loss HIP %.4f oracle (fp32) %.4f (rel %.1e); pixel gradient rel-L2 per synthetic "
      "clip: median %.2e, p90 %.2e, max %.2e over %d clips; %s   (setup %.0f s, HIP step %.2f s, oracle %.0f s)" % (
          ncls, loss_hip, loss_ref, abs(loss_hip / loss_ref - 1), np.median(per), np.quantile(per, 0.9), per.max(), per.size,
          ("within the test's bars on loss (1e-3) and median (3e-3); clips above 1e-2: %d, above 5e-2: %d" % (int((per > 1e-2).sum()), int((per > 5e-2).sum()))) if ok else "OUT OF TOLERANCE", t1 - t0, t2 - t1, t3 - t2))
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r06_parity_dc51.json")
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump({"classes": ncls, "loss_hip": loss_hip, "loss_oracle_fp32": loss_ref, "loss_rel": abs(loss_hip / loss_ref - 1),
           "grad_rel_l2_per_clip": per.tolist(), "median": float(np.median(per)), "p90": float(np.quantile(per, 0.9)), "max": float(per.max()),
           "oracle_seconds": t3 - t2, "worst_classes_in_fp64": worst, "command": "python tools/parity_dc51.py %d" % ncls}, open(out, "w"), indent=1)
sys.exit(0 if ok else 1)
