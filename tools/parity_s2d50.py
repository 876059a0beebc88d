"""Configuration 3 at its full size against the oracle: one distill.S2DTrainer step (static + dynamic memories through the
hallucinator, DM loss, backward to the dynamic memories and the hallucinator) over C = 50 classes x (64 real + 1 composed) clips
112x112x16 in the shipped precision mode, against the same step on the CPU oracle in fp64 (class by class).
   python tools/parity_s2d50.py [classes] [seed]      -> gpurun_out/r06_parity_s2d50.json + a summary on stdout"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import ref_cpu as R
from video_distillation_amd import distill, plan

C = int(sys.argv[1]) if len(sys.argv) > 1 else 50
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 33
T, S, B, NP, vpc, spc, dpc = 16, 112, 64, 66, 1, 2, 2


def rel(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / b.norm())


t0 = time.time()
dev = torch.device("cuda:0")
geo = plan.NetGeometry(T, S, S)
g = torch.Generator(device=dev).manual_seed(seed)
base = torch.randn(C, 1, T, 3, S, S, device=dev, generator=g)
clips = (base + 0.1 * torch.randn(C, NP, T, 3, S, S, device=dev, generator=g)).reshape(C * NP, T, 3, S, S)
del base
pool = distill.RealPool(clips, [NP] * C, [c * NP for c in range(C)])
static = torch.randn(C * spc, 3, S, S, device=dev, generator=g)
dynamic = torch.randn(C, dpc, T, 1, S, S, device=dev, generator=g)
hal_w = (torch.rand(3, 4, 3, 3, 3, device=dev, generator=g) * 2 - 1) * 0.096
hal_b = (torch.rand(3, device=dev, generator=g) * 2 - 1) * 0.096
be = distill.HipBackend(geo, dev)
tr = distill.S2DTrainer(be, pool, C, vpc, spc, dpc, B, static.clone(), dynamic.clone(), hal_w.clone(), hal_b.clone(), lr_dynamic=0.01, lr_hal=1e-6)
sidx, didx = tr.indices(0)
idx = distill.sample_real_indices(0, pool.counts, pool.offsets, B, list(range(C)))
loss_hip = float(tr.step(0))
if hasattr(tr, "sync"):
    tr.sync()
torch.cuda.synchronize()
g_dyn, g_w, g_b = (t.detach().cpu() for t in tr.last_grads)
t1 = time.time()
# ---- the oracle, fp64 ----
torch.set_num_threads(min(32, os.cpu_count() or 1))
params = [w.cpu().double() for w in be.new_network(seed=0)]
st = static.cpu().double()
dy = dynamic.reshape(C * dpc, T, 1, S, S).cpu().double().requires_grad_(True)
w, b = hal_w.cpu().double().requires_grad_(True), hal_b.cpu().double().requires_grad_(True)
img = R.hallucinator(st[torch.as_tensor(sidx)], dy[torch.as_tensor(didx)], w, b)
loss = torch.zeros((), dtype=torch.float64)
for c in range(C):
    real = clips[torch.as_tensor(idx[c * B:(c + 1) * B], device=dev)].cpu().double()
    with torch.no_grad():
        f_real = R.convnet3d_embed(real, params)
    loss = loss + R.dm_class_term(f_real, R.convnet3d_embed(img[c * vpc:(c + 1) * vpc], params))
gd, gw, gb = torch.autograd.grad(loss, [dy, w, b])
t2 = time.time()
g_dyn = g_dyn.reshape(gd.shape)
rows = [rel(g_dyn[r], gd[r]) for r in sorted(set(int(v) for v in didx))]
untouched = [r for r in range(C * dpc) if r not in set(int(v) for v in didx)]
zero_ok = all(float(g_dyn[r].abs().sum()) == 0.0 for r in untouched)
out = {"classes": C, "seed": seed, "loss_hip": loss_hip, "loss_oracle_fp64": float(loss), "loss_rel": abs(loss_hip / float(loss) - 1),
       "g_dynamic_rel_l2": rel(g_dyn, gd), "g_dynamic_rel_l2_per_selected_memory": rows, "g_hal_w_rel_l2": rel(g_w, gw), "g_hal_b_rel_l2": rel(g_b, gb),
       "unselected_memories_exactly_zero": zero_ok, "oracle_seconds": t2 - t1,
       "command": "python tools/parity_s2d50.py %d %d" % (C, seed)}
print("config 3, %d classes x (64 real + 1 composed) clips 112x112x16, shipped mode vs fp64 oracle: loss %.6f vs %.6f (rel %.1e); dynamic-memory gradient "
      "rel-L2 %.2e (per selected memory: median %.2e, max %.2e); hallucinator weight / bias gradient %.2e / %.2e; unselected memories exactly zero: %s   "
      "(HIP side %.0f s, oracle %.0f s)" % (C, loss_hip, float(loss), out["loss_rel"], out["g_dynamic_rel_l2"], float(np.median(rows)), max(rows),
                                            out["g_hal_w_rel_l2"], out["g_hal_b_rel_l2"], zero_ok, t1 - t0, t2 - t1))
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r06_parity_s2d50.json")
os.makedirs(os.path.dirname(path), exist_ok=True)
json.dump(out, open(path, "w"), indent=1)
