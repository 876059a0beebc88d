"""Host issue time vs device time of one MTT+Ours iteration (S2DMTTTrainer, config 5 shape)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_distillation_amd import distill, plan

C, T, S = 400, 8, 64
dev = torch.device("cuda:0")
geo = plan.NetGeometry(T, S, S)
gen = torch.Generator(device=dev); gen.manual_seed(99)
traj = [distill.fresh_full_network(5, C, dev)]
for e in range(11):
    traj.append([p + 0.01 * p.abs().mean() * torch.randn(p.shape, device=dev, generator=gen) for p in traj[-1]])
ops = distill.HipMTTOps(geo, C, dev, dropout_p=0.5, batch_hint=256)
static = torch.randn(C * 2, 3, S, S, device=dev, generator=gen)
dynamic = torch.randn(C, 2, T, 1, S, S, device=dev, generator=gen)
hal_w = torch.empty(3, 4, 3, 3, 3, device=dev).uniform_(-0.096, 0.096, generator=gen)
hal_b = torch.empty(3, device=dev).uniform_(-0.096, 0.096, generator=gen)
tr = distill.S2DMTTTrainer(ops, C, 1, 2, 2, static, dynamic, hal_w, hal_b, syn_lr=0.01, lr_dynamic=0.01, lr_hal=0.01, lr_lr=1e-5,
                           syn_steps=10, batch_syn=256, expert_epochs=1, max_start_epoch=10)
for it in range(2):
    tr.step(it, traj)
torch.cuda.synchronize()
for it in range(2, 6):
    t0 = time.perf_counter()
    tr.step(it, traj)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("MTT+Ours iteration: host issue %.1f ms, total %.1f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
