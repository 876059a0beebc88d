"""Host issue time vs device time of a DM step (DMTrainer, config 2 shape)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_distillation_amd import distill, plan
dev = torch.device("cuda:0")
geo = plan.NetGeometry(16, 112, 112)
pool = distill.RealPool.synthetic(50, list(range(50)), 93, geo, dev)
be = distill.HipBackend(geo, dev, chunk=4096)
tr = distill.DMTrainer(be, pool, 50, 1, 64, lr_img=1.0)
for it in range(3):
    tr.step(it, overlap=True)
tr.sync(); torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for it in range(3, 3 + n):
    tr.step(it, overlap=True)
t1 = time.perf_counter()
tr.sync(); torch.cuda.synchronize()
t2 = time.perf_counter()
print("DM step: host issue %.2f ms, total %.2f ms per step" % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
