import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_distillation_amd import distill, plan
ncls = int(sys.argv[1]) if len(sys.argv) > 1 else 7
geo = plan.NetGeometry(16, 112, 112)
dev = torch.device("cuda:0")
be = distill.HipBackend(geo, dev)
pool = distill.RealPool.synthetic(ncls, list(range(ncls)), 93, geo, dev)
tr = distill.DMTrainer(be, pool, ncls, 1, 64, lr_img=1.0)
for it in range(3): tr.step(it, overlap=True)
tr.sync(); torch.cuda.synchronize()
n = 30
t0 = time.perf_counter()
for it in range(3, 3 + n): tr.step(it, overlap=True)
t1 = time.perf_counter()
tr.sync(); torch.cuda.synchronize()
t2 = time.perf_counter()
print("classes %d: host enqueue %.2f ms/step, total %.2f ms/step" % (ncls, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
