"""Host issue time vs device time of one DC step (GMTrainer, class lanes): is the step launch bound?
usage: python tools/dc_host_time.py [classes] [ipc]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_distillation_amd import distill, plan  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 51
ipc = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
geo = plan.NetGeometry(16, 112, 112)
pool = distill.RealPool.synthetic(C, list(range(C)), 70, geo, dev)
ops = distill.HipGMOps(dev, "ours")
tr = distill.GMTrainer(ops, pool, geo, C, ipc, 64, lr_img=0.1)
for it in range(2):
    tr.step(it)
torch.cuda.synchronize()
for it in range(2, 5):
    t0 = time.perf_counter()
    tr.step(it)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("lanes %s: host issue %.1f ms, total %.1f ms" % (os.environ.get("VD_GM_LANES", "8"), (t1 - t0) * 1e3, (t2 - t0) * 1e3))
