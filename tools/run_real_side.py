"""The real side of ONE benchmark step, exactly as bench.py launches it (config 2: 50 classes x 64 clips = 3200 clips of
112x112x16 per launch, index gather out of the resident pixel rows, eight dithered operand sets in levels 0 / 1, level 1 also
writing the low plane, the last level in hi+lo pairs), three times and nothing else -- for the rocprofv3 PMC passes
(FETCH_SIZE / WRITE_SIZE in separate runs) behind `roofline.traffic`, and for kernel-trace runs of the same launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_distillation_amd import distill, plan
C, B, per = int(os.environ.get("VD_RS_CLASSES", "50")), 64, 93
dev = torch.device("cuda:0")
geo = plan.NetGeometry(16, 112, 112)
pool = distill.RealPool.synthetic(C, list(range(C)), per, geo, dev, seed=1234, kind="templates", noise=1.5)
be = distill.HipBackend(geo, dev, chunk=C * B)
tr = distill.DMTrainer(be, pool, C, 1, B, lr_img=1.0, momentum=0.5)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for it in range(reps):
    idx = torch.as_tensor(distill.sample_real_indices(it, pool.counts, pool.offsets, B, tr.classes), device=dev)
    be.set_real_weights(be.new_network(seed=it), B)
    tr._real_features(idx)
torch.cuda.synchronize()
print("real side: %d launches of %d clips per level" % (reps, C * B))
