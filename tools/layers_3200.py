import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_distillation_amd import engine, plan, distill
n = 3200
geo = plan.NetGeometry(16, 112, 112)
pool = torch.randn(64, 16, 3, 112, 112, device="cuda")
idx = torch.randint(0, 64, (n,), device="cuda")
e = engine.EmbedEngine(geo, prec="f16", chunk=n); e.set_weights(distill.fresh_network_weights(1, "cuda:0"))
rows = e.pool_rows(pool)
for _ in range(2): e.forward(pool, index=idx, rows=rows)
e.profile = []
for _ in range(5): e.forward(pool, index=idx, rows=rows)
torch.cuda.synchronize()
import numpy as np
for nm in ("fwd0", "fwd1", "fwd2"):
    print(nm, "%.3f ms" % np.median([a.elapsed_time(b) for (k, c, a, b) in e.profile if k == nm]), end=" | ")
print()
