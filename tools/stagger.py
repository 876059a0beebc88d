import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_distillation_amd import engine, plan
nclips = 512
geo = plan.NetGeometry(16, 112, 112)
x = torch.randn(nclips, 16, 3, 112, 112, device="cuda")
params = [torch.randn(s, device="cuda") * 0.02 for s in [(64,3,3,7,7),(64,),(128,64,3,7,7),(128,),(128,128,3,7,7),(128,)]]
eng = engine.EmbedEngine(geo, prec="f16", chunk=nclips)
eng.set_weights(params)
eng.forward(x); torch.cuda.synchronize()
res = {}
for rnd in range(3):
    for stg in (0, 1, 2, 3, 4, 6, 8, 12):
        for dp in eng.fwd: dp.params.stagger = stg
        eng.profile = []
        eng.forward(x); torch.cuda.synchronize()
        for name, n, a, b in eng.profile:
            res.setdefault((name, stg), []).append(a.elapsed_time(b))
for name in ("fwd0", "fwd1", "fwd2"):
    print(name, " ".join("s%d=%.2f" % (d, min(res[(name, d)])) for d in (0, 1, 2, 3, 4, 6, 8, 12)))
