import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_distillation_amd import engine, plan, distill
geo = plan.NetGeometry(16, 112, 112)
n = int(sys.argv[1]); prec = sys.argv[2]
x = torch.randn(n, 16, 3, 112, 112, device="cuda")
w = distill.fresh_network_weights(1, "cuda:0")
eng = engine.EmbedEngine(geo, prec=prec, chunk=512)
eng.set_weights(w)
for dp in eng.fwd: dp.params.persist = 0
f0 = eng.forward(x); torch.cuda.synchronize(); print("persist 0 ok", flush=True)
for g in (1, 4, 16):
    for dp in eng.fwd: dp.params.persist = g
    f1 = eng.forward(x); torch.cuda.synchronize()
    print("persist", g, "equal", bool(torch.equal(f0, f1)), float((f0 - f1).abs().max()), flush=True)
