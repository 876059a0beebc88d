import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import ref_cpu as R
from video_distillation_amd import engine, plan
params = R.init_params(4)
g = torch.Generator().manual_seed(6)
x = torch.randn(3, 8, 3, 64, 64, generator=g)
gf = torch.randn(3, 256, generator=g)
def ref(dt):
    xr = x.to(dt).clone().requires_grad_(True)
    (R.convnet3d_embed(xr, [p.to(dt) for p in params]) * gf.to(dt)).sum().backward()
    return xr.grad.double()
g32, g64 = ref(torch.float32), ref(torch.float64)
def rel(a, b): return float((a-b).norm()/b.norm())
print("fp32 oracle vs fp64: %.3e" % rel(g32, g64))
for prec in ("bf16x3", "f16x3", "f16", "bf16"):
    eng = engine.EmbedEngine(plan.NetGeometry(8, 64, 64), prec=prec)
    eng.set_weights([p.cuda() for p in params])
    _, sv = eng.forward(x.cuda(), keep=True)
    dx = eng.backward(sv, gf.cuda()).double().cpu()
    d = (dx - g64)
    per = [(float(d[i].norm()/g64[i].norm())) for i in range(3)]
    nbad = int(((d.abs() > 1e-3 * g64.abs().max())).sum())
    print(prec, "vs fp64 rel-l2 %.3e  per clip %s  n(|err|>1e-3 max)=%d of %d" % (rel(dx, g64), per, nbad, d.numel()))
