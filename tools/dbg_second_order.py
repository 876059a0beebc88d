import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_gpu_train import _oracle_vjp, _rel, R
from video_distillation_amd import plan, train
T, H, W, K = 8, 64, 64, 5
g = torch.Generator().manual_seed(35)
x = R.standardise_batch(torch.randn(3, T, 3, H, W, generator=g))
labels = torch.randint(0, K, (3,), generator=g)
params = R.init_params(805, 3, K)
te = train.GradMatchEngine(plan.NetGeometry(T, H, W), K, (2, 1, 1), "cuda:0")
mask = (torch.rand(3, te.C, te.Tp, generator=g) < 0.5).float() * 2.0
vv = [torch.randn(p.shape, generator=g) for p in params]
pc = [p.cuda() for p in params]
for use_mask in (False, True):
    for which, keep in (("head", (6, 7)), ("w2", (4,)), ("b2", (5,)), ("w1", (2,)), ("b1", (3,)), ("w0", (0,)), ("b0", (1,))):
        v = [t if i in keep else torch.zeros_like(t) for i, t in enumerate(vv)]
        res = []
        for b in range(3):
            mb = mask[b:b + 1] if use_mask else None
            want, _ = _oracle_vjp(x[b:b + 1], labels[b:b + 1], params, mb, v)
            _, _, gw, state = te.param_grads(x[b:b + 1].cuda(), labels[b:b + 1].cuda(), pc, None if mb is None else mb.cuda())
            dx = te.vjp(state, [t.cuda() for t in v], pc)
            res.append(_rel(dx, want))
        print("mask", use_mask, which, ["%.1e" % e for e in res], flush=True)
