import sys, os, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import ref_cpu as R
from video_distillation_amd import engine, plan
def rel(a, b): return float((a.double().cpu() - b.double()).norm() / (b.double().norm() + 1e-30))
params = R.init_params(5)
bad = 0
for geom in [(6, 48, 64), (10, 80, 96), (14, 112, 80), (16, 96, 96), (8, 128, 64), (16, 64, 112), (12, 64, 48)]:
    T, H, W = geom
    try:
        g = torch.Generator().manual_seed(T + H + W)
        x = torch.randn(4, T, 3, H, W, generator=g)
        want = R.convnet3d_embed(x, params)
        gf = torch.randn(want.shape, generator=g)
        xr = x.double().clone().requires_grad_(True)
        (R.convnet3d_embed(xr, [p.double() for p in params]) * gf.double()).sum().backward()
        for prec, hint in (("f16", None), ("f16x3", None), ("f16x3", 4), ("bf16x3", 4)):
            eng = engine.EmbedEngine(plan.NetGeometry(T, H, W), prec=prec, chunk=3, batch_hint=hint)
            eng.set_weights([p.cuda() for p in params])
            f, sv = eng.forward(x.cuda(), keep=True)
            dx = eng.backward(sv, gf.cuda())
            torch.cuda.synchronize()
            e_f = rel(f, want); per = sorted(rel(dx[i], xr.grad[i]) for i in range(4))
            # flip-tolerant criterion: clips without a pooling near-tie agree to operand precision; bf16 pairs (8+8 bits, never used
            # for a forward whose decisions matter: GradMatchEngine routes by an f16x3 forward) flip in most random clips here
            clean = per[0] if prec == "bf16x3" else per[1]
            ok = e_f < (2e-3 if prec == "f16" else 3e-5) and (prec == "f16" or (clean < 1e-4 and per[-1] < 5e-2))
            bad += (not ok)
            print(geom, prec, hint, "fwd %.1e" % e_f, "bwd per-clip", ["%.0e" % v for v in per], "OK" if ok else "FAIL", flush=True)
    except Exception as e:
        bad += 1
        print(geom, "EXC", repr(e)[:200], flush=True)
print("failures:", bad)
