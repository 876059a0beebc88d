"""A/B of the parity-class programs as one launch (VD_MULTI_LAUNCH=1, default) against one launch per program (=0), same process:
input-gradient pass of the embed engine, one evaluate_synset training step, one second-order pass (GradMatchEngine.vjp).
usage: python tools/multi_launch_ab.py [T H W clips]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_distillation_amd import distill, engine, plan, train
T, H, W, n = [int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (8, 64, 64, 256))]
geo = plan.NetGeometry(T, H, W)
K = 10
x = torch.randn(n, T, 3, H, W, device="cuda")
full = distill.fresh_full_network(1, K, "cuda:0")
labels = torch.arange(n, device="cuda") % K
pool = (2, 2, 2) if H > 64 else (2, 1, 1)


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for prec in ("f16x3", "f16"):
    eng = engine.EmbedEngine(geo, prec=prec, chunk=4096, prec_bwd=prec, batch_hint=n)
    eng.set_weights(full[:6])
    gf = torch.randn(n, eng.num_feat, device="cuda")
    f, sv = eng.forward(x, keep=True)
    for mode in ("0", "1"):
        os.environ["VD_MULTI_LAUNCH"] = mode
        print("embed backward %s, %d clips %dx%dx%d, multi=%s: %.3f ms" % (prec, n, H, W, T, mode, timed(lambda: eng.backward(sv, gf))))
te = train.TrainEngine(geo, K, pool, "cuda:0", batch_hint=n)
for mode in ("0", "1"):
    os.environ["VD_MULTI_LAUNCH"] = mode
    print("train step (loss_and_grads) f16x3 multi=%s: %.3f ms" % (mode, timed(lambda: te.loss_and_grads(x, labels, full))))
gm = train.GradMatchEngine(geo, K, pool, "cuda:0", batch_hint=n)
v = [torch.randn_like(p) * 0.1 for p in full]
for mode in ("0", "1"):
    os.environ["VD_MULTI_LAUNCH"] = mode
    _, _, _, state = gm.param_grads(x, labels, full)
    print("second-order pass (vjp + parameter adjoint) bf16x3 multi=%s: %.3f ms" % (mode, timed(lambda: gm.vjp(state, v, full, param_adjoint=True))))
