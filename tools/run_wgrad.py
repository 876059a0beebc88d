"""One layer's weight-gradient program stand-alone (for timing and PMC passes).
usage: python tools/run_wgrad.py [layer 0|1|2] [nclips] [prec] [reps] [size 112|64]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_distillation_amd import engine

layer = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
prec = sys.argv[3] if len(sys.argv) > 3 else "f16x3"
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
small = len(sys.argv) > 5 and sys.argv[5] == '64'
cin, cout, t, h, w = ([(3, 64, 8, 64, 64), (64, 128, 8, 16, 16), (128, 128, 4, 4, 4)] if small else
                      [(3, 64, 16, 112, 112), (64, 128, 16, 28, 28), (128, 128, 8, 7, 7)])[layer]
op = engine.WgradOp(cin, cout, t, h, w, n, prec, "cuda:0")
pl = op.plan
print("plan %s: box %s, nbox %d, S %d, CC %d, pitch_c %d slots" % (pl.name, pl.meta["box"], pl.nbox, pl.S, pl.CC, pl.types[0].pitch_c))
g = torch.Generator(device="cuda").manual_seed(1)
dt = torch.float16 if prec.startswith("f16") else torch.bfloat16
oh, ow = (h + 6 - 7) // 2 + 1, (w + 6 - 7) // 2 + 1
dy = torch.randn(op.planes, n * (cout // 8) * t * oh * ow, 8, device="cuda", generator=g).to(dt).view(torch.int16)
dw = torch.zeros(cout, cin, 3, 7, 7, device="cuda")
if cin == 3:
    x = torch.randn(n, t, 3, h, w, device="cuda", generator=g)
    run = lambda: op.run(x, True, 0, dy, int(dy[0].numel() // 8), dw)
else:
    x = torch.randn(op.planes, n * (cin // 8) * t * h * w, 8, device="cuda", generator=g).to(dt).view(torch.int16)
    run = lambda: op.run(x, False, int(x[0].numel() // 8), dy, int(dy[0].numel() // 8), dw)
run(); torch.cuda.synchronize()
engine.LAUNCH_PROFILE = []
for _ in range(reps):
    run()
torch.cuda.synchronize()
ms = [a.elapsed_time(b) for (_, _, _, a, b) in engine.LAUNCH_PROFILE]
flop = 2.0 * cout * cin * 147 * t * oh * ow * n
print("wgrad layer %d %s n=%d: conv program %.3f ms (min %.3f) -> %.0f TFLOP/s algorithmic" % (layer, prec, n, sum(ms) / len(ms), min(ms), flop / min(ms) / 1e9))
