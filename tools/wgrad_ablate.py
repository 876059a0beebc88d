import os, sys
sys.path.insert(0, "/root/repo")
os.environ["VD_LIB_VARIANT"] = "dbg"
import torch
from video_distillation_amd import engine
layer, n, prec = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
cin, cout, t, h, w = [(3, 64, 16, 112, 112), (64, 128, 16, 28, 28), (128, 128, 8, 7, 7)][layer]
op = engine.WgradOp(cin, cout, t, h, w, n, prec, "cuda:0")
g = torch.Generator(device="cuda").manual_seed(1)
dt = torch.float16 if prec.startswith("f16") else torch.bfloat16
oh, ow = (h - 1) // 2 + 1, (w - 1) // 2 + 1
dy = torch.randn(op.planes, n * (cout // 8) * t * oh * ow, 8, device="cuda", generator=g).to(dt).view(torch.int16)
dw = torch.zeros(cout, cin, 3, 7, 7, device="cuda")
if cin == 3:
    x = torch.randn(n, t, 3, h, w, device="cuda", generator=g)
    run = lambda: op.run(x, True, 0, dy, int(dy[0].numel() // 8), dw)
else:
    x = torch.randn(op.planes, n * (cin // 8) * t * h * w, 8, device="cuda", generator=g).to(dt).view(torch.int16)
    run = lambda: op.run(x, False, int(x[0].numel() // 8), dy, int(dy[0].numel() // 8), dw)
for dbg, what in ((0, "full"), (1, "no epilogue"), (2, "no K loop"), (4, "no patch DMA"), (3, "no K loop, no epilogue"), (7, "nothing"), (16, "cached B"), (17, "cached B, no epilogue")):
    op.dp.params.dbg = dbg
    run(); torch.cuda.synchronize()
    engine.LAUNCH_PROFILE = []
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for (_, _, _, a, b) in engine.LAUNCH_PROFILE]
    print("layer %d %s n=%d  %-26s %.3f ms" % (layer, prec, n, what, min(ms)))
