"""In-process ablation of the forward kernels (dbg: 1 = no epilogue, 2 = no K loop, 4 = no patch DMA)."""
import sys, os
os.environ.setdefault("VD_LIB_VARIANT", "dbg")   # needs the build with the dbg hooks (hip.build(debug_hooks=True))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from video_distillation_amd import engine, plan
nclips = int(sys.argv[1]) if len(sys.argv) > 1 else 512
prec = sys.argv[2] if len(sys.argv) > 2 else "f16"
geo = plan.NetGeometry(16, 112, 112)
x = torch.randn(nclips, 16, 3, 112, 112, device="cuda")
params = [torch.randn(s, device="cuda") * 0.02 for s in [(64,3,3,7,7),(64,),(128,64,3,7,7),(128,),(128,128,3,7,7),(128,)]]
eng = engine.EmbedEngine(geo, prec=prec, chunk=nclips)
eng.set_weights(params)
eng.forward(x); torch.cuda.synchronize()
DBGS = [int(v) for v in sys.argv[3].split(',')] if len(sys.argv) > 3 else (0, 1, 2, 4, 3, 5, 6, 7)
res = {}
for rnd in range(3):
    for dbg in DBGS:
        for dp in eng.fwd: dp.params.dbg = dbg
        eng.profile = []
        eng.forward(x); torch.cuda.synchronize()
        for name, n, a, b in eng.profile:
            res.setdefault((name, dbg), []).append(a.elapsed_time(b))
for name in ("fwd0", "fwd1", "fwd2"):
    print(name, " ".join("dbg%d=%.2f" % (d, min(res[(name, d)])) for d in DBGS))
