"""Diagnostic: per-workgroup phase timeline (s_memtime) of a forward layer."""
import sys, os
os.environ.setdefault("VD_LIB_VARIANT", "dbg")   # needs the build with the dbg hooks (hip.build(debug_hooks=True))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from video_distillation_amd import engine, plan
nclips = 512
li = int(sys.argv[1]) if len(sys.argv) > 1 else 0
alone = "--alone" in sys.argv        # one workgroup per CU: a workgroup's phases without a partner on its SIMDs
if li == 0:
    os.environ["VD_L0_BREG"] = "0"   # (the generic tile-program kernel carries these stamps; the first-level kernel: tools/stamps_l0.py)
geo = plan.NetGeometry(16, 112, 112)
x = torch.randn(nclips, 16, 3, 112, 112, device="cuda")
params = [torch.randn(s, device="cuda") * 0.02 for s in [(64,3,3,7,7),(64,),(128,64,3,7,7),(128,),(128,128,3,7,7),(128,)]]
prec = [a for a in sys.argv[2:] if not a.startswith("--")]
prec = prec[0] if prec else "f16"
if prec != "f16":
    nclips = 64
    x = x[:nclips]
eng = engine.EmbedEngine(geo, prec=prec, chunk=nclips)
eng.set_weights(params)
eng.forward(x); torch.cuda.synchronize()
dp = eng.fwd[li]
grid = dp.plan.grid(nclips)
buf = torch.zeros(grid * 8, dtype=torch.int64, device="cuda")
dp.params.dbg = 8 | (0x100 if alone else 0); dp.params.stamps = buf.data_ptr()
eng.forward(x); torch.cuda.synchronize()
dp.params.dbg = 0
t = buf.cpu().numpy().reshape(grid, 8).astype(np.float64)
ok = t[:, 0] > 0
print("valid WGs", ok.sum(), "of", grid)
t = t[ok]
t0 = t[:, 0].min()
d = np.diff(t[:, :8], axis=1)   # phases: setup, dma-issue, dma-wait, kloop(first chunk + rest), epilogue
print("grid", grid, "kernel span (cycles of 100MHz*?):", t[:, 7].max() - t0)
names = ["setup(prologue)", "dma issue", "dma wait+barrier", "K loop (all chunks)", "epi: wait other waves", "epi: max+stage+barrier", "epi: stores issue"]
for k, nm in enumerate(names):
    print("%-22s median %8.0f  p10 %8.0f  p90 %8.0f" % (nm, np.median(d[:, k]), np.percentile(d[:, k], 10), np.percentile(d[:, k], 90)))
print("total per WG median", np.median(t[:, 7] - t[:, 0]))


ids = np.arange(grid)[ok]
for x in range(2):
    sel = (ids % 8) == x
    tx = t[sel]
    t0x = tx[:, 0].min(); span = tx[:, 7].max() - t0x
    res = (tx[:, 7] - tx[:, 0]).sum()
    st = np.sort(tx[:, 0] - t0x)
    mid = st[(st > span * 0.3) & (st < span * 0.7)]
    print("XCD-group %d: span %.0f cycles, concurrent WGs %.1f (%.2f per CU), WG starts / 1000 cycles %.2f" % (
        x, span, res / span, res / span / 32, len(mid) / (span * 0.4) * 1000))
