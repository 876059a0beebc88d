"""Diagnostic: where a workgroup of the first-level kernel (conv0_breg2_kernel) spends its cycles -- s_memtime sums per phase
over its box walk, from the library built with the dbg hooks.  usage: python tools/stamps_breg2.py [clips]"""
import sys, os
os.environ.setdefault("VD_LIB_VARIANT", "dbg")
os.environ["VD_BREG_DBG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from video_distillation_amd import engine, plan
nclips = int(sys.argv[1]) if len(sys.argv) > 1 else 512
alone = "--alone" in sys.argv         # one workgroup per CU (conv0_breg3 only): the phases without a partner on the SIMDs
geo = plan.NetGeometry(16, 112, 112)
x = torch.randn(nclips, 16, 3, 112, 112, device="cuda")
params = [torch.randn(s, device="cuda") * 0.02 for s in [(64, 3, 3, 7, 7), (64,), (128, 64, 3, 7, 7), (128,), (128, 128, 3, 7, 7), (128,)]]
eng = engine.EmbedEngine(geo, prec="f16", chunk=nclips)
eng.set_weights(params)
eng.forward(x); torch.cuda.synchronize()
dp = eng.fwd[0]
assert dp.breg_ok and dp.breg_variant in (2, 3)
grid = 4096
buf = torch.zeros(grid * 8, dtype=torch.int64, device="cuda")
dp.params.dbg = 8 | (0x100 if alone else 0); dp.params.stamps = buf.data_ptr()
eng.forward(x); torch.cuda.synchronize()
dp.params.dbg = 0
t = buf.cpu().numpy().reshape(grid, 8).astype(np.float64)
t = t[t[:, 6] > 0]
names = ["K loop", "barrier A (partners done with the patch)", "pool + stage", "DMA issue", "slots out", "wait landing + barrier B"] \
    if dp.breg_variant == 2 else ["K loop", "issue row loads of the next box", "pool + stage", "slots out", "barrier (partners done with the patch)",
                                  "wait rows + expand + barrier"]
per_box = t[:, :6] / t[:, 6:7]
print("workgroups %d, boxes per workgroup %.1f; s_memtime units per box (median over workgroups):" % (len(t), np.median(t[:, 6])))
for k, nm in enumerate(names):
    print("  %-44s %8.1f  (p10 %8.1f, p90 %8.1f)" % (nm, np.median(per_box[:, k]), np.percentile(per_box[:, k], 10), np.percentile(per_box[:, k], 90)))
print("  %-44s %8.1f" % ("sum", np.median(per_box.sum(1))), " whole walk / boxes %.1f" % np.median(t[:, 7] / t[:, 6]))
