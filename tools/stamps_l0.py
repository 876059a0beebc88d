"""Diagnostic: where the two role groups of the first-level kernel (conv0_breg_kernel) spend their cycles -- s_memtime sums per
role phase over a workgroup's box walk, from the library built with the dbg hooks (VD_DBG_HOOKS).
usage: python tools/stamps_l0.py [clips] [VD_L0_BREG variant 4|5]"""
import sys, os
os.environ.setdefault("VD_LIB_VARIANT", "dbg")
os.environ["VD_BREG_DBG"] = "1"
if len(sys.argv) > 2:
    os.environ["VD_L0_BREG"] = sys.argv[2]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from video_distillation_amd import engine, plan
nclips = int(sys.argv[1]) if len(sys.argv) > 1 else 512
geo = plan.NetGeometry(16, 112, 112)
x = torch.randn(nclips, 16, 3, 112, 112, device="cuda")
params = [torch.randn(s, device="cuda") * 0.02 for s in [(64, 3, 3, 7, 7), (64,), (128, 64, 3, 7, 7), (128,), (128, 128, 3, 7, 7), (128,)]]
eng = engine.EmbedEngine(geo, prec="f16", chunk=nclips)
eng.set_weights(params)
eng.forward(x); torch.cuda.synchronize()
dp = eng.fwd[0]
assert dp.breg_ok
grid = 8192
buf = torch.zeros(grid * 8, dtype=torch.int64, device="cuda")
dp.params.dbg = 8; dp.params.stamps = buf.data_ptr()
for _ in range(3):        # (sustained clocks)
    eng.forward(x)
torch.cuda.synchronize()
buf.zero_()
import numpy as _np
n_slots0 = nclips * 16 * 3 * 112 * 15
slots0 = eng._buf("slots0", (eng.planes, n_slots0, 8), torch.int16)
n1 = nclips * int(_np.prod(dp.plan.out_shape[:-1])); act1 = eng._buf("act1", (eng.planes, n1, 8), torch.int16)
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
dp.run(slots0, n_slots0, eng._weights[1], act1.data_ptr(), n1, None, nclips)
e1.record(); torch.cuda.synchronize()
launch_ms = e0.elapsed_time(e1)
dp.params.dbg = 0
t = buf.cpu().numpy().reshape(grid, 8).astype(np.float64)
t = t[t[:, 6] > 0]
names = ["K phase up to the K loop", "K loop (128 MFMAs)", "barrier after the K loop", "row loads issued + pool + stage", "wait rows + expand",
         "slots out + scalars + barrier"]
per_box = t[:, :6] / t[:, 6:7]
print("VD_L0_BREG=%d: role groups %d, boxes per group %.1f; s_memtime units per box (median over groups):" % (dp.breg_variant, len(t), np.median(t[:, 6])))
for k, nm in enumerate(names):
    print("  %-36s %8.1f  (p10 %8.1f, p90 %8.1f)" % (nm, np.median(per_box[:, k]), np.percentile(per_box[:, k], 10), np.percentile(per_box[:, k], 90)))
print("  %-36s %8.1f   K phase %.1f, other phase %.1f" % ("sum = two phases", np.median(per_box.sum(1)), np.median(per_box[:, :3].sum(1)), np.median(per_box[:, 3:].sum(1))),
      " whole walk / boxes %.1f" % np.median(t[:, 7] / t[:, 6]))
# clock: a CU walks (boxes of the launch / CUs) boxes, two per (K phase + other phase) pair of its workgroup
ncu = torch.cuda.get_device_properties(0).multi_processor_count
boxes_per_cu = nclips * dp.plan.nbox / ncu
cyc = boxes_per_cu / 2 * np.median(t[:, 7] / t[:, 6])
print("  launch %.3f ms for %d clips; %.0f boxes per CU x %.0f s_memtime units per box pair / 2 = %.3g units -> %.2f GHz if s_memtime counts shader clocks" % (
    launch_ms, nclips, boxes_per_cu, np.median(t[:, 7] / t[:, 6]), cyc, cyc / launch_ms / 1e6))
