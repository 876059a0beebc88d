"""A/B of the first-level single-pass kernel's K loops on one box: VD_L0_BREG=4 (one LDS read per MFMA) against 5 (frame-sharing:
every A fragment read once for the tiles it serves) and 0 (generic tile-program kernel), the same frame-tile program, alternating
launches of `clips` clips 112x112x16 (default 3200 = one bench step's real side).  usage: python tools/l0_ab.py [clips] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from video_distillation_amd import engine, plan, hip
import ctypes
nclips = int(sys.argv[1]) if len(sys.argv) > 1 else 3200
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
geo = plan.NetGeometry(16, 112, 112)
params = [torch.randn(s, device="cuda") * 0.05 for s in [(64, 3, 3, 7, 7), (64,), (128, 64, 3, 7, 7), (128,), (128, 128, 3, 7, 7), (128,)]]
engs = {}
variants = tuple(sys.argv[3].split(",")) if len(sys.argv) > 3 else ("5", "4", "0")
for v in variants:
    os.environ["VD_L0_BREG"] = v
    e = engine.EmbedEngine(geo, prec="f16", chunk=nclips, ntw0=1)
    e.set_weights(params)
    engs[v] = e
e5 = engs["5"]
L = hip.lib(); st = hip.stream_ptr(e5.device)
x = torch.randn(64, 16, 3, 112, 112, device="cuda")
n_slots0 = nclips * 16 * 3 * 112 * 15
slots0 = torch.empty((1, n_slots0, 8), dtype=torch.int16, device="cuda")
per = 64 * 16 * 3 * 112 * 15
for k in range(0, nclips, 64):       # 64 distinct clips, repeated
    m = min(64, nclips - k)
    L.vd_pix2rows(hip.ptr(x), None, ctypes.c_int64(m), 16, 112, 112, ctypes.c_void_p(slots0.data_ptr() + k // 64 * per * 16), None, e5.prec, st)
n1 = nclips * int(np.prod(e5.fwd[0].plan.out_shape[:-1]))
outs = {v: torch.empty((1, n1, 8), dtype=torch.int16, device="cuda") for v in engs}
times = {v: [] for v in engs}
for r in range(reps + 2):
    for v, e in engs.items():
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record()
        e.fwd[0].run(slots0, n_slots0, e._weights[1], outs[v].data_ptr(), n1, None, nclips)
        b.record(); torch.cuda.synchronize()
        if r >= 2:
            times[v].append(a.elapsed_time(b))
fl = 2.0 * e5.fwd[0].plan.meta["macs_per_unit"] * nclips
for v in engs:
    t = np.array(times[v])
    print("VD_L0_BREG=%s: %d clips  median %.3f ms  min %.3f  max %.3f  -> %.0f TFLOP/s algorithmic (%.3f of 2.5 PF)  bitwise==generic: %s" % (
        v, nclips, np.median(t), t.min(), t.max(), fl / np.median(t) / 1e9, fl / np.median(t) / 1e9 / 2500, bool(torch.equal(outs[v], outs["0"])) if "0" in outs else None))
