"""Per-layer timing of the embed forward at full resolution (one class worth of clips)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_distillation_amd import engine, plan, hip
import ctypes
nclips = int(sys.argv[1]) if len(sys.argv) > 1 else 128
precs = sys.argv[2].split(",") if len(sys.argv) > 2 else ["f16", "f16x3"]
geo = plan.NetGeometry(16, 112, 112)
dims = geo.layer_dims()
x = torch.randn(nclips, 16, 3, 112, 112, device="cuda")
params = [torch.randn(s, device="cuda") * 0.02 for s in [(64,3,3,7,7),(64,),(128,64,3,7,7),(128,),(128,128,3,7,7),(128,)]]
macs = [d[1]*d[5]*d[6]*d[7]*d[0]*147 for d in dims]
for prec in precs:
    eng = engine.EmbedEngine(geo, prec=prec, chunk=nclips)
    eng.set_weights(params)
    for _ in range(2): eng.forward(x)
    torch.cuda.synchronize()
    # time whole forward
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record(); 
    for _ in range(3): eng.forward(x)
    ev[1].record(); torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 3
    fl = 2*sum(macs)*nclips
    print("%s: forward %d clips %.2f ms  -> %.1f TFLOP/s algorithmic (%.1f%% of 2.5PF)" % (prec, nclips, ms, fl/ms/1e9, fl/ms/1e9/2500*100))
    # per-layer: re-run individual stages
    L = hip.lib(); st = hip.stream_ptr(eng.device)
    g = geo
    import numpy as np
    n_slots0 = nclips*16*3*112*15
    slots0 = eng._buf("slots0", (eng.planes, n_slots0, 8), torch.int16)
    n1 = nclips*int(np.prod(eng.fwd[0].plan.out_shape[:-1])); act1 = eng._buf("act1", (eng.planes, n1, 8), torch.int16)
    n2 = nclips*int(np.prod(eng.fwd[1].plan.out_shape[:-1])); act2 = eng._buf("act2", (eng.planes, n2, 8), torch.int16)
    feats = torch.empty(nclips, 2048, device="cuda")
    def t(fn, reps=3):
        fn(); torch.cuda.synchronize()
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b)/reps
    w = eng._weights
    lo = slots0[1] if eng.planes == 2 else None
    tp = t(lambda: L.vd_pix2rows(hip.ptr(x), None, ctypes.c_int64(nclips), 16, 112, 112, hip.ptr(slots0[0]), hip.ptr(lo), eng.prec, st))
    t0 = t(lambda: eng.fwd[0].run(slots0, n_slots0, w[1], act1.data_ptr(), n1, None, nclips))
    t1 = t(lambda: eng.fwd[1].run(act1, n1, w[3], act2.data_ptr(), n2, None, nclips))
    t2 = t(lambda: eng.fwd[2].run(act2, n2, w[5], feats.data_ptr(), 0, None, nclips))
    print("   pix2slots %.2f ms | L0 %.2f ms (%.0f TF) | L1 %.2f ms (%.0f TF) | L2 %.2f ms (%.0f TF)" % (
        tp, t0, 2*macs[0]*nclips/t0/1e9, t1, 2*macs[1]*nclips/t1/1e9, t2, 2*macs[2]*nclips/t2/1e9))
