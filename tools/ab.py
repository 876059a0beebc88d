"""In-process A/B of kernel builds: python tools/ab.py tagA tagB ... (libvd_hip_<tag>.so next to the default)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from video_distillation_amd import engine, plan, hip
tags = sys.argv[1:]
nclips = 512
geo = plan.NetGeometry(16, 112, 112)
x = torch.randn(nclips, 16, 3, 112, 112, device="cuda")
params = [torch.randn(s, device="cuda") * 0.02 for s in [(64,3,3,7,7),(64,),(128,64,3,7,7),(128,),(128,128,3,7,7),(128,)]]
eng = engine.EmbedEngine(geo, prec="f16", chunk=nclips)
eng.set_weights(params)
libs = {}
for t in ["default"] + tags:
    path = hip.LIB_PATH if t == "default" else hip.LIB_PATH.replace(".so", "_%s.so" % t)
    L = ctypes.CDLL(path)
    for name in hip.EXPORTS: getattr(L, name).restype = ctypes.c_int
    libs[t] = L
res = {}
for rnd in range(5):
    for t, L in libs.items():
        hip._lib = L
        eng.profile = []
        eng.forward(x); torch.cuda.synchronize()
        for name, n, a, b in eng.profile:
            res.setdefault((t, name), []).append(a.elapsed_time(b))
for t in libs:
    print("%-16s" % t, " ".join("%s med %.3f min %.3f" % (nm, np.median(res[(t, nm)][1:]), min(res[(t, nm)][1:])) for nm in ("fwd0", "fwd1", "fwd2")))
