"""Launch each forward layer a few times (for rocprofv3 PMC passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from video_distillation_amd import engine, plan
nclips = int(sys.argv[1]) if len(sys.argv) > 1 else 512
prec = sys.argv[2] if len(sys.argv) > 2 else "f16"
geo = plan.NetGeometry(16, 112, 112)
x = torch.randn(nclips, 16, 3, 112, 112, device="cuda")
params = [torch.randn(s, device="cuda") * 0.02 for s in [(64,3,3,7,7),(64,),(128,64,3,7,7),(128,),(128,128,3,7,7),(128,)]]
# VD_RUN_L1_HILO=1: the real side's shipped configuration -- last level in hi+lo pairs (level 1 emits both planes)
eng = engine.EmbedEngine(geo, prec=prec, chunk=nclips, last_hilo=os.environ.get("VD_RUN_L1_HILO") == "1")
eng.set_weights(params)
for _ in range(3): eng.forward(x)
torch.cuda.synchronize()
