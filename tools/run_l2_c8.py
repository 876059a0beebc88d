"""The real side's last level alone (VD_PREC_F16C8 program of EmbedEngine(last_hilo='c8'), 3200 clips per launch), N launches on random
operands -- for kernel traces and PMC passes of that program by itself (tools/archive_r04/pmc_c8.sh)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_distillation_amd import engine, plan
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3200
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
geo = plan.NetGeometry(16, 112, 112)
eng = engine.EmbedEngine(geo, prec="f16", chunk=n, last_hilo="c8")
params = [torch.randn(s, device="cuda") * 0.02 for s in [(64,3,3,7,7),(64,),(128,64,3,7,7),(128,),(128,128,3,7,7),(128,)]]
eng.set_weights(params)
dp, pl = eng.fwd2x, eng.fwd2x.plan
per2 = 16 * 8 * 7 * 7
act2 = (torch.randn(2, n * per2, 8, device="cuda") * 0.5).to(torch.float16).view(torch.int16)
act2[1].zero_()
feats = torch.empty(n, geo.num_feat, device="cuda")
ts = []
for r in range(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    dp.run(act2, n * per2, params[5], feats.data_ptr(), 0, None, n, out_scale=eng.c8_scales)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
print("program %s epi %d S %d ncl %d: %d clips per launch, ms per launch: %s" % (pl.name, pl.epi, pl.S, pl.ncl, n, " ".join("%.3f" % t for t in ts)))
