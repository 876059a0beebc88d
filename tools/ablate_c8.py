"""Ablation of the fp8-corrected last level (VD_PREC_F16C8, position tiles) with the dbg build: what a launch of 3200 clips costs
without its patch DMA (4), with one cached B fragment (16), without the A-image conversions (0x400), the correction products
(0x800), the low-part reads (0x1000), the B-image conversion (0x2000), without its epilogue (1), without its K loop (2).  Results are garbage; times are not."""
import sys, os
os.environ.setdefault("VD_LIB_VARIANT", "dbg")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_distillation_amd import engine, plan
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3200
geo = plan.NetGeometry(16, 112, 112)
eng = engine.EmbedEngine(geo, prec="f16", chunk=n, last_hilo="c8")
params = [torch.randn(s, device="cuda") * 0.02 for s in [(64,3,3,7,7),(64,),(128,64,3,7,7),(128,),(128,128,3,7,7),(128,)]]
eng.set_weights(params)
dp = eng.fwd2x
pl = dp.plan
per2 = 16 * 8 * 7 * 7          # 16-byte slots of one clip's level-1 output (16 chunks x 8 frames x 7 x 7)
act2 = (torch.randn(2, n * per2, 8, device="cuda") * 0.5).to(torch.float16).view(torch.int16)
act2[1].zero_()
feats = torch.empty(n, geo.num_feat, device="cuda")
print("program", pl.name, "epi", pl.epi, "S", pl.S, "ncl", pl.ncl, "boxes", pl.nbox, "lds slots", pl.lds_slots)
DBGS = [0, 4, 16, 16 | 0x3c00, 0x400, 0x800, 0x1000, 0x2000, 0x400 | 0x2000, 0x800 | 0x400 | 0x1000 | 0x2000, 4 | 16, 1, 2, 0]
res = {}
for rnd in range(3):
    for dbg in DBGS:
        dp.params.dbg = dbg
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        dp.run(act2, n * per2, params[5], feats.data_ptr(), 0, None, n, out_scale=eng.c8_scales)
        e1.record(); torch.cuda.synchronize()
        res.setdefault(dbg, []).append(e0.elapsed_time(e1))
for dbg in dict.fromkeys(DBGS):
    print("dbg 0x%04x: %.3f ms" % (dbg, min(res[dbg])))
