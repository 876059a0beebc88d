"""How often a forward's pooling decisions differ from the fp64 oracle's, for the HIP fp16 hi+lo forward and for the fp32 oracle
(the reference's arithmetic, networks.py:747-751 on a CPU) on the same clips and weights: the discrete events that dominate every
gradient comparison of the twice-differentiable passes.  Also the features' relative distance from fp64.
   python tools/flip_rate.py [clips] [frames] [size] [seeds]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oracle import ref_cpu as R
from tests import argmax_tools as A
from video_distillation_amd import engine, plan

nclips = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8
S = int(sys.argv[3]) if len(sys.argv) > 3 else 64
seeds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
torch.set_num_threads(min(32, os.cpu_count() or 1))
geo = plan.NetGeometry(T, S, S)
tot = {"hip": [0, 0, 0], "fp32": [0, 0, 0], "windows": [0, 0, 0]}
for seed in range(seeds):
    params = R.init_params(100 + seed, 3, 50)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(nclips, T, 3, S, S, generator=g)
    c64, c32 = [], []
    with torch.no_grad():
        f64 = R.feature_layers(x.double().permute(0, 2, 1, 3, 4), [p.double() for p in params[:6]], collect=c64)
        f32 = R.feature_layers(x.permute(0, 2, 1, 3, 4), params[:6], collect=c32)
    eng = engine.EmbedEngine(geo, prec="f16x3", device="cuda:0", chunk=1 << 30)
    eng.set_weights([p.cuda() for p in params[:6]])
    feats, saved = eng.forward(x.cuda(), keep=True)
    (_, nb, am0, am1, am2), = saved
    dec = A.compare_decisions(x, params, (am0, am1, am2))
    f64f = f64.reshape(nclips, -1)
    rel = lambda a: float((a.double().cpu().reshape(nclips, -1) - f64f).norm() / f64f.norm())
    line = "seed %d: features vs fp64: HIP f16x3 %.2e, fp32 oracle %.2e | flips vs fp64 per level (HIP / fp32 oracle / windows):" % (seed, rel(feats), rel(f32))
    for li, (_, pool) in enumerate(R.LAYER_SPECS):
        a64, m64, t64 = A.oracle_windows(c64[3 * li], pool[0])
        a32, m32, t32 = A.oracle_windows(c32[3 * li].double(), pool[0])
        d64, d32 = t64 <= 0, t32 <= 0
        mism = (d64 != d32) | ((~d64) & (~d32) & (a64 != a32))
        tot["hip"][li] += dec[li]["mismatch"]; tot["fp32"][li] += int(mism.sum()); tot["windows"][li] += int(a64.numel())
        line += "  L%d %d / %d / %d" % (li, dec[li]["mismatch"], int(mism.sum()), int(a64.numel()))
    print(line)
print("total flips HIP %s, fp32 oracle %s, windows %s -> HIP / fp32 per level: %s" % (
    tot["hip"], tot["fp32"], tot["windows"], ["%.2f" % (h / max(f, 1)) for h, f in zip(tot["hip"], tot["fp32"])]))
