"""First-layer forward program (conv0_breg_kernel) stand-alone: time per launch and a bitwise comparison of the features between the runs of one call (tools/lib_ab.sh runs it once per library build, VD_LIB_PATH)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_distillation_amd import engine, plan
nclips = int(sys.argv[1]) if len(sys.argv) > 1 else 512
knobs = sys.argv[2].split(",") if len(sys.argv) > 2 else ["1", "0"]
geo = plan.NetGeometry(16, 112, 112)
x = torch.randn(nclips, 16, 3, 112, 112, device="cuda")
params = [torch.randn(s, device="cuda") * 0.02 for s in [(64, 3, 3, 7, 7), (64,), (128, 64, 3, 7, 7), (128,), (128, 128, 3, 7, 7), (128,)]]
eng = engine.EmbedEngine(geo, prec="f16", chunk=nclips)
eng.set_weights(params)
outs, times = {}, {}
for rnd in range(4):
    for k in knobs:
        os.environ["VD_L0_KNOB"] = k
        eng.profile = []
        f = eng.forward(x); torch.cuda.synchronize()
        outs[k] = f
        times.setdefault(k, []).append(min(a.elapsed_time(b) for name, n, a, b in eng.profile if name == "fwd0"))
print("  ".join("knob %s: %.3f ms" % (k, min(times[k])) for k in knobs), " bitwise equal:", all(torch.equal(outs[k], outs[knobs[0]]) for k in knobs))
