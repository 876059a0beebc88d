"""Full-size (112x112x16) late-regime parity of the shipped mixed mode against the all-f16x3 mode, teacher-forced: class-patterned
real clips (base clip per class + 10 % noise, as fixture G12 at 64x64x8), synthetic clips initialised from a real one, `steps`
DM steps; at every step the mixed trainer is put on the f16x3 trainer's state and both take the step.  Writes the per-step
loss / pixel-gradient errors and a summary.  usage: python tools/long_parity.py [classes] [steps] [out.json]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from video_distillation_amd import distill, plan

C = int(sys.argv[1]) if len(sys.argv) > 1 else 10
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
out = sys.argv[3] if len(sys.argv) > 3 else None
dev = torch.device("cuda:0")
geo = plan.NetGeometry(16, 112, 112)
NP, B, mu = 80, 64, 0.5
g = torch.Generator(device=dev).manual_seed(12)
base = torch.randn(C, 1, 16, 3, 112, 112, device=dev, generator=g)
clips = (base + 0.1 * torch.randn(C, NP, 16, 3, 112, 112, device=dev, generator=g)).reshape(C * NP, 16, 3, 112, 112)
pool = distill.RealPool(clips, [NP] * C, [c * NP for c in range(C)])
syn0 = clips[::NP].clone()


def trainer(**kw):
    be = distill.HipBackend(geo, dev, chunk=4096, **kw)
    return distill.DMTrainer(be, pool, C, 1, B, lr_img=20.0, momentum=mu, image_syn=syn0.clone())


ta = trainer(prec_real="f16x3", prec_syn="f16x3", prec_bwd="f16x3")
bwd = os.environ.get("VD_LP_BWD", "f16")          # input-gradient operands of the mixed trainers
tb = trainer(prec_bwd=bwd)
tv = trainer(prec_bwd=bwd); tv.be.dither_enabled = False
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())        # noqa: E731
rec = {"dither": {"loss": [], "grad": []}, "value_pass": {"loss": [], "grad": []}, "feature_gap": []}
for it in range(steps):
    state = (ta.image_syn.clone(), ta.buf.clone(), ta.steps_done)
    la = float(ta.step(it)); ta.sync()
    ga = ta.buf - mu * state[1] if it > 0 else ta.buf.clone()
    for tr, key in ((tb, "dither"), (tv, "value_pass")):
        tr.image_syn.copy_(state[0]); tr.buf.copy_(state[1]); tr.steps_done = state[2]
        lt = float(tr.step(it)); tr.sync()
        gt = tr.buf - mu * state[1] if it > 0 else tr.buf.clone()
        rec[key]["loss"].append(abs(lt / la - 1)); rec[key]["grad"].append(rel(gt, ga))
    rec["feature_gap"].append(la)
summ = {k: {"loss_rel_max": max(v["loss"]), "loss_rel_median": float(np.median(v["loss"])), "grad_rel_l2_max": max(v["grad"]),
            "grad_rel_l2_median": float(np.median(v["grad"]))} for k, v in rec.items() if isinstance(v, dict)}
summ["config"] = "C=%d classes x (64 real + 1 syn) clips 112x112x16, %d teacher-forced steps, class-patterned pool (10 %% noise), lr_img 20" % (C, steps)
summ["loss_first_last"] = [rec["feature_gap"][0], rec["feature_gap"][-1]]
print(json.dumps(summ, indent=1))
if out:
    json.dump({"summary": summ, "per_step": rec}, open(out, "w"))
