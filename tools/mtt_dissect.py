"""ONE student step of configuration 5 (400 classes, 256 clips 64x64x8) taken apart: the first-order parameter gradient g, and for
several adjoint directions V the second-order products xbar = d<V, g>/dx and Hv = d<V, g>/dtheta of train.GradMatchEngine, each
against the same quantities from fp64 autograd of the oracle (oracle.ref_cpu.convnet3d_logits + F.cross_entropy,
create_graph=True: what distill_baseline.py:243-262 differentiates), next to the fp32 oracle's own distance from fp64.
Directions: the real one of an MTT iteration's last step (-lr * 2 (theta - target) / |theta0 - target|^2), and that direction
restricted to one parameter group at a time -- xbar is linear in V, so the restricted runs say WHICH path carries an error.
   python tools/mtt_dissect.py [--modes f16x3,bf16x3] [--classes 400] [--batch 256] [--out gpurun_out/mtt_dissect.json]"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from oracle import ref_cpu as R
from tests import argmax_tools as A
from video_distillation_amd import distill, networks, plan

ap = argparse.ArgumentParser()
ap.add_argument("--modes", default="f16x3,bf16x3")
ap.add_argument("--classes", type=int, default=400)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--groups", default="all,w0,w1,w2,head,biases")
ap.add_argument("--out", default=None)
ap.add_argument("--routed", action="store_true", help="the oracles take the pooling decisions the first mode's HIP forward recorded (arithmetic only)")
args = ap.parse_args()
C, B, T, S, lr = args.classes, args.batch, 8, 64, 0.01
geo = plan.NetGeometry(T, S, S)
g = torch.Generator().manual_seed(77)
theta = R.init_params(5050, 3, C)
target = [q + 0.02 * q.abs().mean() * torch.randn(q.shape, generator=g) for q in theta]
dist0 = sum(float(((a - b) ** 2).sum()) for a, b in zip(theta, target))
static = torch.randn(B, 3, S, S, generator=g)
dynamic = torch.randn(B, T, 1, S, S, generator=g)
hal_w = torch.empty(3, 4, 3, 3, 3).uniform_(-0.096, 0.096, generator=g)
hal_b = torch.empty(3).uniform_(-0.096, 0.096, generator=g)
x = R.hallucinator(static, dynamic, hal_w, hal_b).detach()
labels = torch.randperm(C, generator=g)[:B] if B <= C else torch.randint(0, C, (B,), generator=g)
V_all = [(-lr * 2.0 / dist0) * (a - b) for a, b in zip(theta, target)]
GROUPS = {"all": range(8), "w0": [0], "w1": [2], "w2": [4], "head": [6, 7], "biases": [1, 3, 5]}
directions = {name: [v if i in GROUPS[name] else torch.zeros_like(v) for i, v in enumerate(V_all)] for name in args.groups.split(",")}


def rel(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    n = float(b.norm())
    return float((a - b).norm() / n) if n > 0 else float(a.norm())


ROUTES = None


def oracle(dt, V):
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    th = [p.to(dt).clone().requires_grad_(True) for p in theta]
    xx = x.to(dt).clone().requires_grad_(True)
    ce = F.cross_entropy(R.convnet3d_logits(xx, th) if ROUTES is None else A.routed_logits(xx, th, ROUTES), labels)
    gr = torch.autograd.grad(ce, th, create_graph=True)
    s = sum((v.to(dt) * gi).sum() for v, gi in zip(V, gr))
    out = torch.autograd.grad(s, [xx] + th)
    return [gi.detach() for gi in gr], out[0], list(out[1:])


hip = {}
for mode in args.modes.split(","):
    fmt_name, _, target = mode.partition("@")          # "f16x3@32768": the gradient operands' scale target (VD_GRAD_TARGET) for this leg
    os.environ["VD_GRAD_TARGET"] = target or "1024"
    networks.set_precision(match=fmt_name)
    ops = distill.HipMTTOps(geo, C, "cuda:0", dropout_p=0.0, batch_hint=B)
    gr, handle = ops.grads([p.cuda() for p in theta], x.cuda(), labels.cuda())
    if args.routed and ROUTES is None:
        ROUTES = A.routes_from_argmax([a.cpu() for a in handle[0]["am"]], (B, T, 3, S, S), theta)
    gr = [t.cpu() for t in gr]
    for name, V in directions.items():
        dx, hv = ops.hvp(handle, [v.cuda() for v in V])
        hip[(mode, name)] = (gr, dx.cpu(), [t.cpu() for t in hv])
    del ops, handle
    torch.cuda.empty_cache()
t0 = time.time()
jobs = [(name, dt) for name in directions for dt in (torch.float64, torch.float32)]
with ThreadPoolExecutor(max_workers=max(1, min(4, (os.cpu_count() or 1) // 32))) as ex:
    res = list(ex.map(lambda j: oracle(j[1], directions[j[0]]), jobs))
ref = {j: r for j, r in zip(jobs, res)}
t_or = time.time() - t0


out = {"classes": C, "batch": B, "oracle_seconds": t_or, "routed": bool(args.routed), "directions": {}}


def report(gr, xbar, hv, g64, x64, h64):
    per_clip = sorted(rel(xbar[i], x64[i]) for i in range(B))
    chan = [rel(xbar[:, :, c].double().sum(), x64[:, :, c].double().sum()) for c in range(3)]      # sums over clips / frames / pixels per colour channel
    return {"g": [rel(a, b) for a, b in zip(gr, g64)], "xbar_all": rel(xbar, x64), "xbar_clip_median": per_clip[B // 2],
            "xbar_clip_max": per_clip[-1], "xbar_channel_sums": chan, "hv": [rel(a, b) for a, b in zip(hv, h64)]}


fmt = lambda r: "g %s | xbar all %.1e clip med %.1e max %.1e chan-sums %s | hv %s" % (
    " ".join("%.0e" % v for v in r["g"]), r["xbar_all"], r["xbar_clip_median"], r["xbar_clip_max"],
    " ".join("%.0e" % v for v in r["xbar_channel_sums"]), " ".join("%.0e" % v for v in r["hv"]))
for name in directions:
    g64, x64, h64 = ref[(name, torch.float64)]
    rec = {"fp32_oracle": report(*ref[(name, torch.float32)], g64, x64, h64)}
    print("direction %-6s fp32 oracle: %s" % (name, fmt(rec["fp32_oracle"])))
    for mode in args.modes.split(","):
        rec[mode] = report(*hip[(mode, name)], g64, x64, h64)
        print("          %-7s HIP      : %s" % (mode, fmt(rec[mode])))
    out["directions"][name] = rec
path = args.out or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "mtt_dissect.json")
os.makedirs(os.path.dirname(path), exist_ok=True)
json.dump(out, open(path, "w"), indent=1)
print("oracle %.0f s" % t_or)
