"""Configuration 5 at the benchmark's unroll length: one distill.S2DMTTTrainer iteration ("MTT+Ours": 400 classes, 256-clip
hallucinator-composed student batches 64x64x8) with syn_steps = 10 against oracle.ref_cpu.mtt_step chained through the
oracle's hallucinator, in fp32 AND in fp64 -- the HIP path is judged by its distance from the fp64 result next to the fp32
oracle's own distance from it (the reference computes in fp32: distill_baseline.py:231-262, distill_s2d_ms.py:236-300).
   python tools/parity_mtt10.py [steps] [--modes f16x3,bf16x3] [--classes 400] [--batch 256] [--out gpurun_out/parity_mtt10.json]
One pair of oracle evaluations serves every operand mode of train.GradMatchEngine (VD_PREC_MATCH).

Over ten unrolled steps the FREE comparison is decided by pooling near-ties (one window routed the other way in an early step
moves the median memory row by percents; whether HIP, the fp32 oracle or neither has one depends on the seed), so the tool also
evaluates the oracle ROUTED by the decisions the HIP forwards recorded (tests/argmax_tools.py: mtt_step_routed) in fp64 and
fp32 -- the same piecewise-linear function on both sides: what is left is arithmetic -- and counts the windows the fp64 values
would have routed otherwise (all must be near-ties).  --no-free skips the free oracles."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import ref_cpu as R
from tests import argmax_tools as A
from video_distillation_amd import distill, networks, plan

ap = argparse.ArgumentParser()
ap.add_argument("steps", nargs="?", type=int, default=10)
ap.add_argument("--modes", default="f16x3,bf16x3")
ap.add_argument("--classes", type=int, default=400)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--out", default=None)
ap.add_argument("--threads", type=int, default=32)
ap.add_argument("--seed", type=int, default=505)
ap.add_argument("--no-free", action="store_true")
args = ap.parse_args()
steps, C, batch = args.steps, args.classes, args.batch
vpc, spc, dpc, T, S, syn_lr = 1, 2, 2, 8, 64, 0.01


def rel(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / b.norm())


geo = plan.NetGeometry(T, S, S)
g = torch.Generator().manual_seed(args.seed)
start = R.init_params(10 * args.seed, 3, C)
target = [q + 0.02 * q.abs().mean() * torch.randn(q.shape, generator=g) for q in start]
static = torch.randn(C * spc, 3, S, S, generator=g)
dynamic = torch.randn(C, dpc, T, 1, S, S, generator=g)
hal_w = torch.empty(3, 4, 3, 3, 3).uniform_(-0.096, 0.096, generator=g)
hal_b = torch.empty(3).uniform_(-0.096, 0.096, generator=g)
rng = np.random.default_rng(args.seed // 9 - 1 if args.seed == 505 else args.seed + 1)
chunks = [torch.as_tensor(rng.permutation(C)[:batch]) for _ in range(steps)]
draws = [(rng.integers(0, 2, batch), rng.integers(0, 2, batch)) for _ in range(steps)]

hip_runs = {}
tr = None
for mode in args.modes.split(","):
    networks.set_precision(match=mode)
    ops = distill.HipMTTOps(geo, C, "cuda:0", dropout_p=0.0, batch_hint=batch)
    tr = distill.S2DMTTTrainer(ops, C, vpc, spc, dpc, static.cuda(), dynamic.cuda(), hal_w.cuda(), hal_b.cuda(), syn_lr=syn_lr,
                               lr_dynamic=0.01, lr_hal=0.01, lr_lr=1e-5, syn_steps=steps, batch_syn=batch, expert_epochs=1, max_start_epoch=1)
    tr.draws = draws
    tr.keep_tape = True
    t0 = time.time()
    grand_hip = float(tr.step(0, [start, target], start_epoch=0, index_chunks=chunks, update=False))
    torch.cuda.synchronize()
    g_dyn, g_w, g_b, _, g_lr = tr.last_grads
    hip_runs[mode] = dict(grand=grand_hip, g_dyn=g_dyn.cpu(), g_w=g_w.cpu(), g_b=g_b.cpu(), g_lr=float(g_lr), seconds=time.time() - t0,
                          routes=[A.routes_from_argmax([a.cpu() for a in handle[0]["am"]], (batch, T, 3, S, S), start)
                                  for _, _, handle, _ in tr.last_tape])
    tr.last_tape = None
    del ops
    torch.cuda.empty_cache()

def oracle(dt, routes=None, stats=None):
    torch.set_num_threads(min(args.threads, os.cpu_count() or 1))
    dyn = dynamic.reshape(C * dpc, T, 1, S, S).to(dt).clone().requires_grad_(True)
    w, b = hal_w.to(dt).clone().requires_grad_(True), hal_b.to(dt).clone().requires_grad_(True)
    xs, labels = [], []
    for s, these in enumerate(chunks):
        label, sidx, didx = tr.indices(these, s, 0)
        xs.append(R.hallucinator(static.to(dt)[sidx], dyn[didx], w, b))
        labels.append(label)
    x_all = torch.cat(xs)
    ch = [torch.arange(s * batch, (s + 1) * batch) for s in range(steps)]
    if routes is None:
        grand, gx, glr = R.mtt_step([q.to(dt) for q in start], [q.to(dt) for q in target], x_all.detach(), torch.cat(labels), syn_lr, ch, dtype=dt)
    else:
        grand, gx, glr = A.mtt_step_routed([q.to(dt) for q in start], [q.to(dt) for q in target], x_all.detach(), torch.cat(labels), syn_lr, ch,
                                           routes, dt, stats)
    gd, gw, gb = torch.autograd.grad(x_all, [dyn, w, b], grad_outputs=gx)
    return dict(grand=float(grand), g_dyn=gd, g_w=gw, g_b=gb, g_lr=float(glr))


from concurrent.futures import ThreadPoolExecutor


def against(r, o):
    rows = [i for i in range(C * dpc) if float(o["g_dyn"][i].abs().sum()) > 0]
    per_row = sorted(rel(r["g_dyn"][i], o["g_dyn"][i]) for i in rows)
    return {"grand_loss_rel": abs(r["grand"] / o["grand"] - 1), "d_syn_lr_rel": abs(r["g_lr"] / o["g_lr"] - 1),
            "g_dynamic_rel_l2": rel(r["g_dyn"], o["g_dyn"]), "g_dynamic_per_row_median": per_row[len(per_row) // 2],
            "g_dynamic_per_row_p90": per_row[int(0.9 * len(per_row))], "g_dynamic_per_row_max": per_row[-1],
            "g_hal_w_rel_l2": rel(r["g_w"].reshape(-1), o["g_w"].reshape(-1)), "g_hal_b_rel_l2": rel(r["g_b"], o["g_b"])}


short = lambda d: " ".join("%s %.2e" % (k.replace("g_dynamic_", "dyn_").replace("_rel_l2", "").replace("_rel", ""), v) for k, v in d.items())
t1 = time.time()
modes = list(hip_runs)
lead = modes[0]
same_routes = {m: all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for ra, rb in zip(hip_runs[m]["routes"], hip_runs[lead]["routes"])
                      for a, b in zip(ra, rb)) for m in modes}
stats64 = []
jobs = {"routed64": lambda: oracle(torch.float64, hip_runs[lead]["routes"], stats64), "routed32": lambda: oracle(torch.float32, hip_runs[lead]["routes"])}
if not args.no_free:
    jobs.update({"free64": lambda: oracle(torch.float64), "free32": lambda: oracle(torch.float32)})
with ThreadPoolExecutor(max_workers=max(1, min(4, (os.cpu_count() or 1) // 32))) as ex:
    futs = {k: ex.submit(f) for k, f in jobs.items()}
    res = {k: f.result() for k, f in futs.items()}
t2 = time.time()
out = {"syn_steps": steps, "classes": C, "batch": batch, "seed": args.seed, "oracle_seconds_wall": t2 - t1,
       "command": "python tools/parity_mtt10.py %d --modes %s --classes %d --batch %d --seed %d" % (steps, args.modes, C, batch, args.seed), "modes": {}}
# ---- arithmetic: everybody on the decisions the HIP forwards recorded ----
ref_r = against(res["routed32"], res["routed64"])
mism = [[d["mismatch"] for d in st] for st in stats64]
far = [[d["not_near_tie"] for d in st] for st in stats64]
out["routed"] = {"decisions_of": lead, "fp32_arithmetic_vs_fp64": ref_r, "windows_fp64_would_route_otherwise_per_step_and_level": mism,
                 "of_those_no_near_tie": far, "windows_per_level": [d["windows"] for d in stats64[0]]}
print("syn_steps %d, C %d, batch %d, seed %d  (oracles %.0f s wall)" % (steps, C, batch, args.seed, t2 - t1), flush=True)
print("ROUTED by the HIP forwards' decisions (%d windows differ from the fp64 values' own choice over the %d steps, %d of them no near-tie):" % (
    sum(map(sum, mism)), steps, sum(map(sum, far))))
print("  fp32 arithmetic vs fp64: %s" % short(ref_r))
for m in modes:
    if not same_routes[m]:
        print("  (%s recorded other decisions than %s: its routed comparison is skipped)" % (m, lead))
        continue
    a = against(hip_runs[m], res["routed64"])
    a["ratio_to_fp32_arithmetic"] = {k: (a[k] / ref_r[k] if ref_r[k] > 0 else None) for k in ref_r}
    out["modes"].setdefault(m, {})["routed_vs_fp64"] = a
    print("  HIP %-7s vs fp64: %s" % (m, " ".join("%s %.2e (x%.1f)" % (
        k.replace("g_dynamic_", "dyn_").replace("_rel_l2", "").replace("_rel", ""), a[k], a[k] / ref_r[k] if ref_r[k] > 0 else float("nan")) for k in ref_r)))
# ---- free: every side takes its own decisions ----
if not args.no_free:
    o64 = res["free64"]
    ref = against(res["free32"], o64)
    rows64 = [i for i in range(C * dpc) if float(o64["g_dyn"][i].abs().sum()) > 0]
    untouched = sorted(set(range(C * dpc)) - set(rows64))
    out["oracle_fp32_vs_fp64"] = ref
    out["touched_rows"] = len(rows64)
    print("FREE (every side its own decisions):")
    print("  fp32 oracle vs fp64 oracle: %s" % short(ref))
    for m in modes:
        a = against(hip_runs[m], o64)
        a["ratio_to_fp32_oracle"] = {k: (a[k] / ref[k] if ref[k] > 0 else None) for k in ref}
        a["untouched_rows_exactly_zero"] = all(float(hip_runs[m]["g_dyn"][i].abs().sum()) == 0.0 for i in untouched)
        out["modes"].setdefault(m, {})["free_vs_fp64"] = a
        print("  HIP %-7s vs fp64 oracle: %s" % (m, " ".join("%s %.2e (x%.1f)" % (
            k.replace("g_dynamic_", "dyn_").replace("_rel_l2", "").replace("_rel", ""), a[k], a[k] / ref[k] if ref[k] > 0 else float("nan")) for k in ref)))
for m in modes:
    out["modes"].setdefault(m, {}).update(hip_seconds=hip_runs[m]["seconds"], grand_loss=hip_runs[m]["grand"])
path = args.out or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_mtt%d.json" % steps)
os.makedirs(os.path.dirname(path), exist_ok=True)
json.dump(out, open(path, "w"), indent=1)
