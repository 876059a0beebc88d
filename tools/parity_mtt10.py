"""Configuration 5 at the benchmark's unroll length: one distill.S2DMTTTrainer iteration ("MTT+Ours": 400 classes, 256-clip
hallucinator-composed student batches 64x64x8) with syn_steps = 10 -- tests/test_gpu_config_geometry.py runs 2 unrolled steps --
against oracle.ref_cpu.mtt_step chained through the oracle's hallucinator (fp32).
   python tools/parity_mtt10.py [steps]      -> gpurun_out/r05_parity_mtt10.json + a summary on stdout"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import ref_cpu as R
from video_distillation_amd import distill, plan

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
C, vpc, spc, dpc, T, S, batch, syn_lr = 400, 1, 2, 2, 8, 64, 256, 0.01


def rel(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / b.norm())


geo = plan.NetGeometry(T, S, S)
g = torch.Generator().manual_seed(505)
start = R.init_params(5050, 3, C)
target = [q + 0.02 * q.abs().mean() * torch.randn(q.shape, generator=g) for q in start]
static = torch.randn(C * spc, 3, S, S, generator=g)
dynamic = torch.randn(C, dpc, T, 1, S, S, generator=g)
hal_w = torch.empty(3, 4, 3, 3, 3).uniform_(-0.096, 0.096, generator=g)
hal_b = torch.empty(3).uniform_(-0.096, 0.096, generator=g)
ops = distill.HipMTTOps(geo, C, "cuda:0", dropout_p=0.0, batch_hint=batch)
tr = distill.S2DMTTTrainer(ops, C, vpc, spc, dpc, static.cuda(), dynamic.cuda(), hal_w.cuda(), hal_b.cuda(), syn_lr=syn_lr,
                           lr_dynamic=0.01, lr_hal=0.01, lr_lr=1e-5, syn_steps=steps, batch_syn=batch, expert_epochs=1, max_start_epoch=1)
rng = np.random.default_rng(55)
chunks = [torch.as_tensor(rng.permutation(C)[:batch]) for _ in range(steps)]
tr.draws = [(rng.integers(0, 2, batch), rng.integers(0, 2, batch)) for _ in range(steps)]
t0 = time.time()
grand_hip = float(tr.step(0, [start, target], start_epoch=0, index_chunks=chunks, update=False))
torch.cuda.synchronize()
t1 = time.time()
g_dyn, g_w, g_b, _, g_lr = tr.last_grads
torch.set_num_threads(min(32, os.cpu_count() or 1))
dyn = dynamic.reshape(C * dpc, T, 1, S, S).clone().requires_grad_(True)
w, b = hal_w.clone().requires_grad_(True), hal_b.clone().requires_grad_(True)
xs, labels = [], []
for s, these in enumerate(chunks):
    label, sidx, didx = tr.indices(these, s, 0)
    xs.append(R.hallucinator(static[sidx], dyn[didx], w, b))
    labels.append(label)
x_all = torch.cat(xs)
grand_ref, gx, glr_ref = R.mtt_step(start, target, x_all.detach(), torch.cat(labels), syn_lr,
                                    [torch.arange(s * batch, (s + 1) * batch) for s in range(steps)])
gd_ref, gw_ref, gb_ref = torch.autograd.grad(x_all, [dyn, w, b], grad_outputs=gx)
t2 = time.time()
# the same in fp64: over ten unrolled steps the fp32 oracle is itself a rounding-limited answer
dyn64 = dynamic.reshape(C * dpc, T, 1, S, S).double().clone().requires_grad_(True)
w64, b64 = hal_w.double().clone().requires_grad_(True), hal_b.double().clone().requires_grad_(True)
xs64 = []
for s, these in enumerate(chunks):
    label, sidx, didx = tr.indices(these, s, 0)
    xs64.append(R.hallucinator(static.double()[sidx], dyn64[didx], w64, b64))
x64 = torch.cat(xs64)
grand64, gx64, glr64 = R.mtt_step([q.double() for q in start], [q.double() for q in target], x64.detach(), torch.cat(labels), syn_lr,
                                  [torch.arange(s * batch, (s + 1) * batch) for s in range(steps)])
gd64, gw64, gb64 = torch.autograd.grad(x64, [dyn64, w64, b64], grad_outputs=gx64)
t3 = time.time()
rows64 = [i for i in range(C * dpc) if float(gd64[i].abs().sum()) > 0]
hip_vs_64 = sorted(rel(g_dyn[i], gd64[i]) for i in rows64)
f32_vs_64 = sorted(rel(gd_ref[i], gd64[i]) for i in rows64)
print("against the fp64 oracle (%.0f s): grand loss HIP %.1e / fp32 oracle %.1e; d/d syn_lr %.1e / %.1e; dynamic-memory gradient all rows %.2e / %.2e, "
      "per touched row median %.1e / %.1e, max %.1e / %.1e; hallucinator weight %.1e / %.1e, bias %.1e / %.1e" % (
          t3 - t2, abs(grand_hip / float(grand64) - 1), abs(float(grand_ref) / float(grand64) - 1), abs(float(g_lr) / float(glr64) - 1),
          abs(float(glr_ref) / float(glr64) - 1), rel(g_dyn, gd64), rel(gd_ref, gd64), hip_vs_64[len(hip_vs_64) // 2], f32_vs_64[len(f32_vs_64) // 2],
          hip_vs_64[-1], f32_vs_64[-1], rel(g_w.reshape(-1), gw64.reshape(-1)), rel(gw_ref.reshape(-1), gw64.reshape(-1)), rel(g_b, gb64), rel(gb_ref, gb64)))
vs64 = {"grand_loss_rel": [abs(grand_hip / float(grand64) - 1), abs(float(grand_ref) / float(grand64) - 1)],
        "d_syn_lr_rel": [abs(float(g_lr) / float(glr64) - 1), abs(float(glr_ref) / float(glr64) - 1)],
        "g_dynamic_rel_l2": [rel(g_dyn, gd64), rel(gd_ref, gd64)], "g_dynamic_per_row_median": [hip_vs_64[len(hip_vs_64) // 2], f32_vs_64[len(f32_vs_64) // 2]],
        "g_dynamic_per_row_max": [hip_vs_64[-1], f32_vs_64[-1]], "g_hal_w_rel_l2": [rel(g_w.reshape(-1), gw64.reshape(-1)), rel(gw_ref.reshape(-1), gw64.reshape(-1))],
        "g_hal_b_rel_l2": [rel(g_b, gb64), rel(gb_ref, gb64)], "order": "[HIP vs fp64 oracle, fp32 oracle vs fp64 oracle]"}
rows = [i for i in range(C * dpc) if float(gd_ref[i].abs().sum()) > 0]
per_row = sorted(rel(g_dyn[i], gd_ref[i]) for i in rows)
untouched = [i for i in range(C * dpc) if i not in set(rows)]
out = {"syn_steps": steps, "grand_loss_hip": grand_hip, "grand_loss_oracle_fp32": float(grand_ref), "grand_loss_rel": abs(grand_hip / float(grand_ref) - 1),
       "d_syn_lr_rel": abs(float(g_lr) / float(glr_ref) - 1), "g_dynamic_rel_l2": rel(g_dyn, gd_ref), "g_dynamic_per_touched_row": per_row,
       "g_hal_w_rel_l2": rel(g_w.reshape(-1), gw_ref.reshape(-1)), "g_hal_b_rel_l2": rel(g_b, gb_ref),
       "against_fp64_oracle": vs64, "untouched_rows_exactly_zero": all(float(g_dyn[i].abs().sum()) == 0.0 for i in untouched), "hip_seconds": t1 - t0, "oracle_seconds": t2 - t1,
       "command": "python tools/parity_mtt10.py %d" % steps}
print("config 5, syn_steps %d: grand loss HIP %.6f oracle (fp32) %.6f (rel %.1e); d/d syn_lr rel %.1e; dynamic-memory gradient rel-L2 all %.2e "
      "(per touched row: median %.1e, p90 %.1e, max %.1e, %d rows); hallucinator weight / bias %.1e / %.1e; untouched rows zero: %s   (HIP %.1f s, oracle %.0f s)" % (
          steps, grand_hip, float(grand_ref), out["grand_loss_rel"], out["d_syn_lr_rel"], out["g_dynamic_rel_l2"], per_row[len(per_row) // 2],
          per_row[int(0.9 * len(per_row))], per_row[-1], len(rows), out["g_hal_w_rel_l2"], out["g_hal_b_rel_l2"], out["untouched_rows_exactly_zero"],
          t1 - t0, t2 - t1))
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r05_parity_mtt10.json")
os.makedirs(os.path.dirname(path), exist_ok=True)
json.dump(out, open(path, "w"), indent=1)
