"""BASELINE configurations 4 and 5 at THEIR OWN geometry, trainer against oracle (the fixtures G9 / G10 / G13 pin the same code
paths at 64x64x8 with at most 8 clips):

  * config 4 (HMDB51 gradient matching): class terms of ``distill.GMTrainer`` with 64 real + 5 synthetic clips of 112x112x16,
    a 51-way head, the 'ours' metric and eight class lanes, against the reference-shaped double backward on the oracle --
    dCE/dparams on the real batch (detached), on the synthetic clips with ``create_graph=True``, ``match_loss``, backward to
    the pixels (utils.py:634-687; the calls of distill_baseline.py:243-262);
  * config 5 (Kinetics-400 "MTT+Ours"): one iteration of ``distill.S2DMTTTrainer`` with 400 classes, 256-clip
    hallucinator-composed student batches of 64x64x8 clips, two unrolled student steps, against ``oracle.ref_cpu.mtt_step``
    chained through the oracle's hallucinator (distill_s2d_ms.py:236-300).

The CPU side of each test takes 10 - 30 s on the GPU box's host."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / b.norm())


def test_gradient_matching_class_terms_at_config4_geometry(monkeypatch):
    from video_distillation_amd import distill, plan
    monkeypatch.setenv("VD_GM_LANES", "8")
    K, ipc, B, T, S, ncls = 51, 5, 64, 16, 112, 2
    geo = plan.NetGeometry(T, S, S)
    g = torch.Generator().manual_seed(404)
    base = torch.randn(ncls, T, 3, S, S, generator=g)
    clips = (base[:, None] + 0.7 * torch.randn(ncls, B + ipc, T, 3, S, S, generator=g)).reshape(-1, T, 3, S, S)
    counts = [B + ipc] * ncls + [0] * (K - ncls)
    offsets = [c * (B + ipc) for c in range(ncls)] + [0] * (K - ncls)
    params = R.init_params(4040, 3, K)
    lr_img = 0.1
    # rank 0 of 26 owns classes 0 and 1 of the 51 (class_range); no collective is issued by a one-outer-loop step
    assert distill.class_range(K, 0, 26) == (0, 2)
    tr = distill.GMTrainer(distill.HipGMOps("cuda:0", "ours"), distill.RealPool(clips.cuda(), counts, offsets), geo, K, ipc, batch_real=B,
                           lr_img=lr_img, rank=0, world=26, outer_loop=1, inner_loop=1, dropout_p=0.0, net_init=lambda it: params)
    syn0 = tr.image_syn.detach().clone()
    assert tuple(syn0.shape) == (ncls * ipc, T, 3, S, S)
    idx = distill.sample_real_indices(0, counts, offsets, B, [0, 1]).reshape(ncls, B)
    loss_hip = float(tr.step(0))
    g_hip = ((syn0 - tr.image_syn) / lr_img).cpu()            # first step of SGD(momentum): buf = g
    # ---- the oracle: the reference's call sequence, class by class ----
    p = [q.clone().requires_grad_(True) for q in params]
    loss_ref, g_ref = 0.0, []
    for k in range(ncls):
        xr = clips[torch.as_tensor(idx[k])]
        gw_real = [t.detach() for t in torch.autograd.grad(F.cross_entropy(R.convnet3d_logits(xr, p), torch.full((B,), k)), p)]
        xs = syn0[k * ipc:(k + 1) * ipc].cpu().clone().requires_grad_(True)
        gw_syn = torch.autograd.grad(F.cross_entropy(R.convnet3d_logits(xs, p), torch.full((ipc,), k)), p, create_graph=True)
        loss = R.match_loss(gw_syn, gw_real, "ours")
        (gx,) = torch.autograd.grad(loss, xs)
        loss_ref += float(loss)
        g_ref.append(gx)
    g_ref = torch.cat(g_ref)
    per_clip = [_rel(g_hip[i], g_ref[i]) for i in range(ncls * ipc)]
    print("config-4 geometry: matching loss HIP %.6f oracle %.6f (rel %.1e); pixel gradient rel-L2 per synthetic clip %s, all %.2e" % (
        loss_hip, loss_ref, abs(loss_hip / loss_ref - 1), ["%.1e" % v for v in per_clip], _rel(g_hip, g_ref)))
    assert abs(loss_hip / loss_ref - 1) < 1e-3                                # north_star's bar on the matching loss
    # pixel gradient: clips without a differing pooling decision agree to ~1e-3; one arg-max near-tie resolved the other way moves
    # a clip's gradient by up to a few 1e-2 -- most clips must be tight, every clip within the flip bound
    assert sorted(per_clip)[len(per_clip) // 2] < 3e-3 and max(per_clip) < 5e-2


def test_mtt_ours_iteration_at_config5_geometry():
    from video_distillation_amd import distill, plan
    C, vpc, spc, dpc, T, S, batch, steps, syn_lr = 400, 1, 2, 2, 8, 64, 256, 2, 0.01
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
                               lr_dynamic=0.01, lr_hal=0.01, lr_lr=1e-5, syn_steps=steps, batch_syn=batch, expert_epochs=1,
                               max_start_epoch=1)
    rng = np.random.default_rng(55)
    chunks = [torch.as_tensor(rng.permutation(C)[:batch]) for _ in range(steps)]
    tr.draws = [(rng.integers(0, 2, batch), rng.integers(0, 2, batch)) for _ in range(steps)]
    grand_hip = float(tr.step(0, [start, target], start_epoch=0, index_chunks=chunks, update=False))
    g_dyn, g_w, g_b, _, g_lr = tr.last_grads
    # ---- the oracle: compose every step's batch with the oracle hallucinator, unroll, back-propagate the grand loss ----
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
    rows = [i for i in range(C * dpc) if float(gd_ref[i].abs().sum()) > 0]
    per_row = sorted(_rel(g_dyn[i], gd_ref[i]) for i in rows)
    print("config-5 geometry: grand loss HIP %.6f oracle %.6f (rel %.1e); d/d syn_lr rel %.1e; dynamic-memory gradient rel-L2 all %.2e "
          "(per touched row: median %.1e, max %.1e, %d rows); hallucinator weight / bias %.1e / %.1e" % (
              grand_hip, float(grand_ref), abs(grand_hip / float(grand_ref) - 1), abs(float(g_lr) / float(glr_ref) - 1), _rel(g_dyn, gd_ref),
              per_row[len(per_row) // 2], per_row[-1], len(rows), _rel(g_w.reshape(-1), gw_ref.reshape(-1)), _rel(g_b, gb_ref)))
    assert abs(grand_hip / float(grand_ref) - 1) < 1e-4
    assert abs(float(g_lr) / float(glr_ref) - 1) < 5e-3
    # rows of the dynamic memory no student batch drew: exactly zero on both sides
    untouched = [i for i in range(C * dpc) if i not in set(rows)]
    assert all(float(g_dyn[i].abs().sum()) == 0.0 for i in untouched[:50])
    assert per_row[len(per_row) // 2] < 3e-3 and per_row[-1] < 8e-2 and _rel(g_dyn, gd_ref) < 1e-2
    assert _rel(g_w.reshape(-1), gw_ref.reshape(-1)) < 5e-3 and _rel(g_b, gb_ref) < 5e-3
