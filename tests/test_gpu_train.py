"""GPU parity of the HIP training step (train.TrainEngine / ConvNet3D.hip_train_step) -- the
evaluate_synset half of the loop (utils.py:765-792, 848-886) -- against the oracle in fp64."""
import os
import types

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu


def _oracle_loss_grads(x, labels, params, mask):
    p64 = [p.double().requires_grad_(True) for p in params]
    m = None if mask is None else mask.double()[:, :, :, None, None]
    logits = R.convnet3d_logits(x.double(), p64, drop_mask=m)
    loss = F.cross_entropy(logits, labels)
    grads = torch.autograd.grad(loss, p64)
    return float(loss), logits.detach(), [g.detach() for g in grads]


def _rel(a, b):
    return float((a.cpu().double() - b).norm() / b.norm())


@pytest.mark.parametrize("geom,B,K,use_mask,prec,prec_bwd,tight", [
    ((8, 64, 64), 5, 7, False, "f16x3", "f16x3", 1e-4),
    ((8, 64, 64), 11, 4, True, "f16x3", "f16x3", 1e-4),
    ((8, 64, 64), 9, 5, True, "f16x3", "f16", 3e-3),
    ((16, 112, 112), 3, 6, True, "f16x3", "f16x3", 1e-4),
    ((8, 64, 64), 6, 3, False, "bf16x3", "bf16x3", 1e-3),
])
def test_loss_and_grads_match_fp64_autograd(geom, B, K, use_mask, prec, prec_bwd, tight):
    """Parameter gradients of one batch vs fp64 autograd of the oracle.  A max-pool window whose two
    largest entries tie to within rounding routes its gradient differently in any two arithmetics
    (the fp32 reference itself differs from fp64 that way, test_gpu_embed._grad_fp64), and such a flip
    is confined to the clip it occurs in.  So: (i) the batch gradient must equal the sum of the HIP
    single-clip gradients (same arg-max decisions: exact up to fp32 summation order), and (ii) clip by
    clip the HIP gradient must match fp64 tightly for all but a few clips, loosely for those."""
    from video_distillation_amd import plan, train
    T, H, W = geom
    g = torch.Generator().manual_seed(B * 100 + K)
    x = R.standardise_batch(torch.randn(B, T, 3, H, W, generator=g))
    labels = torch.randint(0, K, (B,), generator=g)
    params = R.init_params(900 + K, 3, K)
    pool = (2, 2, 2) if H > 64 else (2, 1, 1)
    te = train.TrainEngine(plan.NetGeometry(T, H, W), K, pool, "cuda:0", prec=prec, prec_bwd=prec_bwd)
    mask = None
    if use_mask:
        mask = (torch.rand(B, te.C, te.Tp, generator=g) < 0.5).float() * 2.0
    pc = [p.cuda() for p in params]
    loss_ref, logits_ref, grads_ref = _oracle_loss_grads(x, labels, params, mask)
    loss, logits, grads = te.loss_and_grads(x.cuda(), labels.cuda(), pc, None if mask is None else mask.cuda())
    grads = [gr.clone() for gr in grads]
    torch.cuda.synchronize()
    assert abs(float(loss) - loss_ref) / abs(loss_ref) < 1e-4
    np.testing.assert_allclose(logits.cpu().double().numpy(), logits_ref.numpy(), rtol=1e-3, atol=1e-4)
    errs = [_rel(got, ref) for got, ref in zip(grads, grads_ref)]
    print(geom, B, K, prec, prec_bwd, "batch grad rel-l2 vs fp64:", " ".join("%.1e" % e for e in errs))
    assert max(errs) < 3e-2, errs
    # (i) + (ii): clip by clip
    acc = [torch.zeros_like(gr) for gr in grads]
    n_tight = 0
    for b in range(B):
        mb = None if mask is None else mask[b:b + 1]
        _, _, g1 = te.loss_and_grads(x[b:b + 1].cuda(), labels[b:b + 1].cuda(), pc, None if mb is None else mb.cuda())
        _, _, r1 = _oracle_loss_grads(x[b:b + 1], labels[b:b + 1], params, mb)
        e1 = max(_rel(a_, r_) for a_, r_ in zip(g1, r1))
        n_tight += e1 < tight
        assert e1 < 5e-2, (b, e1)
        for a_, g_ in zip(acc, g1):
            a_ += g_ / B
    print("   clips matching fp64 within %.0e: %d of %d" % (tight, n_tight, B))
    assert n_tight >= B - max(2, B // 4)
    lin = [float((a_ - g_).norm() / g_.norm()) for a_, g_ in zip(acc, grads)]
    print("   batch vs sum of single-clip HIP gradients:", " ".join("%.1e" % e for e in lin))
    assert max(lin) < (2e-3 if prec_bwd == "f16" else 2e-5), lin


def test_sgd_weight_decay_kernel_matches_torch():
    from video_distillation_amd import plan, train
    te = train.TrainEngine(plan.NetGeometry(8, 64, 64), 3, (2, 1, 1), "cuda:0")
    g = torch.Generator().manual_seed(5)
    p = torch.randn(1000, generator=g)
    ref = torch.nn.Parameter(p.clone())
    opt = torch.optim.SGD([ref], lr=0.01, momentum=0.9, weight_decay=5e-4)
    pd, bufs = p.cuda(), [None]
    for step in range(3):
        gr = torch.randn(1000, generator=g)
        ref.grad = gr.clone()
        opt.step()
        bufs = te.sgd_step([pd], [gr.cuda()], bufs, 0.01, 0.9, 5e-4)
    np.testing.assert_allclose(pd.cpu().numpy(), ref.detach().numpy(), rtol=1e-6, atol=1e-7)


def test_standardize_kernel():
    from video_distillation_amd import train
    g = torch.Generator().manual_seed(6)
    x = torch.randn(3, 8, 3, 64, 64, generator=g) * 3 + 1.5
    got = train.standardize(x.cuda()).cpu()
    np.testing.assert_allclose(got.numpy(), R.standardise_batch(x).numpy(), rtol=1e-5, atol=1e-5)
    with pytest.raises(RuntimeError):
        train.standardize(x)


def test_g7_training_curve_on_hip_path(golden_dir):
    """evaluate_synset on the device must take the HIP train step for every batch and reproduce
    the reference's per-epoch training losses (fixture G7 generated from the reference)."""
    from video_distillation_amd import networks, utils
    z = np.load(os.path.join(golden_dir, "g7_evaluate.npz"))
    C, n_test = int(z["C"]), int(z["n_test"])
    g = torch.Generator().manual_seed(int(z["data_seed"]))
    images = torch.randn(C, 8, 3, 64, 64, generator=g)
    test_x = torch.randn(n_test, 8, 3, 64, 64, generator=g)
    torch.manual_seed(int(z["net_seed"]))
    net = networks.ConvNet3D(channel=3, num_classes=C, net_width=128, net_depth=3, net_act='relu', net_norm='none',
                             net_pooling='maxpooling', im_size=(64, 64), frames=8)
    net.dropout.p = 0.0
    args = types.SimpleNamespace(device="cuda", lr_net=float(z["lr_net"]), epoch_eval_train=int(z["epochs"]),
                                 batch_train=256, model="ConvNet3D", eval_mode="SS")
    testloader = torch.utils.data.DataLoader(utils.TensorDataset(test_x, torch.arange(n_test) % C), batch_size=4)
    calls, losses = [], []
    orig_step, orig_epoch = networks.ConvNet3D.hip_train_step, utils.epoch

    def spy_step(self, x, lab, opt):
        calls.append(int(x.shape[0]))
        return orig_step(self, x, lab, opt)

    def spy_epoch(mode, *a):
        out = orig_epoch(mode, *a)
        if mode == 'train':
            losses.append(out[0])
        return out
    networks.ConvNet3D.hip_train_step, utils.epoch = spy_step, spy_epoch
    try:
        net_out, acc_train, acc_test, acc_per = utils.evaluate_synset(0, net, images, torch.arange(C), testloader, args,
                                                                      mode='none')
    finally:
        networks.ConvNet3D.hip_train_step, utils.epoch = orig_step, orig_epoch
    assert calls == [C] * (int(z["epochs"]) + 1)
    print("train losses", losses, "golden", z["train_loss"])
    np.testing.assert_allclose(losses, z["train_loss"], rtol=1e-3)
    assert abs(acc_train - float(z["acc_train"])) < 1e-6
    assert abs(acc_test - float(z["acc_test"])) < 1e-6
    l1 = np.array([float(p.double().abs().sum()) for p in net_out.parameters()])
    np.testing.assert_allclose(l1, z["params_after_l1"], rtol=1e-4)
    np.testing.assert_allclose(net_out.logit.weight.detach().reshape(C, -1)[:, :16].cpu().numpy(), z["logit_w_after"],
                               rtol=1e-3, atol=1e-5)


# ---- gradient matching: second-order pass ------------------------------------------------------

def _oracle_vjp(x, labels, params, mask, v):
    """d <v, dCE/dparams> / dx by double backward of the oracle in fp64."""
    xr = x.double().clone().requires_grad_(True)
    p64 = [p.double().requires_grad_(True) for p in params]
    m = None if mask is None else mask.double()[:, :, :, None, None]
    loss = F.cross_entropy(R.convnet3d_logits(xr, p64, drop_mask=m), labels)
    gw = torch.autograd.grad(loss, p64, create_graph=True)
    s = sum((a * b.double()).sum() for a, b in zip(gw, v))
    return torch.autograd.grad(s, xr)[0], [g.detach() for g in gw]


@pytest.mark.parametrize("geom,B,K,use_mask,which", [
    ((8, 64, 64), 2, 4, False, "all"),
    ((8, 64, 64), 3, 5, True, "all"),
    ((8, 64, 64), 2, 3, False, "head"),
    ((8, 64, 64), 2, 3, False, "w2"),
    ((8, 64, 64), 2, 3, False, "w1"),
    ((8, 64, 64), 2, 3, False, "w0"),
    ((16, 112, 112), 2, 6, True, "all"),
])
def test_second_order_vjp_matches_fp64_double_backward(geom, B, K, use_mask, which):
    from video_distillation_amd import plan, train
    T, H, W = geom
    g = torch.Generator().manual_seed(B * 10 + K)
    x = R.standardise_batch(torch.randn(B, T, 3, H, W, generator=g))
    labels = torch.randint(0, K, (B,), generator=g)
    params = R.init_params(800 + K, 3, K)
    pool = (2, 2, 2) if H > 64 else (2, 1, 1)
    te = train.GradMatchEngine(plan.NetGeometry(T, H, W), K, pool, "cuda:0")
    mask = (torch.rand(B, te.C, te.Tp, generator=g) < 0.5).float() * 2.0 if use_mask else None
    v = [torch.randn(p.shape, generator=g) for p in params]
    keep = {"all": range(8), "head": (6, 7), "w2": (4, 5), "w1": (2, 3), "w0": (0, 1)}[which]
    v = [t if i in keep else torch.zeros_like(t) for i, t in enumerate(v)]
    pc = [p.cuda() for p in params]
    per = []
    for b in range(B):          # clip by clip first: arg-max flips are confined to single clips (see above)
        mb = None if mask is None else mask[b:b + 1]
        want, gw_ref = _oracle_vjp(x[b:b + 1], labels[b:b + 1], params, mb, v)
        _, _, gw, state = te.param_grads(x[b:b + 1].cuda(), labels[b:b + 1].cuda(), pc, None if mb is None else mb.cuda())
        dx = te.vjp(state, [t.cuda() for t in v], pc)
        torch.cuda.synchronize()
        per.append((_rel(dx, want), max(_rel(a_, r_) for a_, r_ in zip(gw, gw_ref))))
    print(geom, B, K, which, "per-clip rel-l2 (vjp, grads):", ["%.1e/%.1e" % e for e in per])
    # a clip whose FIRST-order gradients already differ from fp64 had an arg-max flip (bf16x3 features
    # are exact to ~1e-5 only, so ties flip more often than in f16x3, and more often in large clips):
    # there the second-order result may be off by the same order, everywhere else it must be tight
    for e_vjp, e_g in per:
        assert e_vjp < (2e-4 if e_g < 1e-4 else min(5e-2, 10 * e_g)), per
    if H <= 64:
        assert sum(e[0] >= 2e-4 for e in per) <= 1
    # whole batch == sum of single-clip passes is implied by linearity over clips only for the gradients;
    # the vjp of a batch couples clips through the 1/B of the mean CE only: check it against fp64 loosely
    want, _ = _oracle_vjp(x, labels, params, mask, v)
    _, _, _, state = te.param_grads(x.cuda(), labels.cuda(), pc, None if mask is None else mask.cuda())
    dx = te.vjp(state, [t.cuda() for t in v], pc)
    e = [_rel(dx[b], want[b]) for b in range(B)]
    print("   batch per-clip rel-l2:", ["%.1e" % t for t in e])
    for t, (_, e_g) in zip(e, per):
        assert t < (2e-4 if e_g < 1e-4 else min(5e-2, 10 * e_g)), (e, per)


def test_param_grads_autograd_surface_and_match_loss():
    """net.param_grads(create_graph=True) + match_loss + backward == the same graph in fp64 on the oracle."""
    from video_distillation_amd import networks, utils
    K = 4
    g = torch.Generator().manual_seed(77)
    real = R.standardise_batch(torch.randn(3, 8, 3, 64, 64, generator=g))
    syn = R.standardise_batch(torch.randn(2, 8, 3, 64, 64, generator=g))
    lab_r, lab_s = torch.full((3,), 2), torch.full((2,), 2)
    torch.manual_seed(31)
    net = networks.ConvNet3D(3, K, 128, 3, 'relu', 'none', 'maxpooling', frames=8, im_size=(64, 64)).cuda().train()
    net.dropout.p = 0.0
    params = [p.detach().cpu() for p in net.parameters()]
    args = types.SimpleNamespace(device="cuda", dis_metric="ours")
    _, gw_real = net.param_grads(real.cuda(), lab_r.cuda())
    xs = syn.cuda().requires_grad_(True)
    _, gw_syn = net.param_grads(xs, lab_s.cuda(), create_graph=True)
    loss = utils.match_loss(gw_syn, [t.detach() for t in gw_real], args)
    loss.backward()
    # oracle, fp64
    p64 = [p.double().requires_grad_(True) for p in params]
    gr = torch.autograd.grad(F.cross_entropy(R.convnet3d_logits(real.double(), p64), lab_r), p64)
    xr = syn.double().clone().requires_grad_(True)
    gs = torch.autograd.grad(F.cross_entropy(R.convnet3d_logits(xr, p64), lab_s), p64, create_graph=True)
    want = R.match_loss(list(gs), [t.detach() for t in gr], "ours")
    wgrad = torch.autograd.grad(want, xr)[0]
    per = [_rel(xs.grad[b], wgrad[b]) for b in range(2)]
    print("match_loss %.6f vs %.6f, d/dx per-clip rel-l2" % (float(loss), float(want)), ["%.1e" % e for e in per])
    assert abs(float(loss) - float(want)) / abs(float(want)) < 1e-3
    assert min(per) < 1e-3 and max(per) < 5e-2


def test_g9_gradient_matching_vs_reference_golden(golden_dir):
    """Fixture G9 (generated from the reference's get_network / match_loss composed the upstream-DC way):
    loss and d loss / d syn of one class term, all three metrics, through net.param_grads + utils.match_loss."""
    from video_distillation_amd import networks, utils
    z = np.load(os.path.join(golden_dir, "g9_grad_match.npz"))
    C, lab = int(z["C"]), int(z["label"])
    g = torch.Generator().manual_seed(int(z["data_seed"]))
    real = torch.randn(3, 8, 3, 64, 64, generator=g)
    syn = torch.randn(2, 8, 3, 64, 64, generator=g)
    torch.manual_seed(int(z["net_seed"]))
    net = networks.ConvNet3D(3, C, 128, 3, 'relu', 'none', 'maxpooling', frames=8, im_size=(64, 64)).cuda().train()
    net.dropout.p = 0.0
    lab_r, lab_s = torch.full((3,), lab).cuda(), torch.full((2,), lab).cuda()
    _, gw_real = net.param_grads(real.cuda(), lab_r)
    np.testing.assert_allclose([float(t.double().abs().sum()) for t in gw_real], z["gw_real_l1"], rtol=2e-3)
    for metric in ("ours", "mse", "cos"):
        args = types.SimpleNamespace(device="cuda", dis_metric=metric)
        xs = syn.cuda().requires_grad_(True)
        _, gw_syn = net.param_grads(xs, lab_s, create_graph=True)
        loss = utils.match_loss(gw_syn, [t.detach() for t in gw_real], args)
        (gx,) = torch.autograd.grad(loss, xs)
        rel = abs(float(loss) - float(z["loss_" + metric])) / abs(float(z["loss_" + metric]))
        got = gx[0] if metric == "ours" else gx[:, 3]
        gerr = _rel(got, torch.tensor(z["grad_" + metric]).double())
        print("G9 %s: loss rel %.1e, grad rel-l2 %.1e" % (metric, rel, gerr))
        assert rel < 1e-3                       # north_star bar on the loss
        assert gerr < 2e-2                      # arg-max flips allowed (fp32 reference vs bf16x3), see above


def test_gm_trainer_hip_matches_oracle_trainer():
    """distill.GMTrainer on the HIP ops vs the same trainer on the oracle ops (CPU, fp32): two outer
    iterations with a network update in between (hip_train_step), then a second distillation step
    (momentum on the pixels, fresh network)."""
    from tests.cpu_backend import OracleGMOps
    from video_distillation_amd import distill, plan
    C, ipc = 2, 1
    geo = plan.NetGeometry(8, 64, 64)
    g = torch.Generator().manual_seed(4242)
    clips = torch.randn(C * 3, 8, 3, 64, 64, generator=g)
    syn0 = torch.stack([clips[0], clips[3]]).clone()
    init = lambda it: R.init_params(500 + it, 3, C)       # noqa: E731  same weights on both devices

    def run(ops, dev):
        pool = distill.RealPool(clips.to(dev), [3] * C, [0, 3])
        tr = distill.GMTrainer(ops, pool, geo, C, ipc, batch_real=2, lr_img=1e-3, lr_net=0.01, image_syn=syn0.clone().to(dev),
                               outer_loop=2, inner_loop=1, dropout_p=0.0, net_init=init)
        l0 = float(tr.step(0))
        s0 = tr.image_syn.cpu().clone()
        l1 = float(tr.step(1))
        return [l0, l1], s0, tr.image_syn.cpu()
    want_l, want_s0, want_s = run(OracleGMOps("ours"), "cpu")
    got_l, got_s0, got_s = run(distill.HipGMOps("cuda:0", "ours"), "cuda:0")
    print("GM losses", got_l, "oracle", want_l)
    # first iteration (two pixel updates with a network update in between): same start -> north_star bar on
    # the loss; the pixel update may differ where an arg-max flipped.  The 'ours' metric sums ~5e5 row
    # cosines of near-zero-norm gradient rows and is chaotic in the pixels, so the second iteration (which
    # starts from slightly different pixels) is only checked loosely.
    assert abs(got_l[0] - want_l[0]) / want_l[0] < 1e-3
    assert abs(got_l[1] - want_l[1]) / want_l[1] < 3e-2
    per = [_rel(got_s0[k] - syn0[k], (want_s0[k] - syn0[k]).double()) for k in range(C)]
    print("pixel update of iteration 0, rel-l2 per class:", ["%.1e" % e for e in per],
          "update/pixel norm %.2e" % float((want_s0 - syn0).norm() / syn0.norm()))
    assert max(per) < 5e-2
    assert torch.isfinite(got_s).all() and _rel(got_s, want_s.double()) < 5e-2

    # ---- the same second distillation step TEACHER-FORCED (round 4): both trainers start it from the HIP trainer's pixels and
    # momentum, so what is compared is one step's arithmetic, not the chaotic growth of the first step's 3e-4: the north_star bar
    # on the loss, and per class either a clean pixel gradient (< 3e-3: bf16-pair adjoints on 'ours') or the flip bound
    def forced(ops, dev, state):
        pool = distill.RealPool(clips.to(dev), [3] * C, [0, 3])
        tr = distill.GMTrainer(ops, pool, geo, C, ipc, batch_real=2, lr_img=1e-3, lr_net=0.01, image_syn=state[0].clone().to(dev),
                               outer_loop=1, inner_loop=1, dropout_p=0.0, net_init=init)
        tr.buf.copy_(state[1].to(dev)); tr.steps_done = 2
        before = tr.image_syn.detach().cpu().clone()
        loss = float(tr.step(1))
        g = (tr.buf.cpu() - 0.5 * state[1].cpu())          # buf = momentum * buf + g  (momentum .5, GMTrainer's default)
        return loss, g, before
    hip_tr_state = None
    pool0 = distill.RealPool(clips.to("cuda:0"), [3] * C, [0, 3])
    tr0 = distill.GMTrainer(distill.HipGMOps("cuda:0", "ours"), pool0, geo, C, ipc, batch_real=2, lr_img=1e-3, lr_net=0.01,
                            image_syn=syn0.clone().to("cuda:0"), outer_loop=2, inner_loop=1, dropout_p=0.0, net_init=init)
    float(tr0.step(0))
    hip_tr_state = (tr0.image_syn.detach().cpu().clone(), tr0.buf.detach().cpu().clone())
    l_hip, g_hip, _ = forced(distill.HipGMOps("cuda:0", "ours"), "cuda:0", hip_tr_state)
    l_cpu, g_cpu, _ = forced(OracleGMOps("ours"), "cpu", hip_tr_state)
    per2 = [_rel(g_hip[k], g_cpu[k].double()) for k in range(C)]
    print("teacher-forced second step: loss HIP %.6f oracle %.6f (rel %.1e), pixel gradient rel-l2 per class %s" % (
        l_hip, l_cpu, abs(l_hip / l_cpu - 1), ["%.1e" % e for e in per2]))
    assert abs(l_hip / l_cpu - 1) < 1e-3
    assert min(per2) < 3e-3 and max(per2) < 5e-2


def test_gm_trainer_class_lanes_match_serial(monkeypatch):
    """GMTrainer's class lanes (class k on stream k % lanes with its own engine slot) against the one-stream path on the
    same inputs: the per-class terms are independent, so loss and pixel update agree to the atomics' summation-order
    noise.  Dropout off: the two paths draw their masks in a different order."""
    from video_distillation_amd import distill, plan
    C, ipc = 6, 2
    geo = plan.NetGeometry(8, 64, 64)
    g = torch.Generator().manual_seed(99)
    clips = torch.randn(C * 4, 8, 3, 64, 64, generator=g).to("cuda:0")
    syn0 = clips[[i for c in range(C) for i in (4 * c, 4 * c + 1)]].clone()
    init = lambda it: R.init_params(900 + it, 3, C)       # noqa: E731

    def run(lanes, group=1):
        monkeypatch.setenv("VD_GM_LANES", str(lanes))
        monkeypatch.setenv("VD_GM_GROUP", str(group))
        pool = distill.RealPool(clips, [4] * C, [4 * c for c in range(C)])
        tr = distill.GMTrainer(distill.HipGMOps("cuda:0", "ours"), pool, geo, C, ipc, batch_real=3, lr_img=1e-3, image_syn=syn0.clone(),
                               outer_loop=1, dropout_p=0.0, net_init=init)
        out = []
        for rep in range(3):                      # fresh trainer state each time: the same first step, three times
            tr.image_syn.copy_(syn0); tr.buf.zero_(); tr.steps_done = 0
            loss = float(tr.step(0))
            out.append((loss, (tr.image_syn - syn0).cpu().double()))
        return out
    serial, laned = run(1), run(3) + run(3, group=4)      # (group 4: classes 0-3 and 4-5 share one real-batch forward per lane)
    noise = max(_rel(u, serial[0][1]) for _, u in serial[1:])
    err = max(_rel(u, serial[0][1]) for _, u in laned)
    print("GM lanes: loss serial %s lanes %s; update rel-l2 lanes-vs-serial %.2e (serial run-to-run %.2e)"
          % (["%.6f" % l for l, _ in serial], ["%.6f" % l for l, _ in laned], err, noise))
    for l, _ in laned:
        assert abs(l - serial[0][0]) / serial[0][0] < 1e-4
    assert err < max(1e-3, 5 * noise)


@pytest.mark.parametrize("prec_bwd", ["f16", "f16x3"])
def test_grouped_real_batches_equal_separate_calls(prec_bwd):
    """TrainEngine.loss_and_grads_grouped (one forward over the real batches of several classes, one backward per class on its slice
    of the kept activations and arg-max bytes) against one loss_and_grads call per sub-batch: same losses, same eight gradients up
    to the summation order of the weight-gradient atomics (2e-7 run to run)."""
    from video_distillation_amd import plan, train
    T, H, W, K, groups, per = 8, 64, 64, 6, 3, 5
    g = torch.Generator().manual_seed(31)
    x = R.standardise_batch(torch.randn(groups * per, T, 3, H, W, generator=g)).cuda()
    labels = torch.arange(groups).repeat_interleave(per).cuda()
    params = [p.cuda() for p in R.init_params(55, 3, K)]
    te = train.TrainEngine(plan.NetGeometry(T, H, W), K, (2, 1, 1), "cuda:0", prec="f16x3", prec_bwd=prec_bwd)
    mask = ((torch.rand(groups * per, te.C, te.Tp, generator=g) < 0.5).float() * 2.0).cuda()
    losses, logits, gs = te.loss_and_grads_grouped(x, labels, params, groups, mask)
    assert len(gs) == groups and tuple(logits.shape) == (groups * per, K)
    for k in range(groups):
        sl = slice(k * per, (k + 1) * per)
        loss, lg, want = te.loss_and_grads(x[sl], labels[sl], params, mask[sl])
        want = [t.clone() for t in want]
        assert abs(float(losses[k]) - float(loss)) <= 1e-6 * abs(float(loss))
        assert torch.equal(lg, logits[sl])
        for a, b in zip(gs[k], want):
            assert float((a - b).norm()) <= 2e-6 * float(b.norm()) + 1e-12, (k, tuple(a.shape))
    with pytest.raises(ValueError):
        te.loss_and_grads_grouped(x, labels, params, 4, mask)


@pytest.mark.parametrize("B,K,use_mask", [(1, 4, False), (3, 5, True)])
def test_hessian_vector_product_matches_fp64(B, K, use_mask):
    """param_adjoint=True: d <v, dCE/dparams> / d params (H v, MTT's unrolled inner loop) vs fp64."""
    from video_distillation_amd import plan, train
    T, H, W = 8, 64, 64
    g = torch.Generator().manual_seed(B * 7 + K)
    x = R.standardise_batch(torch.randn(B, T, 3, H, W, generator=g))
    labels = torch.randint(0, K, (B,), generator=g)
    params = R.init_params(700 + K, 3, K)
    te = train.GradMatchEngine(plan.NetGeometry(T, H, W), K, (2, 1, 1), "cuda:0")
    mask = (torch.rand(B, te.C, te.Tp, generator=g) < 0.5).float() * 2.0 if use_mask else None
    v = [torch.randn(p.shape, generator=g) for p in params]
    xr = x.double().clone().requires_grad_(True)
    p64 = [p.double().requires_grad_(True) for p in params]
    m = None if mask is None else mask.double()[:, :, :, None, None]
    gw = torch.autograd.grad(F.cross_entropy(R.convnet3d_logits(xr, p64, drop_mask=m), labels), p64, create_graph=True)
    s = sum((a * b.double()).sum() for a, b in zip(gw, v))
    want = torch.autograd.grad(s, [xr] + p64)
    pc = [p.cuda() for p in params]
    _, _, gw_hip, state = te.param_grads(x.cuda(), labels.cuda(), pc, None if mask is None else mask.cuda())
    dx, hv = te.vjp(state, [t.cuda() for t in v], pc, param_adjoint=True)
    torch.cuda.synchronize()
    e_first = max(_rel(a_, r_.detach()) for a_, r_ in zip(gw_hip, gw))
    errs = [_rel(dx, want[0])] + [_rel(a_, r_) for a_, r_ in zip(hv, want[1:])]
    print("H v rel-l2 (dx, 8 params):", " ".join("%.1e" % e for e in errs), "| first-order %.1e" % e_first)
    tol = 2e-4 if e_first < 1e-4 else min(5e-2, 10 * e_first)      # arg-max flip in this batch: see above
    assert max(errs) < tol, errs


def test_g10_mtt_step_on_hip_vs_reference_golden(golden_dir):
    """distill.MTTTrainer with the HIP ops vs fixture G10 (one MTT iteration run with the REFERENCE's
    ReparamModule + autograd): grand loss, d/d syn_lr, d/d image_syn."""
    from video_distillation_amd import distill, plan
    z = np.load(os.path.join(golden_dir, "g10_mtt_step.npz"))
    C, n_syn = int(z["C"]), int(z["n_syn"])
    start = R.init_params(int(z["net_seed"]), 3, C)
    g = torch.Generator().manual_seed(int(z["data_seed"]))
    target = [p + 0.02 * p.abs().mean() * torch.randn(p.shape, generator=g) for p in start]
    image_syn = torch.randn(n_syn, 8, 3, 64, 64, generator=g)
    ops = distill.HipMTTOps(plan.NetGeometry(8, 64, 64), C, "cuda:0", dropout_p=0.0)
    tr = distill.MTTTrainer(ops, C, image_syn.clone().cuda(), torch.tensor(z["labels"]).cuda(), float(z["syn_lr"]), lr_img=100.0,
                            lr_lr=1e-5, syn_steps=int(z["syn_steps"]), batch_syn=int(z["batch_syn"]), expert_epochs=1,
                            max_start_epoch=1)
    grand = tr.step(0, [start, target], start_epoch=0, index_chunks=[torch.tensor(i) for i in z["indices"]])
    g_img, g_lr = tr.last_grads
    want = torch.tensor(z["grad_img"]).double()
    per = [_rel(g_img[b, ::2, :, ::2, ::2], want[b]) for b in range(n_syn)]
    print("G10 grand loss %.6f vs %.6f, d/dlr %.5e vs %.5e, d/dx per-clip rel-l2 %s"
          % (grand, float(z["grand_loss"]), g_lr, float(z["grad_lr"]), ["%.1e" % e for e in per]))
    assert abs(grand - float(z["grand_loss"])) / float(z["grand_loss"]) < 1e-3
    assert abs(g_lr - float(z["grad_lr"])) / abs(float(z["grad_lr"])) < 1e-2
    assert max(per) < 5e-2 and sorted(per)[len(per) // 2] < 2e-3       # arg-max flips confined to single clips
    got_l1 = [float(g_img[b].double().abs().sum()) for b in range(n_syn)]
    np.testing.assert_allclose(got_l1, z["grad_l1"], rtol=2e-2)
    # optimiser bookkeeping on the device
    np.testing.assert_allclose(tr.image_syn.cpu().numpy(), (image_syn - 100.0 * g_img.cpu()).numpy(), rtol=1e-5, atol=1e-6)


def test_g13_s2d_mtt_trainer_on_hip_vs_reference_golden():
    """distill.S2DMTTTrainer on the HIP ops vs fixture G13 (one "MTT+Ours" iteration of the reference,
    distill_s2d_ms.py:189-300): grand loss, gradients of dynamic / static memories, hallucinator and syn_lr, and the
    four optimiser updates."""
    from tests.test_distributed_cpu import _s2d_mtt_setup, check_s2d_mtt_against_g13
    from video_distillation_amd import distill, plan
    ops = distill.HipMTTOps(plan.NetGeometry(8, 64, 64), 3, "cuda:0", dropout_p=0.0)
    z, tr, traj, chunks = _s2d_mtt_setup(ops=ops, dev="cuda")
    grand = tr.step(0, traj, start_epoch=0, index_chunks=chunks)
    print("G13 grand %.6f vs %.6f, d/dlr %.5e vs %.5e" % (float(grand), float(z["grand_loss"]), float(tr.last_grads[4]), float(z["grad_lr"])))
    check_s2d_mtt_against_g13(z, tr, grand, tol=5e-3)


def test_atomic_accumulation_run_to_run_spread_is_bounded():
    """Weight gradients, the classifier head, the hallucinator's parameter / memory gradients and match_loss accumulate
    with fp32 atomics, so their summation order -- and the last bits -- change from run to run.  Bound the spread: five
    repetitions of the same call stay within 2e-6 rel-L2 of each other (fp32 rounding of sums of a few thousand terms),
    and the kernels without atomics (forward, input gradient) are bitwise reproducible."""
    import types
    from video_distillation_amd import networks, plan, train, utils
    C, B = 5, 6
    g = torch.Generator().manual_seed(606)
    x = torch.randn(B, 8, 3, 64, 64, generator=g).cuda()
    y = torch.tensor([0, 1, 2, 3, 4, 0]).cuda()
    params = [p.cuda() for p in R.init_params(61, 3, C)]
    te = train.TrainEngine(plan.NetGeometry(8, 64, 64), C, (2, 1, 1), "cuda:0", prec="f16x3", prec_bwd="f16x3")
    runs = []
    for _ in range(5):
        loss, logits, grads = te.loss_and_grads(x, y, params, None)
        runs.append((float(loss), logits.clone(), [t.clone() for t in grads]))
    assert all(r[0] == runs[0][0] for r in runs) and all(torch.equal(r[1], runs[0][1]) for r in runs)     # forward: no atomics
    spread = max(_rel(r[2][i], runs[0][2][i].cpu().double()) for r in runs[1:] for i in range(8))
    # hallucinator: g_dyn / g_stat / g_w / g_b all via atomics
    hal = utils.Conv3DNet(img_size=64).cuda()
    st = torch.randn(4, 3, 64, 64, generator=g).cuda().requires_grad_(True)
    dy = torch.randn(4, 8, 1, 64, 64, generator=g).cuda().requires_grad_(True)
    up = torch.randn(4, 8, 3, 64, 64, generator=g).cuda()
    hruns = []
    for _ in range(5):
        for t in (st, dy, hal.encoder.weight, hal.encoder.bias):
            t.grad = None
        out = hal(st, dy)
        out.backward(up)
        hruns.append((out.detach().clone(), st.grad.clone(), dy.grad.clone(), hal.encoder.weight.grad.clone()))
    assert all(torch.equal(h[0], hruns[0][0]) for h in hruns)
    hspread = max(_rel(h[i], hruns[0][i].cpu().double()) for h in hruns[1:] for i in (1, 2, 3))
    # match_loss: 5 shared accumulators
    gw = [torch.randn_like(p) for p in params]
    gs = [torch.randn_like(p) for p in params]
    vals = [float(utils.match_loss(gs, gw, types.SimpleNamespace(device="cuda", dis_metric="ours"))) for _ in range(5)]
    mspread = max(abs(v / vals[0] - 1) for v in vals)
    print("run-to-run spread: training gradients %.1e, hallucinator gradients %.1e, match_loss %.1e" % (spread, hspread, mspread))
    assert spread < 2e-6 and hspread < 2e-6 and mspread < 2e-6


def test_train_step_is_undisturbed_by_a_laned_gm_step(monkeypatch):
    """State that could leak between users of the cached engines: a GMTrainer step on three class lanes --
    slot engines, workspaces created on lane streams, the global precision table -- then ``hip_train_step`` on the default
    stream with the SAME geometry / class count / batch bucket (hence the same cached TrainEngine as lane 0): its loss, logits
    and all eight updated parameter tensors must equal those of the same step taken BEFORE the laned step to the fp32 atomics'
    run-to-run noise (2e-6; a stale workspace or an unjoined lane stream would show as 1e-3 or worse), and its recorded
    pooling decisions bitwise."""
    from video_distillation_amd import distill, networks, plan, train
    C, B = 3, 4
    g = torch.Generator().manual_seed(4242)
    x = torch.randn(B, 8, 3, 64, 64, generator=g).cuda()
    y = (torch.arange(B) % C).cuda()
    p0 = R.init_params(31, 3, C)
    captured = []
    orig_fw = train.TrainEngine._forward

    def spy(self, xx, params):
        out = orig_fw(self, xx, params)
        captured.append([a.clone() for a in out[2]])
        return out

    def one_step():
        net = networks.ConvNet3D(3, C, 128, 3, 'relu', 'none', 'maxpooling', 8, (64, 64)).cuda().train()
        with torch.no_grad():
            for p, q in zip(net.parameters(), p0):
                p.copy_(q)
        net.dropout.p = 0.0
        opt = torch.optim.SGD(net.parameters(), lr=0.05, momentum=0.9, weight_decay=5e-4)
        assert net.hip_trainable(x, opt, torch.nn.CrossEntropyLoss().cuda())
        train.TrainEngine._forward = spy
        try:
            logits, loss = net.hip_train_step(x, y, opt)
        finally:
            train.TrainEngine._forward = orig_fw
        torch.cuda.synchronize()
        return float(loss), logits.clone(), [p.detach().clone() for p in net.parameters()]
    before = one_step()
    # the laned gradient-matching step in between: same geometry and class count, three lanes
    monkeypatch.setenv("VD_GM_LANES", "3")
    clips = torch.randn(C * 4, 8, 3, 64, 64, generator=g).to("cuda:0")
    pool = distill.RealPool(clips, [4] * C, [4 * c for c in range(C)])
    tr = distill.GMTrainer(distill.HipGMOps("cuda:0", "ours"), pool, plan.NetGeometry(8, 64, 64), C, 1, batch_real=4, lr_img=1e-3,
                           outer_loop=1, dropout_p=0.0, net_init=lambda it: R.init_params(900 + it, 3, C))
    float(tr.step(0))                      # (no explicit synchronize: the next step must order itself behind the lanes)
    after = one_step()
    assert all(torch.equal(a, b) for a, b in zip(captured[0], captured[-1])), "pooling decisions changed"
    assert abs(after[0] / before[0] - 1) < 2e-6 and _rel(after[1], before[1].cpu().double()) < 2e-6
    upd = [_rel(a - q.cuda(), (b - q.cuda()).cpu().double()) for a, b, q in zip(after[2], before[2], p0)]
    print("train step before / after a laned GM step: loss %.7f / %.7f, per-tensor update difference %s" % (
        before[0], after[0], ["%.1e" % v for v in upd]))
    assert max(upd) < 2e-5


def test_expert_trajectories_match_the_oracle_loop(tmp_path):
    """checkpoint.train_expert_trajectories (the buffer.py:64-95 producer on the HIP train step) against the same loop
    written out on the oracle: fresh net, SGD(momentum, weight decay) over shuffled mini-batches, parameters recorded
    before training and after every epoch, lr x0.1 with a fresh optimiser after epoch E//2+1; then the
    replay_buffer_{n}.pt round trip."""
    import argparse
    from video_distillation_amd import checkpoint, networks, utils
    C, n, batch, epochs, lr, mom, l2 = 3, 8, 4, 3, 0.02, 0.5, 1e-2
    g = torch.Generator().manual_seed(515)
    x = torch.randn(n, 8, 3, 64, 64, generator=g)
    y = torch.arange(n) % C
    p0 = R.init_params(808, 3, C)

    def factory():
        net = networks.ConvNet3D(3, C, 128, 3, 'relu', 'none', 'maxpooling', 8, (64, 64))
        with torch.no_grad():
            for p, q in zip(net.parameters(), p0):
                p.copy_(q)
        net.dropout.p = 0.0
        return net
    shuffle = torch.Generator()               # the loader's own generator: the net constructor draws from the global one
    loader = torch.utils.data.DataLoader(utils.TensorDataset(x, y), batch_size=batch, shuffle=True, num_workers=0, generator=shuffle)
    args = argparse.Namespace(device="cuda:0", model="ConvNet3D")
    # every HIP train step records the state it started from, its gradients and its pooling decisions
    from video_distillation_amd import train
    steps_rec, orig_lg, orig_fw = [], train.TrainEngine.loss_and_grads, train.TrainEngine._forward

    def spy_forward(self, xx, params):
        feats, nb, am = orig_fw(self, xx, params)
        steps_rec[-1]["am"] = [a.clone() for a in am]
        return feats, nb, am

    def spy_loss_and_grads(self, xx, labels, params, mask=None, state=None):
        steps_rec.append({"x": xx.detach().clone(), "labels": labels.detach().clone(), "params": [q.detach().clone() for q in params]})
        out = orig_lg(self, xx, labels, params, mask, state)
        steps_rec[-1]["grads"] = [gq.detach().clone() for gq in out[2]]
        return out
    train.TrainEngine.loss_and_grads, train.TrainEngine._forward = spy_loss_and_grads, spy_forward
    try:
        torch.cuda.synchronize()
        shuffle.manual_seed(99)
        traj = checkpoint.train_expert_trajectories(factory, loader, args, num_experts=1, train_epochs=epochs, lr_teacher=lr,
                                                    mom=mom, l2=l2, decay=True)
    finally:
        train.TrainEngine.loss_and_grads, train.TrainEngine._forward = orig_lg, orig_fw
    assert len(traj) == 1 and len(traj[0]) == epochs + 1 and len(traj[0][0]) == 8
    assert all(t.device.type == "cpu" for t in traj[0][-1])
    # the oracle loop over the same shuffles
    shuffle.manual_seed(99)
    params = [p.detach().clone().requires_grad_(True) for p in p0]
    bufs, cur_lr, want = [None] * 8, lr, [[p.detach().clone() for p in p0]]
    for e in range(epochs):
        for xb, yb in loader:
            logits = R.convnet3d_logits(R.standardise_batch(xb), params, training=False)
            grads = torch.autograd.grad(torch.nn.functional.cross_entropy(logits, yb), params)
            with torch.no_grad():
                for i, (p, gr) in enumerate(zip(params, grads)):
                    gr = gr + l2 * p
                    bufs[i] = gr.clone() if bufs[i] is None else bufs[i] * mom + gr
                    p -= cur_lr * bufs[i]
        want.append([p.detach().clone() for p in params])
        if e == epochs // 2 + 1:
            cur_lr *= 0.1
            bufs = [None] * 8
    errs = [[_rel(a - s, (b - s).double()) if e else float((a - b).abs().max()) for a, b, s in zip(traj[0][e], want[e], p0)]
            for e in range(epochs + 1)]
    # ---- per step, against the fp64 oracle on the state the HIP step started from (no accumulation, no second attempt) ----
    # A step is CLEAN when all 8 parameter gradients are within 1e-3 rel-L2 of the fp64 oracle's.  Anything above that must be a
    # pooling near-tie: at 256 features per clip and 4 clips per batch ONE last-level window routed the other way moves every
    # gradient by 1e-3 .. 3e-2 -- then the step's recorded arg-max bytes must differ from the fp64 oracle's max_pool3d decisions,
    # and ONLY in windows whose two largest entries the oracle itself separates by < 2e-5 of the level's rms (the HIP forward's
    # own rounding is 4e-6).  An outlier without such a window is a hazard (stale workspace, stream race), and fails.
    from tests import argmax_tools
    log = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "expert_test.log")
    os.makedirs(os.path.dirname(log), exist_ok=True)
    assert len(steps_rec) == epochs * (n // batch)
    flips, report = 0, []
    for k, st in enumerate(steps_rec):
        p64 = [q.double().cpu().requires_grad_(True) for q in st["params"]]
        logits = R.convnet3d_logits(st["x"].double().cpu(), p64, training=False)
        want_g = torch.autograd.grad(torch.nn.functional.cross_entropy(logits, st["labels"].cpu()), p64)
        gerr = [_rel(a, b) for a, b in zip(st["grads"], want_g)]
        dec = argmax_tools.compare_decisions(st["x"], st["params"], st["am"])
        line = "step %d: max gradient error vs fp64 %.2e; arg-max mismatches / not near-tie / worst margin per level: %s" % (
            k, max(gerr), ["%d/%d/%.1e" % (d["mismatch"], d["not_near_tie"], d["worst_margin"]) for d in dec])
        report.append(line)
        print(line)
        assert all(d["not_near_tie"] == 0 for d in dec), line              # every differing decision is a genuine near-tie
        if max(gerr) >= 1e-3:
            flips += 1
            assert max(gerr) < 5e-2 and sum(d["mismatch"] for d in dec[1:]) > 0, "outlier without a flipped window: " + line
    for e, st in enumerate(errs):
        print("expert epoch %d: per-tensor error of the accumulated update" % e, ["%.1e" % v for v in st])
    with open(log, "a") as fp:
        fp.write("\n".join(report + ["epoch %d %s" % (e, " ".join("%.2e" % v for v in st)) for e, st in enumerate(errs)]) + "\n")
    assert max(errs[0]) < 1e-12
    bar = 1e-3 if flips == 0 else 5e-2          # typical 8e-5; after an identified flip the trajectories legitimately part
    assert all(max(st) < bar for st in errs[1:]), (flips, errs)
    path = checkpoint.save_expert_buffer(str(tmp_path), traj)
    assert path.endswith("replay_buffer_0.pt")
    back = checkpoint.load_expert_buffers(str(tmp_path))
    assert len(back) == 1 and all(torch.equal(a, b) for a, b in zip(back[0][2], traj[0][2]))


def test_buffer_driver_writes_reference_format_files(tmp_path):
    """buffer.run on HBM-resident clips: replay_buffer_{n}.pt files of save_interval experts each, first timestamp = the
    expert's initial parameters (8 tensors, parameters() order), and the files feed load_expert_buffers."""
    from video_distillation_amd import buffer, checkpoint
    C = 3
    g = torch.Generator().manual_seed(61)
    clips = torch.randn(7, 8, 3, 64, 64, generator=g).to("cuda:0")
    labels = torch.arange(7) % C
    args = buffer.build_parser().parse_args(["--num_experts", "3", "--train_epochs", "2", "--batch_train", "4", "--lr_teacher", "0.01",
                                             "--save_interval", "2", "--buffer_path", str(tmp_path)])
    files = buffer.run(args, train=(clips, labels), num_classes=C, log=lambda *_: None)
    assert [os.path.basename(f) for f in files] == ["replay_buffer_0.pt"]           # the third expert stays pending, as upstream
    back = checkpoint.load_expert_buffers(str(tmp_path))
    assert len(back) == 2 and len(back[0]) == 3 and len(back[0][0]) == 8
    from video_distillation_amd.distill import FULL_SHAPES
    assert [tuple(t.shape) for t in back[1][2]] == [tuple(s) for s in FULL_SHAPES(C)]
    moved = [float((a - b).abs().max()) for a, b in zip(back[0][0], back[0][2])]
    assert min(moved) > 0 and all(np.isfinite(moved))
    assert not torch.equal(back[0][0][0], back[1][0][0])                             # every expert starts from its own init
