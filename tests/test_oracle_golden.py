"""The CPU oracle (oracle/ref_cpu.py) against the fixtures generated from the reference
(tools/gen_golden.py).  Tolerances: fp32 op-level 1e-5 abs / 1e-4 rel."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ref_cpu as R

torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def close(a, b, rtol=1e-4, atol=1e-5):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, np.asarray(b), rtol=rtol, atol=atol)


def randn(seed, *shapes):
    g = torch.Generator().manual_seed(int(seed))
    return [torch.randn(*s, generator=g) for s in shapes]


def test_g1_init_and_layers(golden_dir):
    z = load(golden_dir, "g1_layers.npz")
    params = R.init_params(int(z["seed"]), 3, 50)
    chk = np.array([float(p.double().sum()) for p in params] + [float(p.double().abs().sum()) for p in params])
    np.testing.assert_allclose(chk, z["checksum"], rtol=1e-12, atol=1e-9)
    close(params[0][:4], z["w0_probe"], rtol=0, atol=0)
    close(params[5], z["b2"], rtol=0, atol=0)
    (x,) = randn(z["x_seed"], (2, 8, 3, 64, 64))
    close(x[:, :, :, ::8, ::8], z["x"], rtol=0, atol=0)
    col = []
    R.feature_layers(x.permute(0, 2, 1, 3, 4), params, collect=col)
    close(col[0][:, ::8, :, ::4, ::4], z["conv0"])
    close(col[2][:, ::8, :, ::2, ::2], z["pool0"])
    close(col[3][:, ::16], z["conv1"])
    close(col[5][:, ::8], z["pool1"])
    close(col[6][:, ::8], z["conv2"])
    close(col[8], z["pool2"])
    close(R.convnet3d_embed(x, params), z["embed"])
    close(R.convnet3d_logits(x, params), z["logits"])


def test_g1_full_resolution(golden_dir):
    z = load(golden_dir, "g1_layers.npz")
    params = R.init_params(int(z["seed"]), 3, 50)
    (x,) = randn(z["x112_seed"], (1, 16, 3, 112, 112))
    emb = R.convnet3d_embed(x, params)
    assert emb.shape == (1, 2048)
    close(emb, z["embed112"])
    close(R.convnet3d_logits(x, params), z["logits112"])


def test_g2_dm_class_term(golden_dir):
    z = load(golden_dir, "g2_dm_class.npz")
    params = R.init_params(int(z["seed"]))
    real, syn = randn(z["data_seed"], (4, 8, 3, 64, 64), (1, 8, 3, 64, 64))
    loss, grad = R.dm_loss_and_grad(params, [real], syn, ipc=1)
    close(loss, z["loss"], rtol=1e-5)
    close(grad, z["grad_syn"], rtol=1e-4, atol=1e-7)
    close(R.convnet3d_embed(real, params).mean(0), z["feat_real_mean"])


def _g3_run(z, class_slices):
    (syn,) = randn(z["syn_seed"], (3, 8, 3, 64, 64))
    buf, losses, grads, syns = None, [], [], []
    for it in range(2):
        params = R.init_params(int(z["net_seeds"][it]))
        reals = randn(z["real_seeds"][it], *[(4, 8, 3, 64, 64)] * 3)
        loss_total, grad_total = 0.0, torch.zeros_like(syn)
        for cls in class_slices:  # "ranks": each computes its own classes, sums are exchanged
            sub = torch.cat([syn[c:c + 1] for c in cls])
            loss, grad = R.dm_loss_and_grad(params, [reals[c] for c in cls], sub, ipc=1)
            loss_total += float(loss)
            for j, c in enumerate(cls):
                grad_total[c] = grad[j]
        syn, buf = R.sgd_momentum_step(syn, grad_total, buf, float(z["lr"]), float(z["momentum"]))
        losses.append(loss_total); grads.append(grad_total); syns.append(syn)
    return losses, grads, syns


def test_g3_two_dm_steps(golden_dir):
    z = load(golden_dir, "g3_dm_steps.npz")
    losses, grads, syns = _g3_run(z, [[0, 1, 2]])
    np.testing.assert_allclose(losses, z["losses"], rtol=1e-5)
    close(grads[0][:, ::2, :, ::4, ::4], z["grad0"], atol=1e-7)
    close(grads[1][:, ::2, :, ::4, ::4], z["grad1"], atol=1e-7)
    close(syns[0][:, ::2, :, ::4, ::4], z["syn1"])
    close(syns[1][:, ::2, :, ::4, ::4], z["syn2"])
    assert abs(float(syns[1].double().sum()) - float(z["syn2_sum"])) < 1e-2
    assert abs(float(syns[1].double().abs().sum()) / float(z["syn2_abs"]) - 1) < 1e-6


def test_g12_late_regime_first_steps(golden_dir):
    """Fixture G12 (24 reference DM iterations in the small-difference regime): the oracle reproduces the first two
    iterations (loss, pixel gradient, updated clips); the GPU test runs all 24."""
    z = load(golden_dir, "g12_dm_late.npz")
    C, B, NP = int(z["C"]), int(z["batch_real"]), int(z["pool_per_class"])
    g = torch.Generator().manual_seed(int(z["data_seed"]))
    base = torch.randn(C, 8, 3, 64, 64, generator=g)
    pool = torch.stack([base[c] + 0.1 * torch.randn(NP, 8, 3, 64, 64, generator=g) for c in range(C)])
    syn, buf = pool[:, 0].clone(), None
    for it in range(2):
        params = R.init_params(int(z["net_seed0"]) + it)
        reals = [pool[c, torch.as_tensor(z["picks"][it][c])] for c in range(C)]
        loss, grad = R.dm_loss_and_grad(params, reals, syn, ipc=1)
        np.testing.assert_allclose(float(loss), z["losses"][it], rtol=1e-5)
        close(grad[:, ::2, :, ::4, ::4], z["grads"][it], rtol=1e-4, atol=1e-9)
        syn, buf = R.sgd_momentum_step(syn, grad, buf, float(z["lr"]), float(z["momentum"]))
        close(syn[:, ::2, :, ::4, ::4], z["syns"][it])
    assert 0.03 < float(z["rel_diff"].mean()) < 0.05      # the regime the fixture is about


@pytest.mark.parametrize("shards", [[[0, 1], [2]], [[0], [1], [2]]])
def test_g8_class_sharding_identity(golden_dir, shards):
    z = load(golden_dir, "g3_dm_steps.npz")
    losses, grads, syns = _g3_run(z, shards)
    np.testing.assert_allclose(losses, z["losses"], rtol=1e-5)
    close(syns[1][:, ::2, :, ::4, ::4], z["syn2"])


def test_g4_hallucinator(golden_dir):
    z = load(golden_dir, "g4_hallucinator.npz")
    static, dynamic, up = randn(z["data_seed"], (3, 3, 64, 64), (3, 8, 1, 64, 64), (3, 8, 3, 64, 64))
    w = torch.tensor(z["weight"]).requires_grad_(True)
    b = torch.tensor(z["bias"]).requires_grad_(True)
    static.requires_grad_(True); dynamic.requires_grad_(True)
    out = R.hallucinator(static, dynamic, w, b)
    assert out.shape == (3, 8, 3, 64, 64)
    close(out[:, :, :, ::2, ::2], z["out"])
    (out * up).sum().backward()
    close(dynamic.grad[:, :, :, ::2, ::2], z["g_dynamic"])
    close(static.grad[:, :, ::2, ::2], z["g_static"], rtol=1e-4, atol=1e-4)
    close(w.grad, z["g_weight"], rtol=1e-4, atol=1e-2)
    close(b.grad, z["g_bias"], rtol=1e-4, atol=1e-2)


def test_g5_s2d_step(golden_dir):
    z = load(golden_dir, "g5_s2d_step.npz")
    C, vpc, spc, dpc = 3, 1, 2, 2
    static_syn, dynamic_syn = randn(z["data_seed"], (C * spc, 3, 64, 64), (C, dpc, 8, 1, 64, 64))
    label, didx, sidx = R.s2d_indices(C, vpc, spc, torch.tensor(z["draws_dyn"]), torch.tensor(z["draws_sta"]))
    np.testing.assert_array_equal(didx.numpy(), z["dynamic_idx"])
    np.testing.assert_array_equal(sidx.numpy(), z["static_idx"])
    w = torch.tensor(z["hal_w"]).requires_grad_(True)
    b = torch.tensor(z["hal_b"]).requires_grad_(True)
    dynamic_syn.requires_grad_(True)
    params = R.init_params(int(z["net_seed"]))
    image_syn = R.hallucinator(static_syn[sidx], dynamic_syn[label, didx], w, b)
    reals = randn(z["real_seed"], *[(4, 8, 3, 64, 64)] * C)
    loss = torch.zeros(())
    for c in range(C):
        loss = loss + R.dm_class_term(R.convnet3d_embed(reals[c], params),
                                      R.convnet3d_embed(image_syn[c * vpc:(c + 1) * vpc], params))
    loss.backward()
    assert abs(float(loss) / float(z["loss"]) - 1) < 1e-5
    close(dynamic_syn.grad[:, :, :, :, ::4, ::4], z["g_dynamic"], atol=1e-8)
    rowabs = dynamic_syn.grad.abs().sum(dim=(2, 3, 4, 5))
    close(rowabs, z["g_dynamic_rowabs"], rtol=1e-4, atol=1e-7)
    assert (rowabs == 0).sum() == C * (dpc - 1)  # unselected dynamic rows get exactly zero grad
    close(w.grad, z["g_hal_w"], rtol=1e-3, atol=1e-6)
    close(b.grad, z["g_hal_b"], rtol=1e-3, atol=1e-6)
    d2, _ = R.sgd_momentum_step(dynamic_syn.detach(), dynamic_syn.grad, None, 10.0, 0.95)
    close(d2[:, :, :, :, ::4, ::4], z["dynamic_after"])
    w2, _ = R.sgd_momentum_step(w.detach(), w.grad, None, 0.01, 0.95)
    close(w2, z["hal_w_after"], rtol=1e-5, atol=1e-7)


def test_g6_match_loss(golden_dir):
    z = load(golden_dir, "g6_match_loss.npz")
    n = int(z["n"])
    gr = [torch.tensor(z["r%d" % i]) for i in range(n)]
    per = [float(R.distance_wb(a, torch.tensor(z["s%d" % i]))) for i, a in enumerate(gr)]
    np.testing.assert_allclose(per, z["ours_per_layer"], rtol=1e-5, atol=1e-6)
    assert per[1] == 0.0 and per[3] == 0.0 and per[5] == 0.0  # 1-D members contribute nothing
    # 5-D fall-through: cosine over the last axis only, one term per (o,i,kt,kh) row
    a, b = gr[0], torch.tensor(z["s0"])
    manual = sum(1 - float((a[idx] * b[idx]).sum()) / (float(a[idx].norm()) * float(b[idx].norm()) + 1e-6)
                 for idx in np.ndindex(*a.shape[:-1]))
    assert abs(manual - per[0]) < 1e-3
    for metric in ("ours", "mse", "cos"):
        gs = [torch.tensor(z["s%d" % i]).requires_grad_(True) for i in range(n)]
        val = R.match_loss(gs, gr, metric)
        close(val, z["val_" + metric], rtol=1e-5)
        val.backward()
        for i, s in enumerate(gs):
            got = s.grad if s.grad is not None else torch.zeros_like(s)
            close(got, z["grad_%s_%d" % (metric, i)], rtol=1e-4, atol=1e-6)
    with pytest.raises(ValueError):
        R.match_loss(gr, gr, "nope")


def test_g6_match_loss_on_network_grads(golden_dir):
    z = load(golden_dir, "g6_match_loss.npz")
    params = [p.requires_grad_(True) for p in R.init_params(int(z["net_seed"]), 3, 5)]
    xr, xs = randn(z["net_data_seed"], (2, 8, 3, 64, 64), (2, 8, 3, 64, 64))
    y = torch.tensor([1, 3])
    outs = []
    for x in (xr, xs):
        torch.manual_seed(int(z["net_drop_seed"]))
        logits = R.convnet3d_logits(x, params, training=True)
        outs.append(torch.autograd.grad(torch.nn.functional.cross_entropy(logits, y), params))
    l1 = np.array([float(t.double().abs().sum()) for t in outs[0]])
    np.testing.assert_allclose(l1, z["net_gw_real_l1"], rtol=1e-4)
    for metric in ("ours", "mse", "cos"):
        close(R.match_loss(outs[1], outs[0], metric), z["net_" + metric], rtol=2e-4)


def test_g7_evaluate_training_curve(golden_dir):
    z = load(golden_dir, "g7_evaluate.npz")
    C, epochs = int(z["C"]), int(z["epochs"])
    images, _ = randn(z["data_seed"], (C, 8, 3, 64, 64), (int(z["n_test"]), 8, 3, 64, 64))
    params = R.init_params(int(z["net_seed"]), 3, C)
    l1 = np.array([float(p.double().abs().sum()) for p in params])
    np.testing.assert_allclose(l1, z["params_before_l1"], rtol=1e-12)
    out = R.train_epochs(params, images, torch.arange(C), float(z["lr_net"]), epochs,
                         [list(range(C))] * (epochs + 1))
    assert len(out["loss"]) == epochs + 1
    np.testing.assert_allclose(out["loss"], z["train_loss"], rtol=2e-4)
    np.testing.assert_allclose(out["acc"], z["train_acc"], atol=1e-6)
    # lr drops by 10x after epoch Epoch//2+1 and stays
    assert out["lr"] == [0.01] * (epochs // 2 + 2) + [0.001] * (epochs - epochs // 2 - 1)
    l1 = np.array([float(p.double().abs().sum()) for p in out["params"]])
    np.testing.assert_allclose(l1, z["params_after_l1"], rtol=1e-5)
    assert z["top5"].shape == (4,)  # [acc, top1, top3, top5]
    assert z["top5"][0] == z["top5"][1] <= z["top5"][2] <= z["top5"][3]


def test_flop_count_matches_baseline_md():
    assert abs(R.dm_step_flops(50, 64, 1, 16, 112, 112) - 36.31e12) / 36.31e12 < 1e-3
    assert abs(R.dm_step_flops(50, 64, 1, 8, 64, 64) - 5.85e12) / 5.85e12 < 1e-3


def test_g9_gradient_matching_class_term(golden_dir):
    """Upstream-DC class term composed from the reference's get_network / match_loss (fixture G9):
    the oracle's double backward must reproduce loss and d loss / d syn for all three metrics."""
    z = load(golden_dir, "g9_grad_match.npz")
    C, lab = int(z["C"]), int(z["label"])
    real, syn = randn(z["data_seed"], (3, 8, 3, 64, 64), (2, 8, 3, 64, 64))
    params = [p.requires_grad_(True) for p in R.init_params(int(z["net_seed"]), 3, C)]
    lab_r, lab_s = torch.full((3,), lab), torch.full((2,), lab)
    gw_real = [t.detach() for t in torch.autograd.grad(F.cross_entropy(R.convnet3d_logits(real, params), lab_r), params)]
    np.testing.assert_allclose([float(t.double().abs().sum()) for t in gw_real], z["gw_real_l1"], rtol=1e-4)
    for metric in ("ours", "mse", "cos"):
        xs = syn.clone().requires_grad_(True)
        gw_syn = torch.autograd.grad(F.cross_entropy(R.convnet3d_logits(xs, params), lab_s), params, create_graph=True)
        loss = R.match_loss(list(gw_syn), gw_real, metric)
        (g,) = torch.autograd.grad(loss, xs)
        close(loss, z["loss_" + metric], rtol=2e-4)
        got = g[0] if metric == "ours" else g[:, 3]
        want = torch.tensor(z["grad_" + metric])
        assert float((got - want).norm() / want.norm()) < 2e-3, metric     # same arithmetic, different summation order


def _g10_inputs(z):
    C, n_syn = int(z["C"]), int(z["n_syn"])
    start = R.init_params(int(z["net_seed"]), 3, C)
    g = torch.Generator().manual_seed(int(z["data_seed"]))
    target = [p + 0.02 * p.abs().mean() * torch.randn(p.shape, generator=g) for p in start]
    image_syn = torch.randn(n_syn, 8, 3, 64, 64, generator=g)
    return start, target, image_syn, torch.tensor(z["labels"]), [torch.tensor(i) for i in z["indices"]]


def test_g10_mtt_step(golden_dir):
    """One MTT iteration re-stated around the reference's ReparamModule (fixture G10)."""
    z = load(golden_dir, "g10_mtt_step.npz")
    start, target, image_syn, labels, chunks = _g10_inputs(z)
    np.testing.assert_allclose(float(R.flatten_params(target).double().abs().sum()), float(z["target_l1"]), rtol=1e-6)
    grand, gx, glr = R.mtt_step(start, target, image_syn, labels, float(z["syn_lr"]), chunks)
    close(grand, z["grand_loss"], rtol=1e-4)
    close(glr, z["grad_lr"], rtol=2e-3)
    np.testing.assert_allclose([float(gx[b].double().abs().sum()) for b in range(gx.shape[0])], z["grad_l1"], rtol=2e-3)
    want = torch.tensor(z["grad_img"])
    assert float((gx[:, ::2, :, ::2, ::2] - want).norm() / want.norm()) < 2e-3


def test_g11_evaluate_synset_multi_static(golden_dir):
    """evaluate_synset(mode='multi-static') of the reference (fixture G11): per-item random choice of static
    image / dynamic memory / hallucinator (python `random`), loader shuffle from torch's global RNG."""
    import random
    z = load(golden_dir, "g11_multi_static_eval.npz")
    C, n_test, epochs, lr = int(z["C"]), int(z["n_test"]), int(z["epochs"]), float(z["lr_net"])
    g = torch.Generator().manual_seed(int(z["data_seed"]))
    static = torch.randn(C * 2, 3, 64, 64, generator=g)
    dynamic = torch.randn(C, 2, 8, 1, 64, 64, generator=g)
    hal_w, hal_b = torch.tensor(z["hal_w"]), torch.tensor(z["hal_b"])

    class Items(torch.utils.data.Dataset):      # utils.py:462-496 restated for spc/C == 2
        def __len__(self):
            return C

        def __getitem__(self, index):
            s_idx = random.randint(0, 1) + index * 2
            d_idx = random.randint(0, 1)
            h = random.randint(0, 1)
            return R.hallucinator(static[s_idx][None], dynamic[index, d_idx][None], hal_w[h], hal_b[h])[0], index
    params = [p.requires_grad_(True) for p in R.init_params(int(z["net_seed"]), 3, C)]
    torch.manual_seed(int(z["rng_seed"])); random.seed(int(z["rng_seed"]))
    loader = torch.utils.data.DataLoader(Items(), batch_size=256, shuffle=True)
    bufs, losses, accs = [None] * 8, [], []
    for ep in range(epochs + 1):
        for img, lab in loader:
            logits = R.convnet3d_logits(R.standardise_batch(img.float()), params)
            loss = F.cross_entropy(logits, lab)
            grads = torch.autograd.grad(loss, params)
            with torch.no_grad():
                for i, (p, gr) in enumerate(zip(params, grads)):
                    gr = gr + 0.0005 * p
                    bufs[i] = gr.clone() if bufs[i] is None else bufs[i] * 0.9 + gr
                    p -= lr * bufs[i]
            losses.append(float(loss)); accs.append(float((logits.argmax(1) == lab).float().mean()))
        if ep == epochs // 2 + 1:
            lr *= 0.1
            bufs = [None] * 8
    np.testing.assert_allclose(losses, z["train_loss"], rtol=2e-4)
    np.testing.assert_allclose(accs, z["train_acc"], atol=1e-6)
    np.testing.assert_allclose([float(p.double().abs().sum()) for p in params], z["params_after_l1"], rtol=1e-5)
