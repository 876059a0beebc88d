"""The deterministic (fixed summation order) training step: ``VD_DETERMINISTIC=1`` / ``hip.set_deterministic(True)``.

The reference's ``epoch('train')`` + ``optimizer.step()`` (utils.py:765-792) on a CPU gives the same result every time it is
run with one seed and thread count.  The HIP training step accumulates weight gradients, the logit conv's gradients, the
bias gradients and the batch statistics of the standardisation with fp32 / fp64 atomics, whose order -- and the sums' last
bits -- change from run to run; 500 SGD epochs on 50 clips amplify that into different networks.  In the ordered mode every
one of those sums has a fixed order: one accumulation copy per box of positions (an atomic add onto a zeroed word with ONE
contributor is exact) folded in index order, gathers instead of scatters in the head, per-workgroup partial sums folded by
one thread.  These tests assert BITWISE equality where the default mode's tests bound a spread."""
import ctypes
import types

import numpy as np
import pytest
import torch

from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a.cpu().double() - b.cpu().double()).norm() / b.cpu().double().norm())


@pytest.fixture
def ordered():
    from video_distillation_amd import hip
    prev = hip.set_deterministic(True)
    yield
    hip.set_deterministic(prev)


@pytest.mark.parametrize("geom,B,C", [((8, 64, 64), 6, 5), ((16, 112, 112), 5, 4)])
def test_gradients_are_bitwise_reproducible_and_equal_the_atomic_mode_to_rounding(geom, B, C):
    """All eight parameter gradients of ``TrainEngine.loss_and_grads``: five repetitions bitwise equal in the ordered mode, and
    within 2e-6 rel-L2 of the atomic mode's (the same products, summed in another order)."""
    from video_distillation_amd import hip, plan, train
    T, H, W = geom
    g = torch.Generator().manual_seed(606)
    x = train.standardize(torch.randn(B, T, 3, H, W, generator=g).cuda())
    y = (torch.arange(B) % C).cuda()
    params = [p.cuda() for p in R.init_params(61, 3, C)]
    pool = (2, 2, 2) if H > 64 else (2, 1, 1)
    te = train.TrainEngine(plan.NetGeometry(T, H, W), C, pool, "cuda:0", prec="f16x3", prec_bwd="f16x3")
    mask = (torch.rand(B, te.C, te.Tp, generator=g) < 0.5).float().cuda() * 2.0
    assert not hip.deterministic()
    _, _, grads = te.loss_and_grads(x, y, params, mask)
    atomic = [t.clone() for t in grads]
    prev = hip.set_deterministic(True)
    try:
        runs = []
        for _ in range(5):
            loss, logits, grads = te.loss_and_grads(x, y, params, mask)
            runs.append((float(loss), logits.clone(), [t.clone() for t in grads]))
        xs = [train.standardize(x * 3 + 1) for _ in range(3)]
    finally:
        hip.set_deterministic(prev)
    for r in runs[1:]:
        assert r[0] == runs[0][0] and torch.equal(r[1], runs[0][1])
        for i in range(8):
            assert torch.equal(r[2][i], runs[0][2][i]), "gradient %d differs between two runs of the ordered mode" % i
    assert torch.equal(xs[0], xs[1]) and torch.equal(xs[0], xs[2])
    spread = [_rel(a, b) for a, b in zip(atomic, runs[0][2])]
    print("ordered vs atomic gradients, rel-L2 per tensor: %s" % ["%.1e" % v for v in spread])
    assert max(spread) < 2e-6
    assert te._wgrad(0, B).replicas == te._wgrad(0, B).plan.nbox or not hip.deterministic()


def test_train_steps_are_bitwise_reproducible_across_a_laned_gm_step(ordered, monkeypatch):
    """Three ``hip_train_step``s (SGD with momentum and weight decay, dropout on, seeded) from the same initial weights: all
    eight parameter tensors bitwise equal between two runs of the process -- with a gradient-matching step on three class
    lanes in between (other streams, the cached engines' workspaces re-used).  The default mode's counterpart
    (test_train_step_is_undisturbed_by_a_laned_gm_step) can only bound the difference by the atomics' noise."""
    from video_distillation_amd import distill, networks, plan
    C, B = 3, 4
    g = torch.Generator().manual_seed(4242)
    x = torch.randn(B, 8, 3, 64, 64, generator=g).cuda()
    y = (torch.arange(B) % C).cuda()
    p0 = R.init_params(31, 3, C)

    def three_steps():
        torch.manual_seed(77)
        net = networks.ConvNet3D(3, C, 128, 3, 'relu', 'none', 'maxpooling', 8, (64, 64)).cuda().train()
        with torch.no_grad():
            for p, q in zip(net.parameters(), p0):
                p.copy_(q)
        opt = torch.optim.SGD(net.parameters(), lr=0.05, momentum=0.9, weight_decay=5e-4)
        assert net.hip_trainable(x, opt, torch.nn.CrossEntropyLoss().cuda())
        losses = []
        for _ in range(3):
            _, loss = net.hip_train_step(x, y, opt)
            losses.append(float(loss))
        torch.cuda.synchronize()
        return losses, [p.detach().clone() for p in net.parameters()]
    before = three_steps()
    monkeypatch.setenv("VD_GM_LANES", "3")
    clips = torch.randn(C * 4, 8, 3, 64, 64, generator=g).to("cuda:0")
    pool = distill.RealPool(clips, [4] * C, [4 * c for c in range(C)])
    tr = distill.GMTrainer(distill.HipGMOps("cuda:0", "ours"), pool, plan.NetGeometry(8, 64, 64), C, 1, batch_real=4, lr_img=1e-3,
                           outer_loop=1, dropout_p=0.0, net_init=lambda it: R.init_params(900 + it, 3, C))
    float(tr.step(0))
    after = three_steps()
    assert before[0] == after[0], (before[0], after[0])
    for a, b in zip(before[1], after[1]):
        assert torch.equal(a, b)


def test_evaluate_synset_is_reproducible_at_fifty_classes(ordered):
    """``evaluate_synset`` (utils.py:848-886) on 50 synthetic clips of 50 classes, 64x64x8, 30 epochs incl. the learning-rate
    switch, run twice from one seed: identical accuracies, identical final parameters (bitwise)."""
    from video_distillation_amd import networks, utils
    C = 50
    g = torch.Generator().manual_seed(11)
    base = torch.randn(C, 8, 3, 64, 64, generator=g)
    syn = base.cuda()
    labels = torch.arange(C).cuda()
    test_x = (base[:, None] + 0.8 * torch.randn(C, 2, 8, 3, 64, 64, generator=g)).reshape(-1, 8, 3, 64, 64).cuda()
    test_y = torch.arange(C).repeat_interleave(2).cuda()
    loader = torch.utils.data.DataLoader(utils.TensorDataset(test_x, test_y), batch_size=64, shuffle=False)
    eargs = types.SimpleNamespace(device="cuda:0", lr_net=0.01, epoch_eval_train=30, batch_train=256, model="ConvNet3D", eval_mode="SS")

    def run():
        torch.manual_seed(1000)
        np.random.seed(5)
        # (constructed directly: utils.get_network reseeds the global generator from the wall clock, like the reference's)
        net = networks.ConvNet3D(3, C, 128, 3, 'relu', 'none', 'maxpooling', 8, (64, 64)).to("cuda:0")
        _, acc_train, acc_test, _ = utils.evaluate_synset(0, net, syn, labels, loader, eargs, mode="none")
        return float(acc_train), float(acc_test), [p.detach().clone() for p in net.parameters()]
    a, b = run(), run()
    assert a[0] == b[0] and a[1] == b[1]
    for p, q in zip(a[2], b[2]):
        assert torch.equal(p, q)


def test_c_handle_training_step_is_bitwise_reproducible(ordered):
    """``vd_train_create`` under ``vd_set_deterministic(1)``: two handles, two steps each from the same state -- parameters and
    momentum buffers bitwise equal; the handle's copies (one per box) make its workspace larger than the atomic mode's."""
    from video_distillation_amd import hip
    L = hip.lib()
    C, B, T, H = 4, 6, 8, 64
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, T, 3, H, H, generator=g).cuda()
    y = (torch.arange(B) % C).cuda()
    p0 = R.init_params(5, 3, C)

    def run():
        h = ctypes.c_void_p()
        hip.check(L.vd_train_create(T, H, H, C, hip.PREC["f16x3"], hip.PREC["f16x3"], ctypes.c_int64(B), ctypes.byref(h)), "vd_train_create")
        try:
            nbytes = int(L.vd_train_workspace_bytes(h))
            ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
            params = [p.clone().cuda().contiguous() for p in p0]
            mom = [torch.zeros_like(p) for p in params]
            arr = ctypes.c_void_p * 8
            for step in range(2):
                hip.check(L.vd_train_step(h, arr(*[p.data_ptr() for p in params]), arr(*[m.data_ptr() for m in mom]), hip.ptr(x),
                                          hip.ptr(y), None, ctypes.c_float(0.05), ctypes.c_float(0.9), ctypes.c_float(5e-4),
                                          int(step == 0), hip.ptr(ws), ctypes.c_int64(nbytes), None, None, hip.stream_ptr()),
                          "vd_train_step")
            torch.cuda.synchronize()
            return params, mom, nbytes
        finally:
            L.vd_train_free(h)
    a, b = run(), run()
    for p, q in zip(a[0] + a[1], b[0] + b[1]):
        assert torch.equal(p, q)
    # ... and bitwise the Python engine's two steps: both planners emit byte-identical programs (tests/test_cplanner.py) and
    # every sum has the same fixed order on either side
    from video_distillation_amd import networks
    net = networks.ConvNet3D(3, C, 128, 3, 'relu', 'none', 'maxpooling', T, (H, H)).cuda().train()
    net.dropout.p = 0.0
    with torch.no_grad():
        for p, q in zip(net.parameters(), p0):
            p.copy_(q)
    opt = torch.optim.SGD(net.parameters(), lr=0.05, momentum=0.9, weight_decay=5e-4)
    assert net.hip_trainable(x, opt, torch.nn.CrossEntropyLoss().cuda())
    for _ in range(2):
        net.hip_train_step(x, y, opt)
    torch.cuda.synchronize()
    diffs = [_rel(p, q) for p, q in zip(a[0], net.parameters())]
    print("C handle vs Python engine, ordered mode, rel-L2 per tensor after two steps: %s" % ["%.1e" % v for v in diffs])
    assert all(torch.equal(p, q.detach()) for p, q in zip(a[0], net.parameters()))
    prev = hip.set_deterministic(False)
    try:
        h = ctypes.c_void_p()
        hip.check(L.vd_train_create(T, H, H, C, hip.PREC["f16x3"], hip.PREC["f16x3"], ctypes.c_int64(B), ctypes.byref(h)), "vd_train_create")
        atomic_bytes = int(L.vd_train_workspace_bytes(h))
        L.vd_train_free(h)
    finally:
        hip.set_deterministic(prev)
    assert a[2] > atomic_bytes
