"""Dithered single-pass weights of the real side (vd_pack_weights_dither, engine.dither_groups, HipBackend.embed_pool):
kernel semantics against a numpy restatement, and the effect it exists for -- the weight-rounding perturbation of a class's
MEAN feature -- against the fp32 oracle, next to plain rn16 weights."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu


def _neighbours(w, dt):
    q = w.to(dt)
    qf = q.float()
    up = torch.nextafter(q, torch.full_like(q, float("inf"))).float()
    dn = torch.nextafter(q, torch.full_like(q, float("-inf"))).float()
    return torch.where(qf <= w, qf, dn), torch.where(qf >= w, qf, up)


def _bitrev(g, G):
    b = G.bit_length() - 1
    return int(format(g, "0%db" % b)[::-1], 2) if b else 0


@pytest.mark.parametrize("prec,dt", [("f16", torch.float16), ("bf16", torch.bfloat16)])
@pytest.mark.parametrize("G", [4, 8, 16])
def test_dither_kernel_matches_restatement_and_is_unbiased(prec, dt, G):
    from video_distillation_amd import hip
    g = torch.Generator().manual_seed(G)
    n = 5000
    w = torch.randn(n, generator=g) * 0.02
    w[:8] = torch.tensor([0.0, -0.0, 1.0, -0.5, 6.0e-8, -6.0e-8, 1e-3, 65504.0])           # zero, representable, subnormal, largest half
    widx = torch.arange(n, dtype=torch.int32)
    widx[100] = -1                                                                            # padding entry of a gather table -> 0
    out = torch.empty((G, n), dtype=torch.int16, device="cuda")
    wd, id_ = w.cuda(), widx.cuda()           # (kept alive: a temporary's block would be handed to the next allocation)
    hip.check(hip.lib().vd_pack_weights_dither(hip.ptr(wd), hip.ptr(id_), ctypes.c_int64(n), G, hip.ptr(out),
                                               hip.PREC[prec], hip.stream_ptr(torch.device("cuda:0"))), "dither")
    got = out.cpu().view(dt).float()
    src = w.clone(); src[100] = 0.0
    lo, hi = _neighbours(src, dt)
    lam = torch.where(hi > lo, (src - lo) / (hi - lo), torch.zeros_like(src))
    k = torch.arange(n, dtype=torch.int64); k[100] = -1
    rot = (((k * 2654435761) % 4294967296) >> 7) % G
    for gi in range(G):
        slot = (_bitrev(gi, G) + rot) % G
        want = torch.where((slot.float() + 0.5) / G < lam, hi, lo)
        assert torch.equal(got[gi], want), (prec, G, gi)
    ulp = (hi - lo)
    assert float(((got.double().mean(0) - src.double()).abs() - ulp.double() / (2 * G)).max()) <= 1e-12     # mean over groups = w to ulp/(2G)
    assert torch.equal(got[:, 2], torch.full((G,), 1.0)) and torch.equal(got[:, 100], torch.zeros(G))
    with pytest.raises(RuntimeError):
        hip.check(hip.lib().vd_pack_weights_dither(hip.ptr(wd), hip.ptr(id_), ctypes.c_int64(n), 6, hip.ptr(out),
                                                   hip.PREC[prec], hip.stream_ptr(torch.device("cuda:0"))), "dither")


def test_dither_groups_rule(monkeypatch):
    from video_distillation_amd.engine import dither_groups
    assert [dither_groups(n, "f16") for n in (64, 8, 4, 12, 6, 3, 1, 0)] == [8, 8, 4, 4, 0, 0, 0, 0]
    assert dither_groups(64, "f16x3") == 0 and dither_groups(64, "bf16") == 8
    monkeypatch.setenv("VD_REAL_DITHER", "16")
    assert dither_groups(64, "f16") == 16 and dither_groups(8, "f16") == 8
    monkeypatch.setenv("VD_REAL_DITHER", "0")
    assert dither_groups(64, "f16") == 0


@pytest.mark.parametrize("geom,C,n", [((8, 64, 64), 2, 16), ((16, 112, 112), 1, 8)])
def test_class_mean_bias_of_the_real_side_with_and_without_dither(monkeypatch, geom, C, n):
    """Similar clips (the late regime of a distillation: the case plain rn16 weights are worst at): error of the class-mean
    feature of the single-pass real side against the fp32 oracle, plain vs dithered; CPU-config and full-size geometry."""
    from video_distillation_amd import distill, plan
    geo = plan.NetGeometry(*geom)
    T, H, W = geom
    g = torch.Generator().manual_seed(5)
    base = torch.randn(C, 1, T, 3, H, W, generator=g)
    pool = (base + 0.1 * torch.randn(C, n, T, 3, H, W, generator=g)).reshape(C * n, T, 3, H, W)
    params = R.init_params(1234, 3, 5)
    with torch.no_grad():
        want = R.convnet3d_embed(pool, params).view(C, n, -1).mean(1)
    idx = torch.arange(C * n, device="cuda")
    err = {}
    for setting in ("0", "8"):
        monkeypatch.setenv("VD_REAL_DITHER", setting)
        be = distill.HipBackend(geo, "cuda:0")
        be.set_real_weights([p.cuda() for p in params[:6]], n)
        assert be._dither == (8 if setting == "8" else 0)
        f = be.embed_pool(pool.cuda(), idx, n).cpu().view(C, n, -1).mean(1)
        err[setting] = float((f - want).norm() / want.norm())
    print("%s class-mean feature error of the f16 real side: plain rn16(W) %.2e, 8 dither groups %.2e" % (geom, err["0"], err["8"]))
    assert err["8"] < 0.6 * err["0"] and err["8"] < 1.2e-4


@pytest.mark.parametrize("geom,n", [((8, 64, 64), 8), ((16, 112, 112), 16)])
def test_one_launch_with_set_selection_equals_one_launch_per_group(geom, n):
    """EmbedEngine.forward_sets (layers 1 and 2: ONE launch, the operand set picked per box from the clip number) against G
    separate forwards with ``group=g``: bitwise the same features; resident 16-bit rows and fp32 clips as the source."""
    from video_distillation_amd import distill, plan
    geo = plan.NetGeometry(*geom)
    g = torch.Generator().manual_seed(n)
    C = 3
    pool = torch.randn(C * n + 5, geom[0], 3, geom[1], geom[2], generator=g).cuda()
    params = [p.cuda() for p in R.init_params(77, 3, 5)[:6]]
    be = distill.HipBackend(geo, "cuda:0")
    be.set_real_weights(params, n)
    G = be._dither
    assert G == 8
    idx = torch.randperm(C * n + 5, generator=g)[:C * n].cuda()
    for resident in (True, False):
        be.resident_rows = resident
        got = be.embed_pool(pool, idx, n)
        want = torch.empty_like(got)
        for gi in range(G):
            sel = idx.view(C, n // G, G)[:, :, gi].reshape(-1)
            want.view(C, n // G, G, -1)[:, :, gi] = be.eng_real.forward(pool, index=sel, group=gi).view(C, n // G, -1)
        assert torch.equal(got, want)
    plain = be.eng_real.forward(pool, index=idx)
    assert not torch.equal(plain, got) and float((plain - got).norm() / plain.norm()) < 1e-3


def test_module_embed_value_pass_follows_real_dither_and_the_weights():
    """networks.ConvNet3D.embed in the mixed mode: which forward the gradient-carrying clips' FEATURES come from --
    exact weights (the real side they meet was dithered) or rn16 weights (the value pass) -- follows ``net.real_dither``:
    "auto" remembers a dithered no-gradient batch only for the weights it ran with and is not disturbed by inference passes;
    True / False state it explicitly (call orders other than the reference's real-then-synthetic)."""
    from video_distillation_amd import engine, networks, plan
    torch.manual_seed(3)
    net = networks.ConvNet3D(3, 5, 128, 3, 'relu', 'none', 'maxpooling', 8, (64, 64)).cuda().train()
    for p in net.parameters():
        p.requires_grad = False
    g = torch.Generator().manual_seed(4)
    real = torch.randn(8, 8, 3, 64, 64, generator=g).cuda()
    syn = torch.randn(1, 8, 3, 64, 64, generator=g).cuda()
    eng = engine.EmbedEngine(plan.NetGeometry(8, 64, 64), prec="f16x3")

    # the value pass rounds the weights of the levels the undithered real side multiplies by plain rn16(W): all three with a
    # single-pass last level, levels 0 / 1 when the real side's last level runs on exact hi+lo weights (real_last = x3, the default)
    levels = (0, 1) if networks.get_precision()["real_last"] == "x3" else (0, 1, 2)

    def expect(quantize):
        eng.set_weights(net._feature_params(), quantize=quantize, quantize_levels=levels)
        return eng.forward(syn)

    def syn_feats():
        return net.embed(syn.clone().requires_grad_(True)).detach()
    assert net.real_dither == "auto"
    assert torch.equal(syn_feats(), expect("f16"))                 # nothing dithered yet: value pass
    net.embed(real)                                                # 8 clips without gradient: 8 dither groups
    assert torch.equal(syn_feats(), expect(None))
    with torch.no_grad():
        net.eval(); net(real[:2]); net.train()                     # an inference pass in between does not disturb it
    assert torch.equal(syn_feats(), expect(None))
    net.embed(real[:3])                                            # a real batch too small to dither: value pass again
    assert torch.equal(syn_feats(), expect("f16"))
    net.embed(real)
    with torch.no_grad():
        net.features[0].weight.mul_(1.5)                           # a weight update forgets the dithered batch
    assert torch.equal(syn_feats(), expect("f16"))
    net.real_dither = True                                         # synthetic clips first, real clips afterwards
    assert torch.equal(syn_feats(), expect(None))
    net.real_dither = False
    f_plain = net.embed(real)
    eng16 = engine.EmbedEngine(plan.NetGeometry(8, 64, 64), prec="f16", last_hilo=(networks.get_precision()["real_last"] == "x3"))
    eng16.set_weights(net._feature_params())
    assert torch.equal(f_plain, eng16.forward(real))               # never dithered (plain rn16 weights in the single-pass levels)
    assert torch.equal(syn_feats(), expect("f16"))


@pytest.mark.parametrize("geom,n", [((8, 64, 64), 32), ((16, 112, 112), 8)])
def test_last_level_in_hi_lo_pairs(geom, n):
    """EmbedEngine(last_hilo=True): level 1's single-pass program also writes the LOW plane of its pooled outputs
    (VdConvParams.emit_lo) and level 2 multiplies hi+lo activations by hi+lo weights.  (a) the high plane is bitwise what the
    plain single-pass program writes and hi + lo reproduces the fp32 pooled value to 2^-22; (b) the features are closer to
    the fp32 oracle than the all-single-pass engine's, per clip and on the class mean; (c) chunked == unchunked."""
    from video_distillation_amd import engine, plan
    geo = plan.NetGeometry(*geom)
    T, H, W = geom
    g = torch.Generator().manual_seed(11)
    x = (torch.randn(1, T, 3, H, W, generator=g) + 0.1 * torch.randn(n, T, 3, H, W, generator=g))
    params = R.init_params(4321, 3, 5)
    with torch.no_grad():
        want = R.convnet3d_embed(x, params)
    w = [p.cuda() for p in params[:6]]
    e1 = engine.EmbedEngine(geo, prec="f16", chunk=4096); e1.set_weights(w)
    e3 = engine.EmbedEngine(geo, prec="f16", chunk=4096, last_hilo=True); e3.set_weights(w)
    f1, f3 = e1.forward(x.cuda()), e3.forward(x.cuda())
    per2 = int(np.prod(e3.fwd[1].plan.out_shape[:-1]))
    hi3 = e3._ws["act2"][:2 * n * per2 * 8].view(2, n * per2, 8)[0].clone()
    lo3 = e3._ws["act2"][:2 * n * per2 * 8].view(2, n * per2, 8)[1].clone()
    hi1 = e1._ws["act2"][:n * per2 * 8].view(n * per2, 8)
    assert torch.equal(hi3, hi1)
    ex = engine.EmbedEngine(geo, prec="f16x3", chunk=4096); ex.set_weights(w)       # the same pooled values from an f16x3 level 1 ...
    ex.forward(x.cuda())
    a_hi = ex._ws["act2"][:2 * n * per2 * 8].view(2, n * per2, 8)
    exact = a_hi[0].view(torch.float16).double() + a_hi[1].view(torch.float16).double()
    got = hi3.view(torch.float16).double() + lo3.view(torch.float16).double()
    single = hi3.view(torch.float16).double()
    # ... differ from ours by levels 0 / 1 running single pass; what matters here: hi + lo carries the value far below f16's 2^-11
    assert bool((lo3.view(torch.float16).double().abs() <= 2.0 ** -11 * single.abs() + 2.0 ** -24).all())          # half an ulp
    assert float((got - exact).norm() / exact.norm()) < 1.05 * float((single - exact).norm() / exact.norm())
    e_clip = [float(((f.cpu() - want).norm(dim=1) / want.norm(dim=1)).mean()) for f in (f1, f3)]
    e_mean = [float((f.cpu().mean(0) - want.mean(0)).norm() / want.mean(0).norm()) for f in (f1, f3)]
    print("%s last level x1 / hi+lo: per-clip feature error %.2e / %.2e, class-mean error %.2e / %.2e" % (geom, *e_clip, *e_mean))
    assert e_clip[1] < 0.92 * e_clip[0] and e_clip[1] < 3e-4
    e3c = engine.EmbedEngine(geo, prec="f16", chunk=3, last_hilo=True); e3c.set_weights(w)
    assert torch.equal(e3c.forward(x.cuda()), f3)
    with pytest.raises(AssertionError):
        e3.forward(x.cuda(), keep=True)
    with pytest.raises(ValueError):
        engine.EmbedEngine(geo, prec="f16x3", last_hilo=True)
