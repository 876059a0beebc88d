"""GPU parity of the HIP embed path (EmbedEngine -> C ABI -> conv_mfma.hip) against the CPU oracle.

Tolerances: the x3 split precisions are fp32-class (<= 2e-5 of the feature scale; north_star
asks for 1e-3 relative on the loss); single-pass f16 / bf16 are reported perf modes with the
error of their 11 / 8 bit operands."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu

TOL = {"bf16x3": 3e-5, "f16x3": 3e-5, "f16": 2e-3, "bf16": 2e-2}


def _engine(geo, prec):
    from video_distillation_amd import engine, plan
    return engine.EmbedEngine(plan.NetGeometry(*geo), prec=prec, device="cuda:0")


def _rel(a, b):
    a = a.double().cpu(); b = b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30)), float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize("prec", ["bf16x3", "f16x3", "f16", "bf16"])
def test_embed_forward_small(prec):
    params = R.init_params(3)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(5, 8, 3, 64, 64, generator=g)
    want = R.convnet3d_embed(x, params)
    eng = _engine((8, 64, 64), prec)
    eng.set_weights([p.cuda() for p in params])
    got = eng.forward(x.cuda())
    torch.cuda.synchronize()
    rl2, rmax = _rel(got, want)
    print(prec, "fwd rel-l2 %.3e rel-max %.3e" % (rl2, rmax))
    assert rl2 < TOL[prec] and rmax < 4 * TOL[prec]


def _grad_fp64(x, gf, params):
    """fp64 input gradient.  The fp32 reference itself differs from this by ~4e-4 rel-l2: a
    max-pool window whose two largest entries tie to within rounding routes its gradient to a
    different position depending on summation order ("arg-max flips"), so gradient parity is
    stated against fp64 and allows a small fraction of flipped windows."""
    xr = x.double().clone().requires_grad_(True)
    (R.convnet3d_embed(xr, [p.double() for p in params]) * gf.double()).sum().backward()
    return xr.grad


def _grad_check(dx, want, prec):
    rl2, rmax = _rel(dx, want)
    d = (dx.double().cpu() - want).abs()
    frac_bad = float((d > 1e-3 * want.abs().max()).double().mean())
    print(prec, "bwd vs fp64: rel-l2 %.3e rel-max %.3e flipped-frac %.2e" % (rl2, rmax, frac_bad))
    if prec == "f16x3":
        assert rl2 < 1e-3 and frac_bad < 1e-3      # 22-bit operands: no worse than the fp32 reference itself
    else:
        assert rl2 < 1e-2 and frac_bad < 1e-2
    return rl2


@pytest.mark.parametrize("prec", ["bf16x3", "f16x3"])
def test_embed_backward_small(prec):
    params = R.init_params(4)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(3, 8, 3, 64, 64, generator=g)
    gf = torch.randn(3, 256, generator=g)
    want = _grad_fp64(x, gf, params)
    eng = _engine((8, 64, 64), prec)
    eng.set_weights([p.cuda() for p in params])
    _, sv = eng.forward(x.cuda(), keep=True)
    dx = eng.backward(sv, gf.cuda())
    torch.cuda.synchronize()
    _grad_check(dx, want, prec)
    # the fp32 CPU reference path is itself no closer to fp64 than we are (x10 slack)
    xr = x.clone().requires_grad_(True)
    (R.convnet3d_embed(xr, params) * gf).sum().backward()
    ref_err = _rel(xr.grad, want)[0]
    assert _rel(dx, want)[0] < 10 * ref_err + 1e-5


@pytest.mark.parametrize("prec", ["bf16x3", "f16"])
def test_embed_full_resolution_golden(prec, golden_dir):
    import os
    z = np.load(os.path.join(golden_dir, "g1_layers.npz"))
    params = R.init_params(int(z["seed"]))
    g = torch.Generator().manual_seed(int(z["x112_seed"]))
    x = torch.randn(1, 16, 3, 112, 112, generator=g)
    eng = _engine((16, 112, 112), prec)
    eng.set_weights([p.cuda() for p in params])
    xx = torch.cat([x, x.flip(0) * 0.5, x * -1.0]).cuda()      # 3 clips: ragged clip group for L2 (ncl=2)
    got = eng.forward(xx)
    torch.cuda.synchronize()
    rl2, rmax = _rel(got[0], torch.tensor(z["embed112"][0]))
    print(prec, "112 fwd vs reference golden rel-l2 %.3e" % rl2)
    assert rl2 < TOL[prec]
    want2 = R.convnet3d_embed(xx[2:3].cpu(), params)
    assert _rel(got[2:3], want2)[0] < TOL[prec]


def test_embed_backward_full_resolution():
    params = R.init_params(8)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 16, 3, 112, 112, generator=g)
    gf = torch.randn(2, 2048, generator=g)
    want = _grad_fp64(x, gf, params)
    eng = _engine((16, 112, 112), "f16x3")
    eng.set_weights([p.cuda() for p in params])
    _, sv = eng.forward(x.cuda(), keep=True)
    dx = eng.backward(sv, gf.cuda())
    torch.cuda.synchronize()
    _grad_check(dx, want, "f16x3")


def test_zero_and_single_clip_edge_cases():
    params = R.init_params(2)
    eng = _engine((8, 64, 64), "bf16x3")
    eng.set_weights([p.cuda() for p in params])
    z = eng.forward(torch.zeros(1, 8, 3, 64, 64).cuda())
    want = R.convnet3d_embed(torch.zeros(1, 8, 3, 64, 64), params)   # bias-only network response
    torch.cuda.synchronize()
    assert _rel(z, want)[0] < 3e-5
    e = eng.forward(torch.zeros(0, 8, 3, 64, 64).cuda())
    assert e.shape == (0, 256)


def test_full_size_properties_chunking_gather_and_streams():
    """Size-independent properties at the benchmark's full clip size (no CPU reference needed):
    chunking invariance, fused index gather == materialised gather, batch independence,
    two-stream overlapped trainer == serial trainer (bitwise)."""
    from video_distillation_amd import distill, engine, plan
    geo = plan.NetGeometry(16, 112, 112)
    g = torch.Generator(device="cuda").manual_seed(3)
    pool = torch.randn(12, 16, 3, 112, 112, device="cuda", generator=g)
    w = distill.fresh_network_weights(5, "cuda:0")
    e1 = engine.EmbedEngine(geo, prec="f16", chunk=512); e1.set_weights(w)
    e2 = engine.EmbedEngine(geo, prec="f16", chunk=5); e2.set_weights(w)      # ragged chunks: 5 + 5 + 2
    f1 = e1.forward(pool)
    f2 = e2.forward(pool)
    assert torch.equal(f1, f2)
    idx = torch.tensor([7, 0, 11, 3, 3, 9], device="cuda")
    assert torch.equal(e1.forward(pool, index=idx), f1[idx])                     # gather fused into the first kernel
    assert torch.equal(e1.forward(pool[4:5]), f1[4:5])                           # clips are independent
    assert torch.isfinite(f1).all() and float(f1.abs().sum()) > 0

    def run(two_streams):
        be = distill.HipBackend(geo, "cuda:0", prec_real="f16", prec_syn="f16x3", chunk=512)
        be.two_streams = be.two_streams and two_streams
        rp = distill.RealPool(pool, [4, 4, 4], [0, 4, 8])
        tr = distill.DMTrainer(be, rp, 3, 1, 3, lr_img=0.5)
        losses = [tr.step(it, overlap=two_streams) for it in range(3)]
        tr.sync(); torch.cuda.synchronize()
        return [float(l) for l in losses], tr.image_syn.clone()
    la, sa = run(False)
    lb, sb = run(True)
    assert la == lb and torch.equal(sa, sb)
    assert la[0] > 0 and not torch.equal(sa, pool[[0, 4, 8]])                    # the step did move the pixels


def test_first_layer_two_tile_wave_layout_agrees_with_frame_tiles():
    """The first level in the round-4 layout (frame-pair row groups, here with two N tiles per wave: ntw0 = 2, generic kernel)
    against the frame-tile program of round 5: other row -> lane maps and another K order per output, the same convolution --
    equal to fp32 summation-order rounding."""
    from video_distillation_amd import distill, engine, plan
    geo = plan.NetGeometry(16, 112, 112)
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn(3, 16, 3, 112, 112, device="cuda", generator=g)
    eng = engine.EmbedEngine(geo, prec="f16", chunk=8, ntw0=1)
    eng.set_weights(distill.fresh_network_weights(2, "cuda:0"))
    assert eng.fwd[0].plan.pair_flip & 0xF == plan.FRAME_TILE_FLIP and eng.fwd[0].breg_ok
    ref = eng.forward(x)
    eng2 = engine.EmbedEngine(geo, prec="f16", chunk=8, ntw0=2)     # 1 wave column x 4 wave rows, 2 N tiles per wave
    eng2.set_weights(distill.fresh_network_weights(2, "cuda:0"))
    assert eng2.fwd[0].plan.NTW == 2 and eng2.fwd[0].plan.pair_flip == 0 and not eng2.fwd[0].breg_ok
    got = eng2.forward(x)
    torch.cuda.synchronize()
    rel = float((got - ref).norm() / ref.norm())
    print("frame tiles vs round-4 layout: features rel-L2 %.2e" % rel)
    assert rel < 3e-4 and not torch.equal(got, ref)         # (fp32 order differences that cross an f16 rounding boundary of a level-0 / level-1 output move that output by one ulp)


@pytest.mark.parametrize("geom,n", [((16, 112, 112), 5), ((8, 64, 64), 37)])
def test_first_layer_kernel_variants_are_bitwise_equal(monkeypatch, geom, n):
    """The first-layer kernels of the single-pass formats -- generic tile program (VD_L0_BREG=0), register-resident B in one
    eight-wave workgroup per CU whose two groups alternate K loop / everything else with one LDS read per MFMA (4), and the same
    with every A fragment of the frame-tile program read once for the up to three tiles it serves (5, default) -- run the same
    tile program with the same K order per output: bitwise equal features, also through the index gather and with dithered
    operand sets switching inside a workgroup's box walk."""
    from video_distillation_amd import distill, engine, plan
    geo = plan.NetGeometry(*geom)
    g = torch.Generator(device="cuda").manual_seed(21)
    pool = torch.randn(n + 3, geom[0], 3, geom[1], geom[2], device="cuda", generator=g)
    idx = torch.randperm(n + 3, generator=torch.Generator().manual_seed(3))[:n - n % 8 if n >= 8 else n].cuda()
    w = distill.fresh_network_weights(9, "cuda:0")
    outs = {}
    for variant in ("0", "4", "5"):
        monkeypatch.setenv("VD_L0_BREG", variant)
        eng = engine.EmbedEngine(geo, prec="f16", chunk=4096, ntw0=1)
        assert eng.fwd[0].breg_ok == (variant != "0") and eng.fwd[0].breg_variant == int(variant)
        assert eng.fwd[0].plan.pair_flip & 0xF == plan.FRAME_TILE_FLIP and eng.fwd[0].plan.pair_flip >> 8 == 9 | (189 << 8)
        G = 8 if idx.numel() % 8 == 0 and idx.numel() >= 8 else 0
        eng.set_weights(w, dither=G)
        rows = eng.pool_rows(pool)
        outs[variant] = (eng.forward(pool), eng.forward(pool, index=idx, rows=rows),
                         eng.forward_sets(pool, idx, rows=rows) if G else None)
    for variant in ("4", "5"):
        for a, b in zip(outs["0"], outs[variant]):
            assert (a is None and b is None) or torch.equal(a, b), variant


@pytest.mark.parametrize("geom,n,prec", [((8, 64, 64), 40, "f16x3"), ((8, 64, 64), 40, "f16"), ((16, 112, 112), 5, "f16x3"),
                                         ((12, 96, 80), 9, "bf16x3"), ((8, 64, 64), 24, "bf16x3")])
def test_parity_classes_in_one_launch_are_bitwise_equal(monkeypatch, geom, n, prec):
    """The four parity-class programs of an input-gradient pass as ONE launch (engine.run_together -> vd_conv_mfma_multi;
    programs of different tile shapes fall into separate launches) against four vd_conv_mfma launches: the same tile programs,
    the same K order per output -- bitwise equal pixel gradients; likewise the accumulating second launches of the
    second-order down sweep (train.GradMatchEngine.vjp: one atomic add per output on top of a plain store)."""
    from video_distillation_amd import distill, engine, plan, train
    geo = plan.NetGeometry(*geom)
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(n, geom[0], 3, geom[1], geom[2], device="cuda", generator=g)
    w = distill.fresh_network_weights(4, "cuda:0")
    eng = engine.EmbedEngine(geo, prec=prec, chunk=4096, prec_bwd=prec)
    eng.set_weights(w)
    gf = torch.randn(n, eng.num_feat, device="cuda", generator=g)
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("VD_MULTI_LAUNCH", mode)
        f, sv = eng.forward(x, keep=True)
        outs[mode] = eng.backward(sv, gf).clone()
    assert torch.equal(outs["0"], outs["1"])
    assert float(outs["1"].abs().max()) > 0
    if prec != "bf16x3" or geom[1] != 64:
        return
    K = 5
    full = distill.fresh_full_network(4, K, "cuda:0")
    labels = torch.arange(n, device="cuda") % K
    gm = train.GradMatchEngine(geo, K, (2, 2, 2) if geom[1] > 64 else (2, 1, 1), "cuda:0")
    v = [torch.randn(p.shape, device="cuda", generator=g) * 0.1 for p in full]
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("VD_MULTI_LAUNCH", mode)
        _, _, _, state = gm.param_grads(x, labels, full)
        res[mode] = gm.vjp(state, v, full).clone()
    assert float(res["1"].abs().max()) > 0
    assert float((res["0"] - res["1"]).norm() / res["1"].norm()) < 1e-6          # (weight-gradient atomics do not feed dx: expected 0)


def test_odd_geometry_on_device():
    """12 x 96 x 80 clips on the GPU: odd conv extents (floor pooling), multi-type box plans, 7 clips per
    workgroup in the last layer's input-gradient passes."""
    params = R.init_params(12)
    g = torch.Generator().manual_seed(13)
    x = torch.randn(9, 12, 3, 96, 80, generator=g)
    gf = torch.randn(9, 384, generator=g)
    want_f = R.convnet3d_embed(x, params)
    want_g = _grad_fp64(x, gf, params)
    eng = _engine((12, 96, 80), "f16x3")
    eng.set_weights([p.cuda() for p in params])
    f, sv = eng.forward(x.cuda(), keep=True)
    dx = eng.backward(sv, gf.cuda())
    torch.cuda.synchronize()
    assert f.shape == (9, 384)
    assert _rel(f, want_f)[0] < 3e-5
    # per clip: exact up to rounding unless an arg-max tie flipped inside that clip (see _grad_fp64)
    per = [_rel(dx[i], want_g[i])[0] for i in range(9)]
    print("odd geometry bwd per-clip rel-l2:", ["%.1e" % v for v in per])
    assert sum(v < 1e-4 for v in per) >= 6 and max(per) < 2e-2
    eng2 = _engine((12, 96, 80), "f16")
    eng2.set_weights([p.cuda() for p in params])
    assert _rel(eng2.forward(x.cuda()), want_f)[0] < 2e-3


def test_single_pass_fp16_backward_with_dynamic_scaling():
    """prec_bwd='f16': one MFMA per product in the input-gradient passes, gradients scaled by a power
    of two per layer (vd_absmax_scale) so they stay inside fp16's exponent range.  Same forward (f16x3)
    => same arg-max decisions => the difference to the x3 backward is pure operand rounding (~2^-11),
    also for gradients far below fp16's smallest normal."""
    from video_distillation_amd import engine, plan
    params = R.init_params(4)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(3, 8, 3, 64, 64, generator=g)
    gf = torch.randn(3, 256, generator=g)
    geo = plan.NetGeometry(8, 64, 64)
    e3 = engine.EmbedEngine(geo, prec="f16x3"); e3.set_weights([p.cuda() for p in params])
    e1 = engine.EmbedEngine(geo, prec="f16x3", prec_bwd="f16"); e1.set_weights([p.cuda() for p in params])
    for mag in (1.0, 1e-7, 1e4):
        _, s3 = e3.forward(x.cuda(), keep=True)
        _, s1 = e1.forward(x.cuda(), keep=True)
        d3 = e3.backward(s3, (gf * mag).cuda())
        d1 = e1.backward(s1, (gf * mag).cuda())
        torch.cuda.synchronize()
        rl2, rmax = _rel(d1, d3)
        print("f16 bwd vs f16x3 bwd at |g|~%g: rel-l2 %.2e rel-max %.2e" % (mag, rl2, rmax))
        assert rl2 < 1e-3 and torch.isfinite(d1).all()
    want = _grad_fp64(x, gf, params)
    assert _rel(e1.backward(e1.forward(x.cuda(), keep=True)[1], gf.cuda()), want)[0] < 2e-3
    # the x3 backward is scaled too: a 1e-7 gradient is reproduced as exactly as an O(1) one
    tiny = e3.backward(e3.forward(x.cuda(), keep=True)[1], (gf * 1e-7).cuda())
    assert _rel(tiny * 1e7, want)[0] < 1e-3


def test_resident_pool_rows_equal_per_step_conversion():
    """Batches read through an index out of the pool converted once to 16-bit pixel rows (HipBackend's
    default for the real clips) == converting the gathered fp32 clips every step (bitwise), both layouts
    of the first-layer program; the two-stream trainer gives the same step either way."""
    from video_distillation_amd import distill, engine, plan
    geo = plan.NetGeometry(16, 112, 112)
    g = torch.Generator(device="cuda").manual_seed(21)
    pool = torch.randn(10, 16, 3, 112, 112, device="cuda", generator=g)
    idx = torch.tensor([7, 0, 3, 3, 9, 1, 2], device="cuda")
    w = distill.fresh_network_weights(4, "cuda:0")
    for prec, ntw0 in (("f16", 2), ("f16", 1), ("f16x3", 1)):
        eng = engine.EmbedEngine(geo, prec=prec, chunk=4, ntw0=ntw0)
        eng.set_weights(w)
        rows = eng.pool_rows(pool, step=4)
        a = eng.forward(pool, index=idx)
        b = eng.forward(pool, index=idx, rows=rows)
        c = eng.forward(pool[idx])
        torch.cuda.synchronize()
        assert torch.equal(a, b) and torch.equal(a, c), (prec, ntw0)


@pytest.mark.parametrize("geom", [(8, 112, 112), (16, 64, 64), (16, 128, 128), (4, 64, 96), (8, 80, 112)])
def test_other_geometries_forward_and_input_gradient(geom):
    """The planner picks wave layouts per geometry (two N tiles per wave where boxes fill 8 M tiles, the balanced
    7-tile layout, multi-type box plans at odd extents): forward in both operand classes and the input gradient
    against the oracle on clip sizes other than the two the configs use."""
    T, H, W = geom
    params = R.init_params(31)
    g = torch.Generator().manual_seed(T * H + W)
    x = torch.randn(3, T, 3, H, W, generator=g)
    want = R.convnet3d_embed(x, params)
    gf = torch.randn(want.shape, generator=g)
    for prec in ("f16", "f16x3"):
        eng = _engine(geom, prec)
        eng.set_weights([p.cuda() for p in params])
        got = eng.forward(x.cuda())
        rel = _rel(got, want)[0]
        lay = [(pl.plan.NTW, pl.plan.MW, pl.plan.MTW) for pl in eng.fwd]
        print(geom, prec, "layouts (NTW,MW,MTW)", lay, "fwd rel-l2 %.2e" % rel)
        assert got.shape == want.shape and rel < TOL[prec]
    f, sv = eng.forward(x.cuda(), keep=True)
    dx = eng.backward(sv, gf.cuda())
    torch.cuda.synchronize()
    per = [_rel(dx[i], _grad_fp64(x[i:i + 1], gf[i:i + 1], params)[0])[0] for i in range(3)]
    print("   input gradient per-clip rel-l2 vs fp64:", ["%.1e" % v for v in per])
    assert sum(v < 1e-4 for v in per) >= 2 and max(per) < 3e-2       # arg-max flips confined to single clips


def test_latency_oriented_programs_equal_throughput_programs():
    """batch_hint only changes the decomposition into workgroups (plan.latency_variant): forward features and
    the input gradient must be bitwise those of the default programs (same K order per output)."""
    from video_distillation_amd import distill, engine, plan
    for geom in ((16, 112, 112), (8, 64, 64)):
        geo = plan.NetGeometry(*geom)
        g = torch.Generator(device="cuda").manual_seed(5)
        x = torch.randn(5, geom[0], 3, geom[1], geom[2], device="cuda", generator=g)
        gf = torch.randn(5, geo.num_feat, device="cuda", generator=g)
        w = distill.fresh_network_weights(3, "cuda:0")
        outs = []
        for hint in (None, 8):
            eng = engine.EmbedEngine(geo, prec="f16x3", chunk=8, prec_bwd="f16", batch_hint=hint)
            eng.set_weights(w)
            f, sv = eng.forward(x, keep=True)
            outs.append((f, eng.backward(sv, gf), [(p.plan.MTW, p.plan.ncl) for p in eng.fwd] + [(p.plan.MTW, p.plan.ncl) for l in eng.bwd for p in l]))
        torch.cuda.synchronize()
        assert outs[0][2] != outs[1][2], "the hint did not change any program"
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_window_form_of_the_dense_unpool_is_bitwise_the_slot_form():
    """vd_unpool_relu_bwd picks the one-thread-per-pool-window kernel for channels-last gradients on even conv grids with aligned
    pointers, and the one-thread-per-slot kernel otherwise (here: forced by a gradient pointer that is only 4-byte aligned).
    Same dense dy slots, both planes, bit for bit -- with and without temporal pooling, incl. ReLU-dead windows."""
    import ctypes
    from video_distillation_amd import hip
    L = hip.lib()
    g = torch.Generator().manual_seed(5)
    for (C, T, OH, OW, pool_t) in ((64, 4, 16, 20, 1), (128, 6, 8, 8, 2)):
        nb, To, Ho, Wo = 3, T // pool_t, OH // 2, OW // 2
        npos = To * Ho * Wo
        gp = torch.randn(nb * npos * C + 1, generator=g).cuda()
        am = torch.randint(0, 4 * pool_t, (nb * C * npos,), generator=g).to(torch.uint8)
        am[torch.rand(am.shape, generator=g) < 0.2] |= 0x80                      # ReLU-dead windows route nothing
        am = am.cuda()
        scale = torch.tensor([4.0], device="cuda")
        nslots = nb * (C // 8) * T * OH * OW
        outs = []
        for off in (0, 1):               # off = 1: the same values at a 4-byte-aligned address -> slot form
            src = gp[off:off + nb * npos * C]
            if off == 1:
                src.copy_(gp[:nb * npos * C].clone())
            hi = torch.zeros(nslots, 8, dtype=torch.int16, device="cuda")
            lo = torch.zeros_like(hi)
            hip.check(L.vd_unpool_relu_bwd(hip.ptr(src), hip.ptr(am), ctypes.c_int64(nb), C, To, Ho, Wo, pool_t, T, OH, OW, 1, hip.ptr(hi),
                                           hip.ptr(lo), hip.PREC["f16x3"], hip.ptr(scale), hip.stream_ptr()), "vd_unpool_relu_bwd")
            outs.append((hi, lo))
        torch.cuda.synchronize()
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        assert int((outs[0][0] != 0).sum()) > 0


def test_batched_operand_packing_is_bitwise_the_single_launches():
    """``engine.batched_packs`` (vd_pack_weights_multi: the operand packing of all programs of a step in one launch per 24 segments):
    the packed forward and input-gradient operands of an engine equal those of ``VD_PACK_BATCH=0`` (one launch per program)."""
    import os
    from video_distillation_amd import engine, plan
    geo = plan.NetGeometry(8, 64, 64)
    w = [p.cuda() for p in R.init_params(5, 3, 5)[:6]]

    def packed(batch):
        old = os.environ.get("VD_PACK_BATCH")
        os.environ["VD_PACK_BATCH"] = batch
        try:
            e = engine.EmbedEngine(geo, prec="f16x3", chunk=8)
            with engine.batched_packs():
                e.set_weights(w)
                e._pack_bwd()
                for rep in range(2):             # (more than 24 queued segments: two launches at the exit)
                    for li in range(3):
                        for dp in e.bwd[li]:
                            dp.pack(e._weights[2 * li])
            torch.cuda.synchronize()
            return [dp.wpk.clone() for dp in e.fwd + [d for layer in e.bwd for d in layer]]
        finally:
            if old is None:
                del os.environ["VD_PACK_BATCH"]
            else:
                os.environ["VD_PACK_BATCH"] = old
    a, b = packed("0"), packed("1")
    assert len(a) == len(b) >= 10 and all(torch.equal(x, y) for x, y in zip(a, b))
    assert any(int(x.abs().max()) > 0 for x in b)
