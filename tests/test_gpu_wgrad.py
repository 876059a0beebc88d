"""Weight gradient of the Conv3d layers as a tile program of the MFMA kernel (plan.plan_wgrad +
engine.WgradOp) against torch autograd (fp64 on the CPU)."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _slots_from_dense(a_bcthw, prec, planes):
    """channels-last 16-bit slots [planes][clip][C/8][T*H*W][8] of a (B,C,T,H,W) fp32 tensor (hi / lo)."""
    B, C, T, H, W = a_bcthw.shape
    cl = a_bcthw.reshape(B, C // 8, 8, T * H * W).permute(0, 1, 3, 2).contiguous()
    dt = torch.float16 if prec.startswith("f16") else torch.bfloat16
    hi = cl.to(dt)
    out = [hi.view(torch.int16)]
    if planes == 2:
        out.append((cl - hi.float()).to(dt).view(torch.int16))
    return torch.stack(out).contiguous()


@pytest.mark.parametrize("prec,tol", [("f16x3", 2e-5), ("f16", 2e-3), ("bf16x3", 1e-4)])
@pytest.mark.parametrize("cfg", [(64, 128, 8, 16, 16, 11), (128, 128, 4, 7, 7, 50), (3, 64, 8, 32, 32, 9)])
def test_wgrad_matches_autograd(prec, tol, cfg):
    from video_distillation_amd import engine
    cin, cout, t, h, w, n = cfg
    g = torch.Generator().manual_seed(cin + n)
    x = torch.randn(n, cin, t, h, w, generator=g)
    wgt = torch.zeros(cout, cin, 3, 7, 7, dtype=torch.double, requires_grad=True)
    y = F.conv3d(x.double(), wgt, None, stride=(1, 2, 2), padding=(1, 3, 3))
    dy = torch.randn(y.shape, generator=g)
    (want,) = torch.autograd.grad(y, wgt, dy.double())
    op = engine.WgradOp(cin, cout, t, h, w, n, prec, "cuda:0")
    dys = _slots_from_dense(dy, prec, op.planes).cuda()
    dw = torch.zeros(cout, cin, 3, 7, 7, device="cuda")
    if cin == 3:
        xs = x.permute(0, 2, 1, 3, 4).contiguous().cuda()      # (B,T,3,H,W) pixels
        op.run(xs, True, 0, dys, dys.shape[1] * dys.shape[2] * dys.shape[3] if False else int(dys[0].numel() // 8), dw)
    else:
        xs = _slots_from_dense(x, prec, op.planes).cuda()
        op.run(xs, False, int(xs[0].numel() // 8), dys, int(dys[0].numel() // 8), dw)
    torch.cuda.synchronize()
    err = float((dw.double().cpu() - want).norm() / want.norm())
    print(cfg, prec, "wgrad rel-l2 %.2e" % err)
    assert err < tol


@pytest.mark.parametrize("prec", ["f16x3", "f16", "bf16x3"])
@pytest.mark.parametrize("cfg", [(64, 8, 32, 32, 1, 11, 1, None), (128, 8, 16, 16, 2, 5, 0, None), (64, 4, 14, 14, 1, 9, 1, None),
                                 (128, 8, 14, 14, 2, 9, 1, "4,7,7"), (64, 6, 14, 14, 1, 3, 0, "4,4,14"), (128, 8, 8, 8, 2, 16, 1, "1,2,2")])
def test_fused_unpool_pack_is_bitwise_the_two_kernel_path(prec, cfg, monkeypatch):
    """vd_unpool_relu_bwd_packed == vd_unpool_relu_bwd followed by vd_pack_dy, bit for bit, for both gradient layouts,
    both pool depths, ragged clip counts (not a multiple of 8) and grids the block does not divide."""
    from video_distillation_amd import engine, hip
    cout, T, OH, OW, pt, n, layout, block = cfg
    if block:           # odd blocks take the one-slot-per-thread kernel, even ones the 2x2-window kernel; partial blocks at the grid edge
        monkeypatch.setenv("VD_WG_BLOCK", block)
    To, Ho, Wo = T // pt, OH // 2, OW // 2
    op = engine.WgradOp(64 if cout == 128 else 3, cout, T, 2 * OH, 2 * OW, n, prec, "cuda:0")
    assert (op.T, op.OH, op.OW) == (T, OH, OW)
    g = torch.Generator(device="cuda").manual_seed(cout + n)
    npos = To * Ho * Wo
    grad = torch.randn((n, cout, npos) if layout == 0 else (n, npos, cout), device="cuda", generator=g)
    am = torch.randint(0, 4 * pt, (n * cout * npos,), device="cuda", generator=g).to(torch.uint8)
    am[::13] = 128                                             # dead windows (ReLU): bit 7 set, never matches a position code
    scale = torch.tensor([4.0, 0.25], device="cuda")
    L, st = hip.lib(), hip.stream_ptr(op.device)
    nslots = n * (cout // 8) * T * OH * OW
    dy = torch.empty((op.planes, nslots, 8), dtype=torch.int16, device="cuda")
    lo = dy[1] if op.planes == 2 else None
    hip.check(L.vd_unpool_relu_bwd(hip.ptr(grad), hip.ptr(am), ctypes.c_int64(n), cout, To, Ho, Wo, pt, T, OH, OW, layout,
                                   hip.ptr(dy[0]), hip.ptr(lo), op.prec, hip.ptr(scale), st), "unpool")
    nt, noh, now = op.plan.meta["box"]
    op.bp.fill_(-1)
    hip.check(L.vd_pack_dy(hip.ptr(dy), ctypes.c_int64(nslots), op.planes, ctypes.c_int64(n), cout, T, OH, OW, nt, noh, now,
                           hip.ptr(op.bp), ctypes.c_int64(op.bp_elems), st), "pack")
    want = op.bp.clone()
    op.bp.fill_(-1)
    blo = op.bp[1] if op.planes == 2 else None
    hip.check(L.vd_unpool_relu_bwd_packed(hip.ptr(grad), hip.ptr(am), ctypes.c_int64(n), cout, To, Ho, Wo, pt, T, OH, OW, layout,
                                          nt, noh, now, hip.ptr(op.bp[0]), hip.ptr(blo), op.prec, hip.ptr(scale), st), "fused")
    torch.cuda.synchronize()
    assert torch.equal(op.bp, want)
    assert int((want != 0).sum()) > 0


@pytest.mark.parametrize("layer,n,prec", [(1, 11, "f16x3"), (0, 9, "f16"), (2, 50, "bf16x3")])
def test_weight_gradient_through_the_c_abi_alone(layer, n, prec):
    """vd_program_build_wgrad (C++ planner) -> vd_program_load -> vd_clip_minor_* / vd_pack_dy -> vd_program_run_wgrad ->
    vd_replica_sum: the weight gradient of a layer without the Python planner or engine, against engine.WgradOp (same
    program, so equal to the atomics' summation order) and against autograd."""
    from video_distillation_amd import engine, hip, plan
    geo = plan.NetGeometry(8, 64, 64)
    cin, cout, t, h, w = geo.layer_dims()[layer][:5]
    g = torch.Generator().manual_seed(layer + n)
    x = torch.randn(n, cin, t, h, w, generator=g)
    wgt = torch.zeros(cout, cin, 3, 7, 7, dtype=torch.double, requires_grad=True)
    y = F.conv3d(x.double(), wgt, None, stride=(1, 2, 2), padding=(1, 3, 3))
    dy = torch.randn(y.shape, generator=g)
    (want,) = torch.autograd.grad(y, wgt, dy.double())
    planes = 2 if prec.endswith("x3") else 1
    L, st, dev = hip.lib(), hip.stream_ptr(torch.device("cuda:0")), "cuda:0"
    blob, nb, block, reps = ctypes.c_void_p(), ctypes.c_int64(), (ctypes.c_int * 3)(), ctypes.c_int()
    hip.check(L.vd_program_build_wgrad(layer, 8, 64, 64, n, planes, ctypes.byref(blob), ctypes.byref(nb), block, ctypes.byref(reps)), "build")
    prog = ctypes.c_void_p()
    hip.check(L.vd_program_load(blob, nb, hip.PREC[prec], ctypes.byref(prog)), "load")
    L.vd_blob_free(blob)
    try:
        T, OH, OW = y.shape[2:]
        CCb, npos = (n + 7) // 8, t * h * w
        xT = torch.empty((planes, cin * CCb * npos, 8), dtype=torch.int16, device=dev)
        if layer == 0:
            xs = x.permute(0, 2, 1, 3, 4).contiguous().cuda()
            hip.check(L.vd_clip_minor_pix(hip.ptr(xs), ctypes.c_int64(n), t, h, w, hip.ptr(xT[0]), hip.ptr(xT[1] if planes == 2 else None),
                                          hip.PREC[prec], st), "clip_minor_pix")
        else:
            xs = _slots_from_dense(x, prec, planes).cuda()
            hip.check(L.vd_clip_minor_cl(hip.ptr(xs), ctypes.c_int64(xs[0].numel() // 8), planes, ctypes.c_int64(n), cin, ctypes.c_int64(npos),
                                         hip.ptr(xT), ctypes.c_int64(xT.shape[1]), st), "clip_minor_cl")
        dys = _slots_from_dense(dy, prec, planes).cuda()
        nt, noh, now = block
        S = nt * noh * now // 2
        nbox = -(-T // nt) * -(-OH // noh) * -(-OW // now)
        bp_elems = nbox * CCb * S * (cout // 32) * 64 * 8
        bp = torch.empty((planes, bp_elems), dtype=torch.int16, device=dev)
        hip.check(L.vd_pack_dy(hip.ptr(dys), ctypes.c_int64(dys[0].numel() // 8), planes, ctypes.c_int64(n), cout, T, OH, OW, nt, noh, now,
                               hip.ptr(bp), ctypes.c_int64(bp_elems), st), "pack_dy")
        copies = torch.zeros((reps.value, cin * 147, cout), dtype=torch.float32, device=dev)
        hip.check(L.vd_program_run_wgrad(prog, hip.ptr(xT), ctypes.c_int64(xT.shape[1]), hip.ptr(bp), ctypes.c_int64(bp_elems), hip.ptr(copies),
                                         ctypes.c_int64(cin * 147 * cout), cin, None, st), "run_wgrad")
        dw = torch.zeros(cout, cin, 3, 7, 7, device=dev)
        hip.check(L.vd_replica_sum(hip.ptr(copies), reps.value, cin * 147, cout, hip.ptr(dw), st), "replica_sum")
        torch.cuda.synchronize()
        assert L.vd_program_run_wgrad(prog, hip.ptr(xT), ctypes.c_int64(xT.shape[1]), hip.ptr(bp), ctypes.c_int64(bp_elems), hip.ptr(copies),
                                      ctypes.c_int64(7), cin, None, st) == -2          # wrong size of the accumulation copies
    finally:
        L.vd_program_free(prog)
    op = engine.WgradOp(cin, cout, t, h, w, n, prec, dev)
    assert tuple(op.plan.meta["box"]) == tuple(block) and op.replicas == reps.value
    ref = torch.zeros_like(dw)
    if layer == 0:
        op.run(xs, True, 0, dys, int(dys[0].numel() // 8), ref)
    else:
        op.run(xs, False, int(xs[0].numel() // 8), dys, int(dys[0].numel() // 8), ref)
    torch.cuda.synchronize()
    same = float((dw - ref).norm() / ref.norm())
    err = float((dw.double().cpu() - want).norm() / want.norm())
    print("C-ABI wgrad layer %d %s: vs engine %.1e, vs autograd %.1e" % (layer, prec, same, err))
    assert same < 1e-6 and err < (2e-3 if prec == "f16" else 1e-4)
