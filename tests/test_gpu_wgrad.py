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
