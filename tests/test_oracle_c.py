"""The plain-C oracle (oracle/conv3d_ref.c, fp64 accumulation) against the torch oracle and the
reference's golden layer outputs (fixture G1) -- an arbitration reference that does not depend
on MKL-DNN's summation order."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ref_cpu as R

ODIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")


@pytest.fixture(scope="module")
def clib():
    subprocess.run(["make", "-s", "-C", ODIR], check=True)
    return ctypes.CDLL(os.path.join(ODIR, "libvdref.so"))


def fp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def test_c_layer_matches_torch_and_golden(clib, golden_dir):
    z = np.load(os.path.join(golden_dir, "g1_layers.npz"))
    params = R.init_params(int(z["seed"]))
    g = torch.Generator().manual_seed(int(z["x_seed"]))
    x = torch.randn(2, 8, 3, 64, 64, generator=g)
    xc = np.ascontiguousarray(x[0].permute(1, 0, 2, 3).numpy())          # (C,T,H,W) of clip 0
    w, b = params[0].numpy().copy(), params[1].numpy().copy()
    y = np.zeros((64, 8, 32, 32), dtype=np.float32)
    clib.vdref_conv3d(fp(xc), fp(w), fp(b), 3, 8, 64, 64, 64, fp(y))
    np.testing.assert_allclose(y[::8, :, ::4, ::4], z["conv0"][0], rtol=1e-4, atol=1e-5)   # reference's own output
    p = np.zeros((64, 8, 16, 16), dtype=np.float32); arg = np.zeros(p.shape, dtype=np.uint8)
    clib.vdref_relu_maxpool(fp(y), 64, 8, 32, 32, 1, fp(p), fp(arg))
    np.testing.assert_allclose(p[::8, :, ::2, ::2], z["pool0"][0], rtol=1e-4, atol=1e-5)
    want = F.max_pool3d(torch.relu(torch.tensor(y)[None]), (1, 2, 2))[0].numpy()
    np.testing.assert_array_equal(p, want)
    assert arg.max() <= 3


def test_c_pool222_argmax_and_input_gradient(clib):
    g = torch.Generator().manual_seed(3)
    C, N, T, H, W = 8, 16, 4, 10, 12
    x = torch.randn(C, T, H, W, generator=g); w = torch.randn(N, C, 3, 7, 7, generator=g) * 0.1
    y = np.zeros((N, 4, 5, 6), dtype=np.float32)
    clib.vdref_conv3d(fp(x.numpy()), fp(w.numpy()), None, C, T, H, W, N, fp(y))
    xt = x[None].clone().requires_grad_(True)
    yt = F.conv3d(xt, w, None, stride=(1, 2, 2), padding=(1, 3, 3))
    np.testing.assert_allclose(y, yt[0].detach().numpy(), rtol=1e-4, atol=1e-5)
    p = np.zeros((N, 2, 2, 3), dtype=np.float32); arg = np.zeros(p.shape, dtype=np.uint8)
    clib.vdref_relu_maxpool(fp(y), N, 4, 5, 6, 2, fp(p), fp(arg))      # floor pooling drops the odd row
    pt, idx = F.max_pool3d(torch.relu(torch.tensor(y)[None]), 2, return_indices=True)   # same conv grid: exact
    np.testing.assert_array_equal(p, pt[0].numpy())
    flat = idx[0].numpy()
    dt, dh, dw = (flat // 30) % 2, ((flat % 30) // 6) % 2, (flat % 6) % 2
    np.testing.assert_array_equal(arg[p > 0], (dt * 4 + dh * 2 + dw)[p > 0])
    dy = torch.randn(1, N, 4, 5, 6, generator=g)
    (dx_t,) = torch.autograd.grad(yt, xt, dy)
    dx = np.zeros((C, T, H, W), dtype=np.float32)
    clib.vdref_conv3d_bwd_data(fp(dy[0].numpy().copy()), fp(w.numpy()), C, T, H, W, N, fp(dx))
    np.testing.assert_allclose(dx, dx_t[0].numpy(), rtol=1e-4, atol=1e-5)
