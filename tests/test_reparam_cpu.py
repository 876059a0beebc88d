"""ReparamModule surface (reparam_module.py:9-177) on CPU: flattening order, views, replacement vectors, buffers."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from video_distillation_amd import networks
from video_distillation_amd.reparam_module import ReparamModule


def test_flat_param_is_parameters_order_and_views_alias_it():
    torch.manual_seed(3)
    net = networks.ConvNet3D(3, 7, 128, 3, 'relu', 'none', 'maxpooling', 8, (64, 64))
    names = [n for n, _ in net.named_parameters()]
    ref = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    rp = ReparamModule(net)
    assert [n for n, _ in rp.named_parameters()] == ["flat_param"]
    assert rp.param_numel == ref.numel() and torch.equal(rp.flat_param.detach(), ref)
    assert rp._param_infos == tuple(("module." + n.rsplit(".", 1)[0], n.rsplit(".", 1)[1]) for n in names)
    tensors = net._all_params()                 # what the HIP path reads: attribute views of the flat vector
    assert [tuple(t.shape) for t in tensors] == [tuple(s) for s in rp._param_shapes]
    assert all(t.data_ptr() >= rp.flat_param.data_ptr() for t in tensors) and not isinstance(net.logit.weight, nn.Parameter)
    other = torch.randn(rp.param_numel)
    with rp.unflattened_param(other):
        assert torch.equal(net.logit.bias, other[-7:])
    assert torch.equal(net.logit.bias, rp.flat_param[-7:])
    with pytest.raises(RuntimeError):           # the wrapped ConvNet3D has no CPU path
        rp(torch.zeros(1, 8, 3, 64, 64), flat_param=other.unsqueeze(0))


def test_generic_module_forward_with_replacement_vector_and_buffers():
    torch.manual_seed(4)
    mod = nn.Sequential(nn.Linear(5, 4), nn.BatchNorm1d(4), nn.Linear(4, 2))
    mod[2].weight = mod[2].weight      # (no sharing here; exercised below)
    x = torch.randn(6, 5)
    want = mod(x).detach()
    rp = ReparamModule(mod)
    np.testing.assert_allclose(rp(x).detach().numpy(), want.numpy(), rtol=1e-6)
    flat = (rp.flat_param.detach() * 0.5).requires_grad_(True)
    out = rp(x, flat_param=flat.unsqueeze(0).expand(1, -1))
    (g,) = torch.autograd.grad(out.sum(), flat)
    assert g.shape == flat.shape and float(g.abs().sum()) > 0
    bufs = [b.clone() for _, _, b in rp._buffer_infos]
    rp.eval()
    out2 = rp(x, flat_param=flat, buffers=bufs)
    assert out2.shape == (6, 2)
    with pytest.raises(NotImplementedError):
        rp.trace(x)


def test_shared_parameter_keeps_one_slot():
    lin = nn.Linear(3, 3, bias=False)
    mod = nn.Sequential(lin, nn.ReLU(), lin)
    rp = ReparamModule(mod)
    assert rp.param_numel == 9
    flat = torch.eye(3).reshape(-1)
    x = torch.tensor([[1.0, -2.0, 3.0]])
    np.testing.assert_allclose(rp(x, flat_param=flat).detach().numpy(), [[1.0, 0.0, 3.0]])
