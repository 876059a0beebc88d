"""The hot path driven WITHOUT Python or torch in the loop: examples/embed_forward.cpp is compiled against
include/vd_hip.h + libvd_hip.so, loads the serialised tile programs (vd_program_load) and runs
ConvNet3D.embed through the C ABI; its output must equal the engine's (same kernels) and match the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("prec,x3,tol", [("f16", False, 2e-3), ("f16x3", True, 3e-5)])
def test_c_driver_runs_embed_through_the_c_abi(tmp_path, prec, x3, tol):
    from video_distillation_amd import engine, hip, plan
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import export_programs
    T, H, W, B = 8, 64, 64, 5
    d = str(tmp_path)
    export_programs.export(d, T, H, W, x3=x3)
    params = R.init_params(21)[:6]
    g = torch.Generator().manual_seed(22)
    x = torch.randn(B, T, 3, H, W, generator=g)
    np.concatenate([p.numpy().reshape(-1) for p in params]).astype(np.float32).tofile(os.path.join(d, "weights.bin"))
    x.numpy().astype(np.float32).tofile(os.path.join(d, "clips.bin"))
    exe = os.path.join(d, "embed_forward")
    libdir = os.path.dirname(hip.LIB_PATH)
    subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O2", "-o", exe, os.path.join(ROOT, "examples", "embed_forward.cpp"),
                    "-L" + libdir, "-lvd_hip", "-Wl,-rpath," + libdir], check=True)
    out = subprocess.run([exe, d, str(B), str(T), str(H), str(W), str(hip.PREC[prec])], check=True, capture_output=True, text=True)
    print(out.stdout.strip())
    feats = torch.from_numpy(np.fromfile(os.path.join(d, "feats.bin"), dtype=np.float32).reshape(B, -1))
    want = R.convnet3d_embed(x, params)
    rel = float((feats - want).norm() / want.norm())
    print("C driver vs oracle rel-l2 %.2e (%s)" % (rel, prec))
    assert feats.shape == want.shape and rel < tol
    eng = engine.EmbedEngine(plan.NetGeometry(T, H, W), prec=prec, chunk=B)
    eng.set_weights([p.cuda() for p in params])
    assert torch.equal(eng.forward(x.cuda()).cpu(), feats)        # same programs, same kernels: bitwise


@pytest.mark.parametrize("prec,tol", [("f16", 2e-3), ("f16x3", 3e-5)])
def test_c_driver_with_the_library_s_own_planner(tmp_path, prec, tol):
    """examples/embed_standalone.cpp: vd_embed_create plans the programs in C++ (no Python-made blob), workspace size from
    vd_embed_workspace_bytes; features bitwise equal to the Python engine's (whose programs come from plan.py)."""
    from video_distillation_amd import engine, hip, plan
    T, H, W, B = 8, 64, 64, 5
    d = str(tmp_path)
    params = R.init_params(23)[:6]
    g = torch.Generator().manual_seed(24)
    x = torch.randn(B, T, 3, H, W, generator=g)
    np.concatenate([p.numpy().reshape(-1) for p in params]).astype(np.float32).tofile(os.path.join(d, "weights.bin"))
    x.numpy().astype(np.float32).tofile(os.path.join(d, "clips.bin"))
    exe = os.path.join(d, "embed_standalone")
    libdir = os.path.dirname(hip.LIB_PATH)
    subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O2", "-o", exe, os.path.join(ROOT, "examples", "embed_standalone.cpp"),
                    "-L" + libdir, "-lvd_hip", "-Wl,-rpath," + libdir], check=True)
    out = subprocess.run([exe, d, str(B), str(T), str(H), str(W), str(hip.PREC[prec])], check=True, capture_output=True, text=True)
    print(out.stdout.strip())
    feats = torch.from_numpy(np.fromfile(os.path.join(d, "feats_standalone.bin"), dtype=np.float32).reshape(B, -1))
    want = R.convnet3d_embed(x, params)
    rel = float((feats - want).norm() / want.norm())
    print("standalone C driver vs oracle rel-l2 %.2e (%s)" % (rel, prec))
    assert feats.shape == want.shape and rel < tol
    eng = engine.EmbedEngine(plan.NetGeometry(T, H, W), prec=prec, chunk=B, batch_hint=B)
    eng.set_weights([p.cuda() for p in params])
    assert torch.equal(eng.forward(x.cuda()).cpu(), feats)


def test_c_driver_dm_class_term_forward_backward_sgd(tmp_path):
    """examples/dm_class_term.cpp: one DM class term entirely through the C ABI (library-planned forward and
    input-gradient programs, vd_dm_loss, vd_embed_backward, vd_sgd_momentum) against the oracle."""
    from video_distillation_amd import hip
    T, H, W, NR = 8, 64, 64, 4
    d = str(tmp_path)
    params = R.init_params(31)[:6]
    g = torch.Generator().manual_seed(32)
    real = torch.randn(NR, T, 3, H, W, generator=g)
    syn = torch.randn(1, T, 3, H, W, generator=g)
    np.concatenate([p.numpy().reshape(-1) for p in params]).astype(np.float32).tofile(os.path.join(d, "weights.bin"))
    real.numpy().astype(np.float32).tofile(os.path.join(d, "real.bin"))
    syn.numpy().astype(np.float32).tofile(os.path.join(d, "syn.bin"))
    exe = os.path.join(d, "dm_class_term")
    libdir = os.path.dirname(hip.LIB_PATH)
    subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O2", "-o", exe, os.path.join(ROOT, "examples", "dm_class_term.cpp"),
                    "-L" + libdir, "-lvd_hip", "-Wl,-rpath," + libdir], check=True)
    out = subprocess.run([exe, d, str(NR), str(T), str(H), str(W)], check=True, capture_output=True, text=True)
    print(out.stdout.strip())
    res = np.fromfile(os.path.join(d, "dm_out.bin"), dtype=np.float32)
    n = syn.numel()
    loss, grad, syn_after = float(res[0]), torch.from_numpy(res[1:1 + n]).view_as(syn), torch.from_numpy(res[1 + n:]).view_as(syn)
    loss_ref, grad_ref = R.dm_loss_and_grad(params, [real], syn, ipc=1)
    rel_l = abs(loss - float(loss_ref)) / float(loss_ref)
    rel_g = float((grad - grad_ref).norm() / grad_ref.norm())
    print("C DM class term vs oracle: loss rel %.2e, gradient rel-l2 %.2e" % (rel_l, rel_g))
    assert rel_l < 1e-3 and rel_g < 2e-3
    want_after, _ = R.sgd_momentum_step(syn, grad, None, 0.5, 0.5)                  # first step: buf = g
    np.testing.assert_allclose(syn_after.numpy(), want_after.numpy(), rtol=1e-5, atol=1e-6)
