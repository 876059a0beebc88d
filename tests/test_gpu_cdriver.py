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
    np.concatenate([p.numpy().reshape(-1) for p in params]).astype(np.float32).tofile(os.path.join(d, "weights.bin"))
    exe = os.path.join(d, "dm_class_term")
    libdir = os.path.dirname(hip.LIB_PATH)
    subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O2", "-o", exe, os.path.join(ROOT, "examples", "dm_class_term.cpp"),
                    "-L" + libdir, "-lvd_hip", "-Wl,-rpath," + libdir], check=True)
    # a class mean over FOUR single-pass real clips, one synthetic clip (the 64-clip bar is tests/test_gpu_parity_late.py's 1e-3).  Three
    # draws of the clips: every sample within round 4's 2e-3 (round 5 had loosened it to 3.5e-3 for one sample), their MEAN within 1.5e-3
    # -- an accumulation-order change moves single samples, a regression moves the mean.  Measured in round 6: 1.07 / 1.08 / 1.02e-3
    rels = []
    for seed in (32, 33, 34):
        g = torch.Generator().manual_seed(seed)
        real = torch.randn(NR, T, 3, H, W, generator=g)
        syn = torch.randn(1, T, 3, H, W, generator=g)
        real.numpy().astype(np.float32).tofile(os.path.join(d, "real.bin"))
        syn.numpy().astype(np.float32).tofile(os.path.join(d, "syn.bin"))
        out = subprocess.run([exe, d, str(NR), str(T), str(H), str(W)], check=True, capture_output=True, text=True)
        print(out.stdout.strip())
        res = np.fromfile(os.path.join(d, "dm_out.bin"), dtype=np.float32)
        n = syn.numel()
        loss, grad, syn_after = float(res[0]), torch.from_numpy(res[1:1 + n]).view_as(syn), torch.from_numpy(res[1 + n:]).view_as(syn)
        loss_ref, grad_ref = R.dm_loss_and_grad(params, [real], syn, ipc=1)
        rel_l = abs(loss - float(loss_ref)) / float(loss_ref)
        rel_g = float((grad - grad_ref).norm() / grad_ref.norm())
        print("C DM class term vs oracle (clips drawn from seed %d): loss rel %.2e, gradient rel-l2 %.2e" % (seed, rel_l, rel_g))
        assert rel_l < 1e-3 and rel_g < 2e-3
        rels.append(rel_g)
        want_after, _ = R.sgd_momentum_step(syn, grad, None, 0.5, 0.5)                  # first step: buf = g
        np.testing.assert_allclose(syn_after.numpy(), want_after.numpy(), rtol=1e-5, atol=1e-6)
    assert sum(rels) / len(rels) < 1.5e-3, rels


@pytest.mark.parametrize("geom,B,K,prec,prec_bwd", [((8, 64, 64), 6, 5, "f16x3", "f16x3"), ((8, 64, 64), 9, 4, "f16x3", "f16"),
                                                   ((16, 112, 112), 4, 3, "bf16x3", "bf16x3")])
def test_training_step_through_the_c_handle(geom, B, K, prec, prec_bwd):
    """vd_train_create / vd_train_step (planners, programs, head, loss, gradients, SGD -- all behind the C ABI) against
    ConvNet3D.hip_train_step for two consecutive steps (momentum, weight decay): per-clip losses, logits and the 8 updated
    parameter tensors."""
    import ctypes
    import torch
    from video_distillation_amd import hip, networks
    T, H, W = geom
    g = torch.Generator().manual_seed(B * 10 + K)
    x = torch.randn(B, T, 3, H, W, generator=g).cuda()
    y = torch.randint(0, K, (B,), generator=g).cuda()
    lr, mom, wd = 0.05, 0.9, 5e-4
    torch.manual_seed(5)
    net = networks.ConvNet3D(3, K, 128, 3, 'relu', 'none', 'maxpooling', T, (H, W)).cuda().train()
    net.dropout.p = 0.0
    p0 = [p.detach().clone() for p in net.parameters()]
    old = networks.get_precision()
    networks.set_precision(train=prec, train_bwd=prec_bwd)
    try:
        opt = torch.optim.SGD(net.parameters(), lr=lr, momentum=mom, weight_decay=wd)
        crit = torch.nn.CrossEntropyLoss().cuda()
        assert net.hip_trainable(x, opt, crit)
        ref = [net.hip_train_step(x, y, opt) for _ in range(2)]
    finally:
        networks.set_precision(train=old["train"], train_bwd=old["train_bwd"])
    L, st = hip.lib(), hip.stream_ptr(torch.device("cuda:0"))
    tr = ctypes.c_void_p()
    hip.check(L.vd_train_create(T, H, W, K, hip.PREC[prec], hip.PREC[prec_bwd], ctypes.c_int64(B), ctypes.byref(tr)), "vd_train_create")
    try:
        nbytes = L.vd_train_workspace_bytes(tr)
        assert nbytes > 0
        # an UNALIGNED workspace pointer of exactly the queried size, with a canary behind it: the handle rounds the pointer up
        # itself and must stay inside [ptr, ptr + nbytes)
        ws_all = torch.full((nbytes + 200 + 4096,), 0xA5, dtype=torch.uint8, device="cuda")
        ws = ws_all[200:]
        params = [p.clone().contiguous() for p in p0]
        bufs = [torch.zeros_like(p) for p in params]
        P8 = (ctypes.c_void_p * 8)(*[p.data_ptr() for p in params])
        M8 = (ctypes.c_void_p * 8)(*[b.data_ptr() for b in bufs])
        loss_c = torch.empty(B, device="cuda")
        logits = torch.empty(B, K, device="cuda")
        got = []
        for step in range(2):
            hip.check(L.vd_train_step(tr, P8, M8, hip.ptr(x), hip.ptr(y), None, ctypes.c_float(lr), ctypes.c_float(mom), ctypes.c_float(wd),
                                      int(step == 0), hip.ptr(ws), ctypes.c_int64(nbytes), hip.ptr(loss_c), hip.ptr(logits), st), "vd_train_step")
            torch.cuda.synchronize()
            assert bool((ws_all[200 + nbytes:] == 0xA5).all()) and bool((ws_all[:200] == 0xA5).all()), "vd_train_step wrote outside its workspace"
            got.append((logits.clone(), float(loss_c.mean())))
        assert L.vd_train_step(tr, P8, M8, hip.ptr(x), hip.ptr(y), None, ctypes.c_float(lr), ctypes.c_float(mom), ctypes.c_float(wd), 0,
                               hip.ptr(ws), ctypes.c_int64(1024), None, None, st) == -7
    finally:
        L.vd_train_free(tr)
    for step in range(2):
        # step 0 starts from identical weights: deterministic.  Step 1 starts from weights that carry the summation order of
        # fp32 atomics on both sides, so a pooling near-tie may resolve differently (~1e-3 on everything downstream): flip-tolerant
        ref_logits, ref_loss = ref[step]
        tol = (2e-5, 2e-4) if step == 0 else (5e-3, 2e-2)
        assert abs(got[step][1] - float(ref_loss)) <= tol[0] * abs(float(ref_loss)) + 1e-6, (step, got[step][1], float(ref_loss))
        assert float((got[step][0] - ref_logits).abs().max()) <= tol[1] * float(ref_logits.abs().max()) + 1e-6
    errs = [float((a - b.detach()).norm() / (b.detach() - q).norm().clamp_min(1e-30)) for a, b, q in zip(params, net.parameters(), p0)]
    print("C train step %s/%s: losses %s, per-tensor error of the two-step update %s" % (prec, prec_bwd, [v for _, v in got], ["%.1e" % v for v in errs]))
    assert float(np.median(errs)) < 1e-3 and max(errs) < 5e-2          # typical 1e-6 .. 1e-5; flip tolerant (see above)
    tr2 = ctypes.c_void_p()
    assert L.vd_train_create(T, H, W, K, hip.PREC["f16"], hip.PREC["f16x3"], ctypes.c_int64(B), ctypes.byref(tr2)) == -2
    assert L.vd_train_create(T, H, W, K, hip.PREC["f16x3"], hip.PREC["bf16x3"], ctypes.c_int64(B), ctypes.byref(tr2)) == -2
