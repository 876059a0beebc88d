"""world_size-2 ``gloo`` runs of the DM / s2d trainers' host logic (class sharding, collectives)
on the CPU with the oracle as compute backend; results must equal the 1-rank run and the
golden fixtures G3 / G5 (sharding identity)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ref_cpu as R
from tests.cpu_backend import OracleBackend
from video_distillation_amd import distill

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def randn(seed, *shapes):
    g = torch.Generator().manual_seed(int(seed))
    return [torch.randn(*s, generator=g) for s in shapes]


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


class _Pool:
    pass


def _g3_setup(rank, world, shard="class", exchange="owner"):
    z = np.load(os.path.join(GOLDEN, "g3_dm_steps.npz"))
    (syn,) = randn(z["syn_seed"], (3, 8, 3, 64, 64))
    reals = [randn(z["real_seeds"][it], *[(4, 8, 3, 64, 64)] * 3) for it in range(2)]
    pool = _Pool(); pool.clips = torch.cat([torch.cat(r) for r in reals]); pool.counts = [4] * 3; pool.offsets = [0, 4, 8]
    lo, hi = distill.class_range(3, rank, world)
    own = list(range(lo, hi))
    if shard == "hybrid" and world > 1:
        block, _, mine = distill.hybrid_partition(3, rank, world)
        own = block + mine
    tr = distill.DMTrainer(OracleBackend(net_seeds=z["net_seeds"]), pool, 3, 1, 4, lr_img=float(z["lr"]),
                           momentum=float(z["momentum"]), rank=rank, world=world, image_syn=syn[own].clone(), shard=shard,
                           exchange=exchange)
    return z, tr


def _g3_run(rank, world, shard="class", exchange="owner"):
    z, tr = _g3_setup(rank, world, shard, exchange)
    orig = distill.sample_real_indices
    losses = []
    try:
        for it in range(2):
            distill.sample_real_indices = lambda it_, counts, offsets, b, classes, it=it: np.concatenate(
                [it * 12 + offsets[c] + np.arange(4) for c in classes]).astype(np.int64)
            losses.append(float(tr.global_loss(tr.step(it))))
    finally:
        distill.sample_real_indices = orig
    return z, losses, tr.gather_syn()


def _worker_g3(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        z, losses, syn = _g3_run(rank, world)
        if rank == 0:
            q.put((losses, syn.numpy()))
    finally:
        dist.destroy_process_group()


def _worker_g3_batch(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        z, losses, syn = _g3_run(rank, world, shard="batch")
        if rank == 0:
            q.put((losses, syn.numpy()))
    finally:
        dist.destroy_process_group()


def _worker_g3_hybrid(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        z, losses, syn = _g3_run(rank, world, shard="hybrid")
        if rank == 0:
            q.put((losses, syn.numpy()))
    finally:
        dist.destroy_process_group()


def _worker_g3_allreduce(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        before = dict(distill.COLLECTIVE_CALLS)
        z, losses, syn = _g3_run(rank, world, shard="hybrid", exchange="allreduce")
        calls = distill.COLLECTIVE_CALLS["all_reduce"] - before["all_reduce"]
        if rank == 0:
            q.put((losses, syn.numpy(), calls))
    finally:
        dist.destroy_process_group()


def _spawn(fn, world):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=fn, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return out


def test_class_range_partitions():
    for C, W in ((50, 8), (51, 8), (400, 8), (3, 2), (5, 8)):
        spans = [distill.class_range(C, r, W) for r in range(W)]
        assert spans[0][0] == 0 and spans[-1][1] == C
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1
    assert [b - a for a, b in [distill.class_range(50, r, 8) for r in range(8)]] == [7, 7, 6, 6, 6, 6, 6, 6]


def test_real_batch_sampling_is_sharding_invariant():
    counts, offsets = [93] * 6, [0, 93, 186, 279, 372, 465]
    full = distill.sample_real_indices(7, counts, offsets, 64, range(6))
    parts = np.concatenate([distill.sample_real_indices(7, counts, offsets, 64, range(a, b)) for a, b in ((0, 2), (2, 6))])
    np.testing.assert_array_equal(full, parts)
    assert len(set(full[:64])) == 64 and full[:64].max() < 93          # a permutation prefix of class 0
    assert distill.sample_real_indices(8, counts, offsets, 64, range(6)).tolist() != full.tolist()
    few = distill.sample_real_indices(1, [3], [0], 8, [0], allow_repeat=True)     # fewer clips than batch_real: opt-in only
    assert few.shape == (8,) and few.max() < 3
    assert distill.sample_real_indices(1, counts, offsets, 64, []).shape == (0,)


def test_dm_trainer_single_rank_matches_golden():
    z, losses, syn = _g3_run(0, 1)
    np.testing.assert_allclose(losses, z["losses"], rtol=1e-5)
    np.testing.assert_allclose(syn[:, ::2, :, ::4, ::4].numpy(), z["syn2"], rtol=1e-4, atol=1e-5)


def test_dm_trainer_two_ranks_gloo_matches_golden():
    losses, syn = _spawn(_worker_g3, 2)
    z = np.load(os.path.join(GOLDEN, "g3_dm_steps.npz"))
    np.testing.assert_allclose(losses, z["losses"], rtol=1e-5)       # all-reduced loss == 1-rank loss
    np.testing.assert_allclose(syn[:, ::2, :, ::4, ::4], z["syn2"], rtol=1e-4, atol=1e-5)   # all-gathered clips


def test_dm_trainer_two_ranks_batch_sharded_matches_golden():
    """Real batch split over ranks + all-reduce of per-class feature sums == the reference step."""
    losses, syn = _spawn(_worker_g3_batch, 2)
    z = np.load(os.path.join(GOLDEN, "g3_dm_steps.npz"))
    np.testing.assert_allclose(losses, z["losses"], rtol=1e-5)
    np.testing.assert_allclose(syn[:, ::2, :, ::4, ::4], z["syn2"], rtol=1e-4, atol=1e-5)


def test_dm_trainer_two_ranks_hybrid_matches_golden():
    """3 classes on 2 ranks, hybrid: one whole class per rank + the third class's real batch split in halves (all-reduce of
    its 256 feature sums), its synthetic clip owned by rank 0, the gathered clips back in class order == the reference step."""
    assert distill.hybrid_partition(3, 0, 2) == ([0], [2], [2]) and distill.hybrid_partition(3, 1, 2) == ([1], [2], [])
    losses, syn = _spawn(_worker_g3_hybrid, 2)
    z = np.load(os.path.join(GOLDEN, "g3_dm_steps.npz"))
    np.testing.assert_allclose(losses, z["losses"], rtol=1e-5)
    np.testing.assert_allclose(syn[:, ::2, :, ::4, ::4], z["syn2"], rtol=1e-4, atol=1e-5)


def test_dm_trainer_two_ranks_pixel_gradient_allreduce_matches_golden():
    """``exchange='allreduce'``: the literal "all-reduce of the matching-loss gradient" -- the full (C * ipc, ...) pixel-gradient
    tensor, each rank's rows scattered into zeros, summed over the ranks, the owner updating from the reduced tensor (the
    tensor distill_baseline.py:353-355 steps on) -- gives the same two steps as owner-computes (hybrid ownership: rank 0 owns
    classes 0 and 2, rank 1 class 1, so the scatter rows are not contiguous)."""
    losses, syn, calls = _spawn(_worker_g3_allreduce, 2)
    z = np.load(os.path.join(GOLDEN, "g3_dm_steps.npz"))
    np.testing.assert_allclose(losses, z["losses"], rtol=1e-5)
    np.testing.assert_allclose(syn[:, ::2, :, ::4, ::4], z["syn2"], rtol=1e-4, atol=1e-5)
    assert calls == 2 * 3          # per step: split-class feature sums, pixel gradients, loss


def _g5_run(rank, world):
    z = np.load(os.path.join(GOLDEN, "g5_s2d_step.npz"))
    C, vpc, spc, dpc = 3, 1, 2, 2
    static_syn, dynamic_syn = randn(z["data_seed"], (C * spc, 3, 64, 64), (C, dpc, 8, 1, 64, 64))
    reals = randn(z["real_seed"], *[(4, 8, 3, 64, 64)] * C)
    pool = _Pool(); pool.clips = torch.cat(reals); pool.counts = [4] * C; pool.offsets = [0, 4, 8]
    tr = distill.S2DTrainer(OracleBackend(net_seeds={0: int(z["net_seed"])}), pool, C, vpc, spc, dpc, 4, static_syn,
                            dynamic_syn, torch.tensor(z["hal_w"]), torch.tensor(z["hal_b"]), lr_dynamic=10.0,
                            lr_hal=0.01, rank=rank, world=world)
    orig = distill.sample_real_indices
    try:
        distill.sample_real_indices = lambda it_, counts, offsets, b, classes: np.concatenate(
            [offsets[c] + np.arange(4) for c in classes]).astype(np.int64)
        loss = tr.step(0, draws=(z["draws_dyn"], z["draws_sta"]))
    finally:
        distill.sample_real_indices = orig
    if world > 1:
        loss = loss.clone(); dist.all_reduce(loss)
    return z, float(loss), tr


def _worker_g5(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        z, loss, tr = _g5_run(rank, world)
        if rank == 0:
            q.put((loss, tr.hal_w.numpy(), tr.dynamic.numpy()))
    finally:
        dist.destroy_process_group()


def test_s2d_trainer_single_rank_matches_golden():
    z, loss, tr = _g5_run(0, 1)
    assert abs(loss / float(z["loss"]) - 1) < 1e-5
    np.testing.assert_allclose(tr.hal_w.numpy(), z["hal_w_after"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(tr.dynamic.view(3, 2, 8, 1, 64, 64)[:, :, :, :, ::4, ::4].numpy(), z["dynamic_after"], rtol=1e-4, atol=1e-5)
    s, d = tr.indices(0, draws=(z["draws_dyn"], z["draws_sta"]))
    np.testing.assert_array_equal(s, z["static_idx"])
    np.testing.assert_array_equal(d - 2 * np.arange(3), z["dynamic_idx"])      # flattened (class, dpc) rows


def test_s2d_trainer_two_ranks_allreduce_hallucinator_grad():
    loss, hal_w, dyn0 = _spawn(_worker_g5, 2)
    z = np.load(os.path.join(GOLDEN, "g5_s2d_step.npz"))
    assert abs(loss / float(z["loss"]) - 1) < 1e-5
    np.testing.assert_allclose(hal_w, z["hal_w_after"], rtol=1e-5, atol=1e-7)   # shared params: grads summed over ranks
    np.testing.assert_allclose(dyn0.reshape(2, 2, 8, 1, 64, 64)[:, :, :, :, ::4, ::4], z["dynamic_after"][:2], rtol=1e-4, atol=1e-5)


# ---- gradient matching (DC) trainer: class sharding identity ------------------------------------

def _gm_run(rank, world, outer_loop=1, inner_loop=1, steps=2):
    from tests.cpu_backend import OracleGMOps
    from video_distillation_amd import plan
    C, ipc = 2, 1
    geo = plan.NetGeometry(8, 64, 64)
    g = torch.Generator().manual_seed(4242)
    clips = torch.randn(C * 3, 8, 3, 64, 64, generator=g)
    pool = distill.RealPool(clips, [3] * C, [0, 3])
    lo, hi = distill.class_range(C, rank, world)
    syn = torch.stack([clips[0], clips[3]])[lo:hi].clone()
    tr = distill.GMTrainer(OracleGMOps("ours"), pool, geo, C, ipc, batch_real=2, lr_img=0.1, lr_net=0.01, rank=rank,
                           world=world, image_syn=syn, outer_loop=outer_loop, inner_loop=inner_loop, dropout_p=0.0)
    losses = [float(tr.global_loss(tr.step(it))) for it in range(steps)]
    return losses, tr.gather_syn()


def _worker_gm(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        losses, syn = _gm_run(rank, world, outer_loop=2, inner_loop=1, steps=1)
        if rank == 0:
            q.put((losses, syn.numpy()))
    finally:
        dist.destroy_process_group()


def test_gm_trainer_class_sharding_identity_two_ranks_gloo():
    """Two ranks owning one class each (outer loop 2: one all-gather + network update in between)
    reproduce the single-rank run: summed loss and gathered synthetic clips."""
    torch.set_num_threads(4)
    want_l, want_s = _gm_run(0, 1, outer_loop=2, inner_loop=1, steps=1)
    got_l, got_s = _spawn(_worker_gm, 2)
    np.testing.assert_allclose(got_l, want_l, rtol=1e-5)
    np.testing.assert_allclose(got_s, want_s.numpy(), rtol=1e-4, atol=5e-5)   # thread-count dependent summation order
    assert want_l[0] > 0 and not np.allclose(want_s.numpy()[0], _gm_run(0, 1, steps=0)[1].numpy()[0])


def test_gm_trainer_momentum_and_fresh_network_per_iteration():
    losses, syn = _gm_run(0, 1, steps=2)
    assert len(losses) == 2 and all(np.isfinite(losses)) and losses[0] != losses[1]


# ---- MTT trainer: explicit reverse sweep vs autograd through the unrolled loop; batch sharding ----

def _mtt_setup(rank=0, world=1):
    from tests.cpu_backend import OracleMTTOps
    z = np.load(os.path.join(GOLDEN, "g10_mtt_step.npz"))
    C, n_syn = int(z["C"]), int(z["n_syn"])
    start = R.init_params(int(z["net_seed"]), 3, C)
    g = torch.Generator().manual_seed(int(z["data_seed"]))
    target = [p + 0.02 * p.abs().mean() * torch.randn(p.shape, generator=g) for p in start]
    image_syn = torch.randn(n_syn, 8, 3, 64, 64, generator=g)
    tr = distill.MTTTrainer(OracleMTTOps(), C, image_syn.clone(), torch.tensor(z["labels"]), float(z["syn_lr"]), lr_img=100.0,
                            lr_lr=1e-5, syn_steps=int(z["syn_steps"]), batch_syn=int(z["batch_syn"]), expert_epochs=1,
                            max_start_epoch=1, rank=rank, world=world)
    return z, tr, [start, target], [torch.tensor(i) for i in z["indices"]], image_syn


def test_mtt_trainer_reverse_sweep_matches_reference_golden():
    """Fixture G10 (reference ReparamModule + autograd): grand loss, d/d image_syn, d/d syn_lr from the
    trainer's explicit reverse sweep; then the optimiser bookkeeping of the update."""
    z, tr, traj, chunks, image_syn = _mtt_setup()
    grand = tr.step(0, traj, start_epoch=0, index_chunks=chunks)
    g_img, g_lr = tr.last_grads
    assert abs(grand - float(z["grand_loss"])) / float(z["grand_loss"]) < 1e-4
    assert abs(g_lr - float(z["grad_lr"])) / abs(float(z["grad_lr"])) < 2e-3
    want = torch.tensor(z["grad_img"])
    assert float((g_img[:, ::2, :, ::2, ::2] - want).norm() / want.norm()) < 2e-3
    np.testing.assert_allclose(tr.image_syn.numpy(), (image_syn - 100.0 * g_img).numpy(), rtol=1e-5, atol=1e-6)   # first step: buf = g
    assert abs(float(tr.syn_lr) - max(float(z["syn_lr"]) - 1e-5 * float(g_lr), 0.001)) < 1e-9


def _worker_mtt(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        z, tr, traj, chunks, _ = _mtt_setup(rank, world)
        grand = tr.step(0, traj, start_epoch=0, index_chunks=chunks)
        if rank == 0:
            q.put((float(grand), tr.last_grads[0].numpy(), float(tr.last_grads[1])))
    finally:
        dist.destroy_process_group()


def test_mtt_trainer_batch_sharded_two_ranks_gloo():
    """Each student step's synthetic batch split over 2 ranks + all-reduce of the flat gradient and of
    the Hessian-vector product == the single-rank iteration."""
    z, tr, traj, chunks, _ = _mtt_setup()
    want = tr.step(0, traj, start_epoch=0, index_chunks=chunks, update=False)
    grand, g_img, g_lr = _spawn(_worker_mtt, 2)
    assert abs(grand - want) / want < 1e-5
    assert abs(g_lr - tr.last_grads[1]) / abs(g_lr) < 1e-3
    ref = tr.last_grads[0].numpy()
    assert np.linalg.norm(g_img - ref) / np.linalg.norm(ref) < 1e-3


# ---- s2d under MTT ("MTT+Ours", BASELINE config 5): fixture G13 from the reference's loop; batch sharding ----

def _s2d_mtt_setup(rank=0, world=1, ops=None, dev="cpu"):
    from tests.cpu_backend import OracleMTTOps
    z = np.load(os.path.join(GOLDEN, "g13_s2d_mtt_step.npz"))
    C = int(z["C"])
    start = R.init_params(int(z["net_seed"]), 3, C)
    g = torch.Generator().manual_seed(int(z["data_seed"]))
    target = [p + 0.02 * p.abs().mean() * torch.randn(p.shape, generator=g) for p in start]
    static = torch.randn(C * int(z["spc"]), 3, 64, 64, generator=g)
    dynamic = torch.randn(C, int(z["dpc"]), 8, 1, 64, 64, generator=g)
    tr = distill.S2DMTTTrainer(ops or OracleMTTOps(), C, int(z["vpc"]), int(z["spc"]), int(z["dpc"]), static.clone().to(dev),
                               dynamic.clone().to(dev), torch.tensor(z["hal_w"]).to(dev), torch.tensor(z["hal_b"]).to(dev),
                               float(z["syn_lr"]), lr_dynamic=float(z["lr_dynamic"]), lr_hal=float(z["lr_hal"]),
                               lr_lr=float(z["lr_lr"]), syn_steps=int(z["syn_steps"]), batch_syn=int(z["batch_syn"]),
                               expert_epochs=1, max_start_epoch=1, lr_static=float(z["lr_static"]), train_static=True,
                               rank=rank, world=world)
    chunks = [torch.tensor(row[row >= 0]) for row in z["indices"]]
    tr.draws = [(d[d >= 0], s[s >= 0]) for d, s in zip(z["draws_dyn"], z["draws_sta"])]
    return z, tr, [start, target], chunks


def check_s2d_mtt_against_g13(z, tr, grand, tol=2e-3):
    C, dpc = int(z["C"]), int(z["dpc"])
    g_dyn, g_w, g_b, g_stat, g_lr = tr.last_grads
    rel = lambda a, b: float((torch.as_tensor(a).double().cpu() - torch.as_tensor(b).double()).norm() / torch.as_tensor(b).double().norm())   # noqa: E731
    assert abs(float(grand) - float(z["grand_loss"])) / float(z["grand_loss"]) < 1e-4
    assert abs(float(g_lr) - float(z["grad_lr"])) / abs(float(z["grad_lr"])) < tol
    gd = g_dyn.view(C, dpc, 8, 1, 64, 64).cpu()
    assert rel(gd[:, :, :, :, ::4, ::4], z["g_dynamic"]) < tol
    assert rel(g_stat.cpu()[:, :, ::4, ::4], z["g_static"]) < tol
    assert rel(g_w, z["g_hal_w"]) < tol and rel(g_b, z["g_hal_b"]) < tol
    # memories no student step drew stay untouched: exactly zero gradient rows
    assert ((gd.abs().sum(dim=(2, 3, 4, 5)) == 0) == torch.tensor(z["g_dynamic_rowabs"] == 0)).all()
    assert ((g_stat.cpu().abs().sum(dim=(1, 2, 3)) == 0) == torch.tensor(z["g_static_rowabs"] == 0)).all()
    # the four optimiser steps (first step: buf = g) and the clipped syn_lr
    np.testing.assert_allclose(tr.hal_w.cpu().numpy(), z["hal_w_after"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(tr.hal_b.cpu().numpy(), z["hal_b_after"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(tr.dynamic.view(C, dpc, 8, 1, 64, 64).cpu()[:, :, :, :, ::4, ::4].numpy(), z["dynamic_after"],
                               rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(tr.static.cpu()[:, :, ::4, ::4].numpy(), z["static_after"], rtol=1e-3, atol=1e-4)
    assert abs(float(tr.syn_lr) - float(z["syn_lr_after"])) < 1e-7


def test_s2d_mtt_trainer_matches_reference_golden():
    z, tr, traj, chunks = _s2d_mtt_setup()
    grand = tr.step(0, traj, start_epoch=0, index_chunks=chunks)
    check_s2d_mtt_against_g13(z, tr, grand)


def _worker_s2d_mtt(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        z, tr, traj, chunks = _s2d_mtt_setup(rank, world)
        grand = tr.step(0, traj, start_epoch=0, index_chunks=chunks)
        if rank == 1:       # any rank holds the full result after the all-reduces
            q.put((float(grand), [t.numpy() for t in tr.last_grads[:4]], float(tr.last_grads[4]), tr.hal_w.numpy(), float(tr.syn_lr)))
    finally:
        dist.destroy_process_group()


def test_s2d_mtt_trainer_batch_sharded_two_ranks_gloo():
    """Student batches of 1 and 2 composed clips split over 2 ranks (one rank idles in the single-item steps): the
    all-reduced memories / hallucinator gradients and the updates equal the single-rank iteration."""
    z, tr, traj, chunks = _s2d_mtt_setup()
    want = float(tr.step(0, traj, start_epoch=0, index_chunks=chunks))
    grand, grads, g_lr, hal_w, syn_lr = _spawn(_worker_s2d_mtt, 2)
    assert abs(grand - want) / want < 1e-5
    for got, ref in zip(grads, tr.last_grads[:4]):
        assert np.linalg.norm(got - ref.numpy()) / np.linalg.norm(ref.numpy()) < 1e-3
    assert abs(g_lr - float(tr.last_grads[4])) / abs(g_lr) < 1e-3
    np.testing.assert_allclose(hal_w, tr.hal_w.numpy(), rtol=1e-4, atol=1e-7)
    assert abs(syn_lr - float(tr.syn_lr)) < 1e-8


# ---- the bench's own partition at full width: C = 50 classes over 8 ranks (blocks 7,7,6,6,6,6,6,6; batch mode 8 x 8) ----

def _c50_run(rank, world, shard):
    """One DM iteration, C=50, ipc=1, batch_real=8, 64x64x8 clips generated from a seed (identical on every rank)."""
    C, B = 50, 8
    g = torch.Generator().manual_seed(5050)
    clips = torch.randn(C * B, 8, 3, 64, 64, generator=g)
    syn = torch.randn(C, 8, 3, 64, 64, generator=g)
    pool = _Pool(); pool.clips = clips; pool.counts = [B] * C; pool.offsets = [c * B for c in range(C)]
    lo, hi = distill.class_range(C, rank, world)
    own = list(range(lo, hi))
    if shard == "hybrid" and world > 1:
        block, _, mine = distill.hybrid_partition(C, rank, world)
        own = block + mine
    tr = distill.DMTrainer(OracleBackend(), pool, C, 1, B, lr_img=0.5, momentum=0.5, rank=rank, world=world,
                           image_syn=syn[own].clone(), shard=shard)
    losses = [float(tr.global_loss(tr.step(it))) for it in range(1)]
    return losses, tr.gather_syn(), ((lo, hi) if shard != "hybrid" else (len(own), own[-1]))


def _worker_c50(rank, world, port, q, shard):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        losses, syn, span = _c50_run(rank, world, shard)
        spans = [None] * world
        dist.all_gather_object(spans, span)
        if rank == 0:
            q.put((losses, syn.numpy(), spans))
    finally:
        dist.destroy_process_group()


def _worker_c50_class(rank, world, port, q):
    _worker_c50(rank, world, port, q, "class")


def _worker_c50_batch(rank, world, port, q):
    _worker_c50(rank, world, port, q, "batch")


def _worker_c50_hybrid(rank, world, port, q):
    _worker_c50(rank, world, port, q, "hybrid")


_C50_CACHE = {}


@pytest.mark.parametrize("worker", [_worker_c50_class, _worker_c50_batch, _worker_c50_hybrid])
def test_dm_trainer_eight_ranks_fifty_classes(worker):
    """world 8, C = 50: uneven class blocks (7,7,6,...) with ragged all-gather of the synthetic clips, the batch
    mode with one real clip of every class per rank, and the hybrid (6 whole classes per rank + classes 48 / 49 split eight
    ways, owned by ranks 0 / 1); all equal the single-rank run."""
    if "ref" not in _C50_CACHE:
        torch.set_num_threads(8)
        _C50_CACHE["ref"] = _c50_run(0, 1, "class")
    want_l, want_syn, _ = _C50_CACHE["ref"]
    losses, syn, spans = _spawn(worker, 8)
    if worker is _worker_c50_hybrid:
        assert spans == [(7, 48), (7, 49), (6, 17), (6, 23), (6, 29), (6, 35), (6, 41), (6, 47)]    # (clips owned, last class)
    else:
        assert [b - a for a, b in spans] == [7, 7, 6, 6, 6, 6, 6, 6] and spans[0][0] == 0 and spans[-1][1] == 50
    np.testing.assert_allclose(losses, want_l, rtol=2e-5)
    np.testing.assert_allclose(syn, want_syn.numpy(), rtol=1e-4, atol=1e-6)


def test_class_with_fewer_clips_than_batch_real_is_rejected():
    """The reference takes the n available clips and averages over n (distill_baseline.py:85); the batched kernels need
    equal batches, so a short class is an error (never a silently re-weighted mean)."""
    with pytest.raises(ValueError):
        distill.sample_real_indices(1, [3], [0], 8, [0])
    assert distill.sample_real_indices(1, [3], [0], 8, [0], allow_repeat=True).shape == (8,)


# ---- worlds that do not divide the reference's batch of 64 (3, 5, 6, 7): every decomposition, and the fallback ----

def _odd_setup(world_for_batch):
    """C = 11 classes (prime: uneven blocks for every world), real batch = 2 x world clips so that the batch and hybrid
    decompositions divide, 64x64x8 clips from a seed (identical on every rank)."""
    C, B = 11, 2 * world_for_batch
    g = torch.Generator().manual_seed(1100 + world_for_batch)
    clips = torch.randn(C * B, 8, 3, 64, 64, generator=g)
    syn = torch.randn(C, 8, 3, 64, 64, generator=g)
    pool = _Pool(); pool.clips = clips; pool.counts = [B] * C; pool.offsets = [c * B for c in range(C)]
    return C, B, pool, syn


def _odd_trainer(rank, world, shard, wb):
    C, B, pool, syn = _odd_setup(wb)
    own = list(range(*distill.class_range(C, rank, world)))
    if shard == "hybrid" and world > 1:
        block, _, mine = distill.hybrid_partition(C, rank, world)
        own = block + mine
    tr = distill.DMTrainer(OracleBackend(), pool, C, 1, B, lr_img=0.5, momentum=0.5, rank=rank, world=world,
                           image_syn=syn[own].clone(), shard=shard)
    loss = float(tr.global_loss(tr.step(0)))
    return loss, tr.gather_syn(), len(own)


def _worker_odd(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out = {}
        for shard in ("class", "batch", "hybrid"):
            loss, syn, nown = _odd_trainer(rank, world, shard, world)
            owned = [None] * world
            dist.all_gather_object(owned, nown)
            out[shard] = (loss, syn.numpy(), owned)
        if rank == 0:
            q.put(out)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [3, 5, 6, 7])
def test_dm_trainer_odd_worlds_every_decomposition(world):
    """Worlds that do not divide the reference's real batch of 64 (utils.py:615-623 is what the ranks replace: nn.DataParallel
    splits ANY batch): whole-class blocks, the batch-sharded real side and the hybrid on 3 / 5 / 6 / 7 gloo ranks, 11 classes
    (uneven blocks everywhere; hybrid: 11 % world left-over classes split world ways, owned round-robin), a real batch that the
    world divides -- loss and all synthetic clips after one step equal the single-rank run; every class is owned exactly once."""
    torch.set_num_threads(8)
    want_l, want_syn, _ = _odd_trainer(0, 1, "class", world)
    out = _spawn(_worker_odd, world)
    for shard in ("class", "batch", "hybrid"):
        loss, syn, owned = out[shard]
        assert sum(owned) == 11, (shard, owned)
        if shard == "hybrid":
            assert max(owned) - min(owned) <= 1 and owned == [11 // world + (1 if r < 11 % world else 0) for r in range(world)]
        np.testing.assert_allclose(loss, want_l, rtol=2e-5, err_msg=shard)
        np.testing.assert_allclose(syn, want_syn.numpy(), rtol=1e-4, atol=1e-6, err_msg=shard)


def test_shard_choice_falls_back_to_class_blocks_when_the_batch_does_not_divide():
    """``distill.choose_shard`` (what ``bench.py --shard auto`` runs): with the reference's 64-clip real batches the hybrid needs
    ``64 % world == 0``; worlds 3, 5, 6, 7 fall back to whole-class blocks, and asking for the batch / hybrid decomposition there
    is an error that says so (never a silently unequal split)."""
    assert [distill.choose_shard(50, 64, w) for w in (1, 2, 3, 4, 5, 6, 7, 8)] == ["class", "class", "class", "class", "class", "class", "class", "hybrid"]
    assert distill.choose_shard(51, 64, 8) == "hybrid" and distill.choose_shard(400, 256, 8) == "class"      # 400 = 8 x 50: even blocks
    assert distill.choose_shard(50, 64, 8, method="dc") == "class"
    assert distill.choose_shard(50, 63, 7) == "hybrid" and distill.choose_shard(50, 64, 7) == "class"        # (blocks 8,7,7,...: 12 % uneven)
    assert distill.choose_shard(50, 60, 6) == "class"                                                        # (blocks 9,9,8,...: exactly 8 %)
    C, B, pool, syn = _odd_setup(3)          # batch of 6
    for world, shard in ((4, "batch"), (4, "hybrid"), (5, "hybrid")):
        with pytest.raises(ValueError, match="divisible by the number of ranks"):
            distill.DMTrainer(OracleBackend(), pool, C, 1, B, lr_img=0.5, rank=0, world=world, image_syn=syn[:3].clone(), shard=shard)
