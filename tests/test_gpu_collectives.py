"""The RCCL data-path collectives of the trainers, EXECUTED on a one-GPU box: bench.py runs in child processes with a
1-rank ``nccl`` process group (VD_BENCH_FORCE_DIST=1) and ``VD_FORCE_COLLECTIVES=1`` / ``VD_FORCE_BATCH_SHARD=1``, which keep
every ``world > 1`` branch alive -- the feature-sum all-reduce of the batch / hybrid decompositions on the synthetic-clip
stream, the hallucinator-gradient all-reduce of s2d, the loss all-reduces, the all-gather of the synthetic clips before
evaluation, DC's per-step loss all-reduce incl. the profiling step every rank must take, MTT's flat-gradient / Hessian-vector
all-reduces per student step.  A 1-rank collective is the identity, so every run must reproduce the plain run's loss; what
the test adds is that the calls are issued to RCCL from the trainers' streams and complete (reference: the only multi-GPU
mechanism upstream is nn.DataParallel, utils.py:615-623).  The N-rank arithmetic of the same code is covered on gloo
(tests/test_distributed_cpu.py)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--sustain-seconds", "0", "--no-extra-legs"]
FORCE = {"VD_BENCH_FORCE_DIST": "1", "VD_FORCE_COLLECTIVES": "1", "VD_FORCE_BATCH_SHARD": "1", "MASTER_PORT": "29541"}


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def _together(*jobs):
    """Run the child benchmarks of one test CONCURRENTLY (round 6: these are logic runs of 2 - 3 tiny steps whose time is process
    start-up -- import, planning, engine construction -- and the children share the one GPU without touching each other's
    results; serially the file took 195 s of the driver's suite).  -> the jobs' results, in order."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
        futs = [ex.submit(j) for j in jobs]
        return [f.result() for f in futs]


def _bench(extra, env=None):
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    e.update(env or {})
    if "MASTER_PORT" in e:          # a fresh rendezvous port per child: consecutive one-rank groups on one fixed port have collided
        e["MASTER_PORT"] = _free_port()
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "1"] + COMMON + extra, cwd=ROOT, env=e, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def test_dm_decompositions_issue_their_collectives_and_keep_the_loss():
    small = ["--classes", "6", "--pool-per-class", "70", "--eval-epochs", "1"]
    shards = (("class", 1), ("batch", 2), ("hybrid", 2))
    # (hybrid on one rank has no class left over; VD_HYBRID_FORCE_SPLIT makes the last two classes split ones)
    plain, *runs = _together(lambda: _bench(small + ["--shard", "class"]),
                             *[lambda sh=sh: _bench(small + ["--shard", sh], dict(FORCE, VD_HYBRID_FORCE_SPLIT="2")) for sh, _ in shards])
    assert plain["collectives"]["all_reduce"] == 0 and plain["collectives"]["backend"] is None
    for (shard, per_step), got in zip(shards, runs):
        c = got["collectives"]
        assert c["backend"] == "nccl" and c["forced_on_one_rank"]
        # 3 steps: the loss all-reduce each, + the feature-sum all-reduce of the batch / hybrid split; one all-gather of the
        # synthetic clips before evaluate_synset
        assert c["all_reduce"] == 3 * per_step and c["all_gather"] == 1, (shard, c)
        assert abs(got["loss_last"] / plain["loss_last"] - 1) < 1e-5, (shard, got["loss_last"], plain["loss_last"])


def test_s2d_dc_mtt_issue_their_collectives_and_keep_the_loss():
    s2d = ["--method", "s2d", "--classes", "4", "--pool-per-class", "70", "--eval-epochs", "0"]
    dc = ["--method", "dc", "--classes", "3", "--ipc", "1", "--frames", "8", "--size", "64", "--batch-real", "8", "--pool-per-class", "12"]
    # (bench.py seeds torch's generators, so both runs draw the same dropout masks; the ORDER of the fp32 atomics is left, and the
    #  matching loss amplifies it when a near-tie of a pooling window goes the other way: 1e-7 in most runs, 1.1e-4 seen once --
    #  so both legs run in the ordered mode, where the training step's sums have a fixed order)
    det = {"VD_DETERMINISTIC": "1"}
    mtt = ["--method", "mtt", "--classes", "8", "--frames", "8", "--size", "64", "--syn-steps", "2", "--batch-syn", "8"]
    sa, sb, da, db, ma, mb = _together(lambda: _bench(s2d), lambda: _bench(s2d, FORCE), lambda: _bench(dc, det),
                                       lambda: _bench(dc, dict(FORCE, **det)), lambda: _bench(mtt, det), lambda: _bench(mtt, dict(FORCE, **det)))
    a, b = sa, sb
    assert b["collectives"]["all_reduce"] == 3 * 2 and abs(b["loss_last"] / a["loss_last"] - 1) < 1e-5
    a, b = da, db
    assert b["collectives"]["all_reduce"] == 3 + 1            # every step's loss, incl. the extra profiling step of bench_dc
    assert abs(b["loss_last"] / a["loss_last"] - 1) < 1e-4 and b["roofline"]["launches"] > 0
    a, b = ma, mb
    # per iteration: flat gradient + Hessian-vector product per student step (2 x 2), hallucinator + dynamic-memory gradients (2);
    # 3 timed / warm-up iterations + 1 profiling iteration
    assert b["collectives"]["all_reduce"] == 4 * (2 * 2 + 2), b["collectives"]
    assert abs(b["grand_loss_last"] / a["grand_loss_last"] - 1) < 1e-4      # (same seed, same dropout masks: atomics' order only)


def test_two_ranks_share_the_gpu_over_gloo_and_match_one_rank():
    """The N = 2 data path with the HIP kernels underneath, on a one-GPU box: ``bench.py --gpus 2`` launched as the driver does
    (torch.distributed.run, one process per rank) with ``VD_BENCH_ONE_DEVICE=1`` -- both ranks on device 0, exchange over gloo
    (RCCL refuses two ranks on one device; the RCCL calls themselves run in the tests above).  Class blocks (4 + 3 classes),
    batch split (every class's real batch halved, feature sums all-reduced) and the hybrid (3 + 3 whole classes, the seventh
    class's batch split) must reproduce the one-rank loss; s2d all-reduces its hallucinator gradients."""
    def two(extra, port):
        e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", VD_BENCH_ONE_DEVICE="1")
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                              "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2"] + COMMON + extra,
                             cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-3000:]
        return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    small = ["--classes", "7", "--pool-per-class", "70", "--eval-epochs", "1"]
    s2d = ["--method", "s2d", "--classes", "4", "--pool-per-class", "70", "--eval-epochs", "0"]
    dc = ["--method", "dc", "--classes", "3", "--ipc", "1", "--frames", "8", "--size", "64", "--batch-real", "8", "--pool-per-class", "12"]
    mtt = ["--method", "mtt", "--classes", "8", "--frames", "8", "--size", "64", "--syn-steps", "2", "--batch-syn", "8"]
    shards = [(k, sh, per) for k, (sh, per) in enumerate((("class", 1), ("batch", 2), ("hybrid", 2)))
              if sh != "hybrid" or os.environ.get("VD_TEST_ALL_SHARDS") == "1"]      # (the eight-rank test below runs the hybrid decomposition)
    # every child of this test at once: four one-rank references and the two-rank runs (a fixed rendezvous port each)
    one, s2d_a, dc_a, mtt_a, s2d_b, dc_b, mtt_b, *two_runs = _together(
        lambda: _bench(small + ["--shard", "class"]), lambda: _bench(s2d), lambda: _bench(dc), lambda: _bench(mtt),
        lambda: two(s2d, 29555), lambda: two(dc, 29556), lambda: two(mtt, 29557),
        *[lambda k=k, sh=sh: two(small + ["--shard", sh], 29551 + k) for k, sh, _ in shards])
    for (k, shard, per_step), got in zip(shards, two_runs):
        c = got["collectives"]
        assert got["n_gpus"] == 2 and c["backend"] == "gloo" and not c["forced_on_one_rank"]
        assert c["all_reduce"] == 3 * per_step and c["all_gather"] == 1, (shard, c)
        assert abs(got["loss_last"] / one["loss_last"] - 1) < 1e-4, (shard, got["loss_last"], one["loss_last"])
        assert "top1" in got["eval"]                                          # rank 0 trained a net on the GATHERED synthetic clips
    a, b = s2d_a, s2d_b
    assert b["collectives"]["all_reduce"] == 3 * 2 and abs(b["loss_last"] / a["loss_last"] - 1) < 1e-4
    # gradient matching: classes 2 + 1, every rank's loss all-reduced per step; trajectory matching: the student batch of 8 split
    # 4 + 4, flat gradient and Hessian-vector product all-reduced per student step (tolerances: see the one-rank test above)
    a, b = dc_a, dc_b
    # (each rank draws its classes' dropout masks from the same seeded generator, i.e. other masks than the one-rank run's:
    #  the losses agree to the dropout noise of this tiny configuration, +-2.5 %)
    assert b["n_gpus"] == 2 and b["collectives"]["all_reduce"] == 3 + 1 and abs(b["loss_last"] / a["loss_last"] - 1) < 8e-2
    a, b = mtt_a, mtt_b
    assert b["n_gpus"] == 2 and b["collectives"]["all_reduce"] == 4 * (2 * 2 + 2), b["collectives"]
    assert abs(b["grand_loss_last"] / a["grand_loss_last"] - 1) < 8e-2


def test_a_failing_rank_fails_the_spawned_run():
    """``bench.py --gpus 2`` (self-spawned): a rank that dies makes the parent stop the other rank and exit non-zero, with no JSON
    line on stdout (VD_BENCH_FAIL_RANK is the test's fault injection)."""
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", VD_BENCH_ONE_DEVICE="1", VD_BENCH_FAIL_RANK="1")
    e.pop("WORLD_SIZE", None); e.pop("RANK", None)
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2"] + COMMON + ["--classes", "4", "--pool-per-class", "70", "--eval-epochs", "0"],
                         cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert "rank 1 exited" in out.stderr


def test_pixel_gradient_allreduce_leg():
    """The literal "all-reduce of the matching-loss gradient" as a counted mode (DMTrainer(exchange='allreduce')): (i) one rank,
    RCCL: the full gradient tensor goes through vd_comm_allreduce_f32 on the synthetic-clip stream (a 1-rank all-reduce is the
    identity, so the leg's loss is the owner-computes loss); (ii) two self-spawned ranks on device 0 over gloo: the reduced
    tensor's rows give the same update as owner-computes.  The JSON line carries bytes and ms of the exchange."""
    small = ["--classes", "6", "--pool-per-class", "70", "--eval-epochs", "0", "--exchange-leg"]
    long = ["--steps", "50", "--warmup", "2"]       # (round 6) the two communicators interleaved for 52 + 5 steps, not 3 + 5: torch's process
    #                                                  group (loss all-reduce, barriers) and hip.Comm (the 120 MB-shaped gradient
    #                                                  tensor on the synthetic-clip stream) -- the order `--exchange allreduce` issues them in

    def spawned_two():
        e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", VD_BENCH_ONE_DEVICE="1")
        e.pop("WORLD_SIZE", None); e.pop("RANK", None)
        out = subprocess.run([sys.executable, "bench.py", "--gpus", "2"] + COMMON + small, cwd=ROOT, env=e, capture_output=True, text=True,
                             timeout=900)
        assert out.returncode == 0, out.stderr[-3000:]
        return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    one, one_long, other, two = _together(
        lambda: _bench(small, dict(FORCE, VD_BENCH_EXCHANGE_LEG="1", MASTER_PORT="29543")),
        lambda: _bench(small + long, dict(FORCE, VD_BENCH_EXCHANGE_LEG="1", MASTER_PORT="29545")),
        lambda: _bench(small + long + ["--exchange", "allreduce"], dict(FORCE, VD_BENCH_EXCHANGE_LEG="1", MASTER_PORT="29544")),
        spawned_two)
    leg = one["exchange_allreduce"]
    assert leg["through"].startswith("vd_comm_allreduce_f32") and leg["allreduce_calls"] == 5
    assert leg["pixel_gradient_bytes_per_step"] == 6 * 16 * 3 * 112 * 112 * 4 and leg["allreduce_ms_mean"] > 0
    assert one["rccl"]["nranks"] == 1 and one["rccl"]["version_code"] > 20000 and one["exchange"]["mode"] == "owner"
    # the leg's trainer starts from the same initial clips and runs iterations 0..6; the main run's loss_last is iteration 2:
    # compare the two modes at equal iterations instead -- a main run timed in allreduce mode whose leg is owner-computes
    assert other["exchange"]["mode"] == "allreduce" and other["exchange"]["allreduce_calls"] >= 50 and other["steps"] == 50
    assert other["rccl"]["nranks"] == 1 and other["ranks_seen"] == 1 and other["value"] is not None and "refused" not in other
    assert abs(other["loss_last"] / one_long["loss_last"] - 1) < 1e-5          # 52 iterations in either mode: the same clips
    assert abs(other["exchange_owner"]["loss_last"] / leg["loss_last"] - 1) < 1e-5
    assert two["ranks_seen"] == 2 and two["clips_per_step"] == [192, 192] and two["rccl"].get("nranks") is None
    # per-rank diagnostics of the timed region (round 5): what each rank did and how long it waited for the others
    pr = two["per_rank"]
    assert [r["rank"] for r in pr] == [0, 1] and all(r["clips_per_step"] == 192 and r["classes_owned"] == 3 for r in pr)
    assert all(r["own_ms_per_step"] > 0 and r["wait_at_barrier_ms"] >= 0 and r["real_side_ms_per_step"] > 0 for r in pr)
    assert min(r["wait_at_barrier_ms"] for r in pr) < 50          # the slower rank does not wait
    assert other["per_rank"][0]["exchange_ms_per_step"] > 0 and one["per_rank"][0]["exchange_ms_per_step"] is None
    assert two["exchange_allreduce"]["through"] == "torch.distributed all_reduce (gloo)"
    assert abs(two["exchange_allreduce"]["loss_last"] / leg["loss_last"] - 1) < 1e-4
    assert abs(two["loss_last"] / one["loss_last"] - 1) < 1e-4


def test_eight_ranks_share_the_gpu_with_the_default_decomposition():
    """``bench.py --gpus 8`` with NO launcher in front (the shape of the driver's one-GPU command; bench.py spawns the ranks;
    the launcher form is exercised by the two-rank test above), eight processes, default --shard auto ->
    the hybrid decomposition: 6 whole classes per rank + the real batches of classes 48 and 49 split eight ways), all ranks on
    device 0 over gloo: the all-reduced DM loss equals the one-rank loss, rank 0 evaluates the gathered 50 synthetic clips."""
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", VD_BENCH_ONE_DEVICE="1")
    e.pop("WORLD_SIZE", None); e.pop("RANK", None)
    args = ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--sustain-seconds", "0", "--no-extra-legs", "--eval-epochs", "1",
            "--eval-seeds", "1"]
    # NO launcher in front: bench.py starts its eight ranks itself (spawn_ranks) and relays rank 0's line
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "8"] + args, cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line on stdout"
    got = json.loads(lines[-1])
    one = _bench(args)
    assert got["n_gpus"] == 8 and got["config"]["parallelism"].startswith("hybrid x8") and one["n_gpus"] == 1
    assert got["ranks_seen"] == 8 and got["launched_by"] == "bench.py spawn_ranks" and one["ranks_seen"] == 1
    assert got["clips_per_step"] == [400] * 8 and one["clips_per_step"] == [3200]          # 6 whole classes + 1/8 of two = 400 per rank
    c = got["collectives"]
    assert c["backend"] == "gloo" and c["all_reduce"] == 3 * 2 and c["all_gather"] == 1, c      # per step: split-class feature sums + loss
    assert abs(got["loss_last"] / one["loss_last"] - 1) < 1e-5, (got["loss_last"], one["loss_last"])
    assert got["eval"]["test_clips"] > 0


def test_vd_comm_c_abi_one_rank_roundtrip():
    """vd_comm_* (include/vd_hip.h): a one-rank RCCL communicator created through the C ABI; all-reduce (in place and out of
    place) and all-gather on a side stream return the data unchanged, sizes / ranks / argument errors are reported."""
    import ctypes
    import torch
    from video_distillation_amd import hip
    L = hip.lib()
    ident = (ctypes.c_char * 128)()
    rc = L.vd_comm_unique_id(ident)
    assert rc == 0, "RCCL not reachable through dlopen (code %d)" % rc
    comm = ctypes.c_void_p()
    assert L.vd_comm_create(ident, 1, 1, ctypes.byref(comm)) == -1          # rank out of range
    hip.check(L.vd_comm_create(ident, 1, 0, ctypes.byref(comm)), "vd_comm_create")
    try:
        assert L.vd_comm_size(comm) == 1 and L.vd_comm_rank(comm) == 0
        st = torch.cuda.Stream()
        x = torch.randn(50 * 2048, device="cuda")
        want = x.clone()
        y = torch.zeros_like(x)
        z = torch.zeros_like(x)
        torch.cuda.synchronize()
        with torch.cuda.stream(st):
            sp = ctypes.c_void_p(st.cuda_stream)
            hip.check(L.vd_comm_allreduce_f32(comm, hip.ptr(x), hip.ptr(y), ctypes.c_int64(x.numel()), sp), "allreduce")
            hip.check(L.vd_comm_allreduce_f32(comm, hip.ptr(x), hip.ptr(x), ctypes.c_int64(x.numel()), sp), "allreduce in place")
            hip.check(L.vd_comm_allgather_f32(comm, hip.ptr(x), hip.ptr(z), ctypes.c_int64(x.numel()), sp), "allgather")
        st.synchronize()
        assert torch.equal(y, want) and torch.equal(x, want) and torch.equal(z, want)
        assert L.vd_comm_allreduce_f32(comm, None, None, ctypes.c_int64(4), None) == -1
        assert L.vd_comm_allreduce_f32(comm, None, None, ctypes.c_int64(0), None) == 0
    finally:
        L.vd_comm_free(comm)
