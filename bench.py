#!/usr/bin/env python3
"""Headline benchmark: DM distillation steps/s (miniUCF101 shape, IPC=1) on N MI355X.

One step = one iteration of the reference's DM loop body (distill_baseline.py:334-355):
fresh random ConvNet3D, C=50 class terms (64 real + ipc synthetic 112x112x16 clips each),
backward to the synthetic pixels, SGD-momentum step.  Synthetic data (no dataset ships):
randn clips standardised per channel, 93 clips per class, generated on the device.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 10 --warmup 3

Rank 0 prints ONE JSON line (contract in the task description) with two extra objects:
  roofline      -- the dominant kernel (conv_mfma, second conv layer, real-clip forward):
                   algorithmic FLOP per launch / mean launch time (HIP events on the launch
                   stream) against the 2.5 PFLOP/s dense 16-bit MFMA peak;
  cpu_baseline  -- the CPU oracle (torch fp32 ops == what the reference runs on a CPU) timed on
                   this host on a bounded sample (whole class terms), extrapolated to steps/s.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--classes", type=int, default=50)
    ap.add_argument("--ipc", type=int, default=1)
    ap.add_argument("--batch-real", type=int, default=64)
    ap.add_argument("--pool-per-class", type=int, default=93)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--size", type=int, default=112)
    ap.add_argument("--prec-real", default=os.environ.get("VD_PREC_REAL", "f16"))
    ap.add_argument("--prec-syn", default=os.environ.get("VD_PREC_SYN", "f16x3"))
    ap.add_argument("--prec-bwd", default=os.environ.get("VD_PREC_BWD", "f16"),
                    help="operand precision of the input-gradient passes (single-pass fp16 with power-of-two scaling)")
    ap.add_argument("--chunk", type=int, default=3200, help="real clips per launch (all of a single-GPU step by default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-classes", type=int, default=10, help="class terms timed for the CPU baseline")
    ap.add_argument("--shard", default="auto", choices=["auto", "class", "batch"],
                    help="multi-GPU decomposition: whole classes per rank, or 1/N of every class's real batch per rank "
                         "(+ one all-reduce of the per-class feature sums); auto = batch when batch_real %% N == 0")
    ap.add_argument("--syn-steps", type=int, default=10, help="--method mtt: unrolled student steps (sh/baseline/MTT.sh)")
    ap.add_argument("--batch-syn", type=int, default=256, help="--method mtt: synthetic clips per student step")
    ap.add_argument("--method", default="dm", choices=["dm", "s2d", "dc", "mtt"],
                    help="dm = distill_baseline.py DM (headline); s2d = DM + static/dynamic memories (config 3); "
                         "dc = gradient matching with match_loss (config 4: use --classes 51 --ipc 5); "
                         "mtt = trajectory matching (config 5: use --classes 400 --frames 8 --size 64)")
    ap.add_argument("--eval-epochs", type=int, default=0,
                    help="after the timed steps, run evaluate_synset on the synthetic clips for this many epochs (HIP train "
                         "step + HIP inference, SURVEY 8(d): eval top-1 beside the metric) and report it under 'eval'")
    ap.add_argument("--dis-metric", default="ours", choices=["ours", "mse", "cos"], help="match_loss metric of --method dc")
    return ap.parse_args()


def conv_layer_macs(geo):
    return [d[1] * d[5] * d[6] * d[7] * d[0] * 147 for d in geo.layer_dims()]


def pmc_traffic(clips_per_launch):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/r01_pmc_traffic.json: separate FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled per
    MI355X_MICROARCH.md), scaled to this run's clips per launch; None if no PMC pass is on file."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if not os.path.exists(path):
        return None
    rec = json.load(open(path))["conv1_fwd_f16"]
    return rec["hbm_bytes_per_launch"] * clips_per_launch / rec["clips_per_launch"]


def cpu_baseline(args, trainer, backend, it, geo):
    """Time the oracle on `cpu_classes` whole class terms (fwd 64 real + ipc syn, bwd to pixels)
    with the SAME weights and clips the GPU used for iteration `it`; also returns the loss
    parity of those class terms."""
    from oracle import ref_cpu as R
    from video_distillation_amd import distill
    ncls = min(args.cpu_classes, len(trainer.classes))
    classes = trainer.classes[:ncls]
    weights = backend.new_network(seed=it)
    params = [w.cpu() for w in weights]
    idx = distill.sample_real_indices(it, trainer.pool.counts, trainer.pool.offsets, args.batch_real, classes)
    real = trainer.pool.clips[torch.as_tensor(idx, device=trainer.pool.clips.device)].cpu()
    syn = trainer.image_syn[:ncls * args.ipc].detach().clone()
    # GPU loss of exactly these class terms
    backend.set_weights(weights)
    f_real = backend.embed_pool(trainer.pool.clips, torch.as_tensor(idx, device=syn.device))
    f_syn, _ = backend.embed_keep(syn)
    loss_gpu = float(backend.dm_loss(f_real, f_syn, ncls)[0].sum())
    reals = [real[c * args.batch_real:(c + 1) * args.batch_real] for c in range(ncls)]
    # thread count: the best of a short calibration (all logical CPUs is NOT the fastest on this
    # host: 256 threads ran 10x slower than 32 on the 2 x 64-core EPYC of the GPU box)
    ncpu = os.cpu_count() or 1
    best = (None, 1)
    for th in sorted({t for t in (8, 16, 32, 64, 128, ncpu) if t <= ncpu}):
        torch.set_num_threads(th)
        with torch.no_grad():
            R.convnet3d_embed(reals[0][:4], params)
            tc = time.perf_counter()
            R.convnet3d_embed(reals[0][:8], params)
            tc = time.perf_counter() - tc
        if best[0] is None or tc < best[0]:
            best = (tc, th)
    torch.set_num_threads(best[1])
    t0 = time.perf_counter()
    loss_cpu, _ = R.dm_loss_and_grad(params, reals, syn.cpu(), args.ipc)
    dt = time.perf_counter() - t0
    per_step = dt / ncls * args.classes
    return {"value": 1.0 / per_step, "unit": "steps/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d of %d class terms (%d real + %d syn clips %dx%dx%d each, fwd + bwd to pixels), %.1f s, "
                      "extrapolated x%d; threads = fastest of a calibration over 8..%d (host has %d logical CPUs)" % (
                          ncls, args.classes, args.batch_real, args.ipc, args.size, args.size,
                          args.frames, dt, args.classes // ncls, ncpu, ncpu),
            "loss_rel_err_vs_gpu": abs(loss_gpu - float(loss_cpu)) / abs(float(loss_cpu))}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=device)

    from video_distillation_amd import distill, plan
    geo = plan.NetGeometry(args.frames, args.size, args.size)
    c_lo, c_hi = distill.class_range(args.classes, rank, world)
    from video_distillation_amd.networks import _batch_hint
    backend = distill.HipBackend(geo, device, prec_real=args.prec_real, prec_syn=args.prec_syn, chunk=args.chunk,
                                 prec_bwd=args.prec_bwd,
                                 syn_batch_hint=_batch_hint((c_hi - c_lo) * args.ipc) if os.environ.get("VD_SYN_HINT", "1") == "1" else None)
    shard = args.shard
    if shard == "auto":
        shard = "batch" if (world > 1 and args.batch_real % world == 0 and args.method == "dm") else "class"
    if shard == "batch":   # every rank holds the whole pool (11 GB) and embeds its slice of each class batch
        pool = distill.RealPool.synthetic(args.classes, list(range(args.classes)), args.pool_per_class, geo, device, seed=1234)
    else:
        pool = distill.RealPool.synthetic(args.classes, list(range(c_lo, c_hi)), args.pool_per_class, geo, device,
                                          seed=1234 + rank)
    if args.method == "dc":
        return bench_dc(args, distill, geo, pool, device, rank, world)
    if args.method == "mtt":
        return bench_mtt(args, distill, geo, pool, device, rank, world)
    if args.method == "dm":
        trainer = distill.DMTrainer(backend, pool, args.classes, args.ipc, args.batch_real, lr_img=1.0, momentum=0.5,
                                    rank=rank, world=world, shard=shard)
    else:   # sh/s2d/s2d_DM_ms.sh: vpc 1, spc 2, dpc 2, static frozen, SGD(.95) on dynamic memory + hallucinator
        gen = torch.Generator(device=device); gen.manual_seed(77)
        static_syn = torch.randn(args.classes * 2, 3, args.size, args.size, device=device, generator=gen)
        dynamic_syn = torch.randn(args.classes, 2, args.frames, 1, args.size, args.size, device=device, generator=gen)
        hal_w = torch.empty(3, 4, 3, 3, 3, device=device).uniform_(-0.096, 0.096, generator=gen)
        hal_b = torch.empty(3, device=device).uniform_(-0.096, 0.096, generator=gen)
        trainer = distill.S2DTrainer(backend, pool, args.classes, 1, 2, 2, args.batch_real, static_syn, dynamic_syn,
                                     hal_w, hal_b, lr_dynamic=1.0, lr_hal=0.01, rank=rank, world=world)
        trainer.image_syn = trainer.dynamic
        trainer.global_loss = lambda l: l

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for it in range(args.warmup):
        trainer.step(it, overlap=True)
    trainer.sync()
    backend.eng_real.profile = []           # HIP-event pairs around the dominant kernel's launches
    barrier()
    t0 = time.perf_counter()
    losses = []
    for it in range(args.warmup, args.warmup + args.steps):
        losses.append(trainer.global_loss(trainer.step(it, overlap=True)))
    trainer.sync()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
    prof = backend.eng_real.profile
    backend.eng_real.profile = None

    if rank == 0:
        macs = conv_layer_macs(geo)
        step_flop = 2.0 * sum(macs) * (args.classes * (args.batch_real + args.ipc) + args.classes * args.ipc)
        ms_per_step = dt / args.steps * 1e3
        out = {
            "metric": "distillation steps/sec (%s, miniUCF101 IPC=%d)" % ("DM" if args.method == "dm" else "DM+Ours s2d", args.ipc),
            "value": args.steps / dt, "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f16" if args.prec_real == "f16" else args.prec_real, "data": "synthetic",
            "config": {"workload": "miniUCF101-shaped DM IPC=%d: C=%d classes x (%d real + %d syn) clips %dx%dx%d, "
                                   "ConvNet3D depth 3, fresh net per step" % (args.ipc, args.classes, args.batch_real,
                                                                               args.ipc, args.size, args.size, args.frames),
                       "precision": {"real_clips": args.prec_real, "syn_clips_fwd": args.prec_syn, "input_gradient": args.prec_bwd,
                                     "accumulate": "f32"},
                       "parallelism": ("real batch sharded x%d + all-reduce of per-class feature sums (410 KB); synthetic clips "
                                       "class-owned, no gradient exchange" % world) if trainer.__dict__.get("shard") == "batch" else
                                      "class-sharded x%d (owner-computes, no gradient exchange)" % world,
                       "pool_per_class": args.pool_per_class,
                       "real_pool": "resident in HBM: fp32 clips + the same clips converted once to the first layer's 16-bit pixel "
                                    "rows; a real batch is an index list (get_images + cast of the reference), no per-step "
                                    "conversion" if backend.resident_rows else "resident in HBM as fp32, converted per step"},
            "loss_last": float(losses[-1]) / args.classes,
            "step_tflops": step_flop / (dt / args.steps) / 1e12,
            "step_frac_of_mfma_peak": step_flop / (dt / args.steps) / 2.5e15,
        }
        # roofline of the dominant kernel: fwd conv layer 1 over the real clips
        if prof:
            times = [a.elapsed_time(b) * 1e-3 for (name, n, a, b) in prof if name == "fwd1"]
            clips = [n for (name, n, a, b) in prof if name == "fwd1"]
            flop_per_launch = 2.0 * macs[1] * float(np.mean(clips))
            achieved = flop_per_launch / float(np.mean(times)) / 1e12
            out["roofline"] = {"bound": "mfma", "kernel": "conv_mfma_kernel<%s, 3, false, 2, 1> = PREC %s, balanced 7-tile layout (conv layer 1 fwd, real clips)" % (
                                   {"bf16": 0, "f16": 1, "bf16x3": 2, "f16x3": 3}[args.prec_real], args.prec_real)
                               if not args.prec_real.endswith("x3") else "conv_mfma_kernel<%s, 7> (conv layer 1 fwd, real clips)" % args.prec_real,
                               "achieved": achieved, "peak": 2500.0, "unit": "TFLOP/s", "frac": achieved / 2500.0,
                               "traffic": pmc_traffic(float(np.mean(clips))), "launches": len(times), "mean_launch_ms": float(np.mean(times)) * 1e3,
                               "flop_per_launch": flop_per_launch}
            for lname, li in (("fwd0", 0), ("fwd2", 2)):
                tt = [a.elapsed_time(b) * 1e-3 for (name, n, a, b) in prof if name == lname]
                cc = [n for (name, n, a, b) in prof if name == lname]
                if tt:
                    out["roofline"][lname + "_tflops"] = 2.0 * macs[li] * float(np.mean(cc)) / float(np.mean(tt)) / 1e12
        if args.eval_epochs > 0 and args.method == "dm" and world == 1:    # (single process: gather_syn is a collective)
            out["eval"] = run_eval(args, trainer, pool, device)
        if world == 1 and not args.no_cpu_baseline and args.method == "dm":
            out["cpu_baseline"] = cpu_baseline(args, trainer, backend, args.warmup + args.steps, geo)
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def run_eval(args, trainer, pool, device):
    """evaluate_synset (utils.py:848-886) on the current synthetic clips: a fresh ConvNet3D trained for
    --eval-epochs epochs with the HIP train step, tested (3 passes, HIP inference) on 4 held-out pool clips per
    class.  The pool is synthetic noise, so the accuracy is chance level by construction; the timing is the point."""
    import types
    from video_distillation_amd import utils
    C = args.classes
    syn = trainer.gather_syn().detach().clone()
    labels = torch.arange(C, device=device).repeat_interleave(args.ipc)
    idx = torch.as_tensor([pool.offsets[c] + pool.counts[c] - 1 - k for c in range(C) for k in range(4)], device=device)
    test = utils.TensorDataset(pool.clips[idx], torch.arange(C, device=device).repeat_interleave(4))
    loader = torch.utils.data.DataLoader(test, batch_size=64, shuffle=False)
    eargs = types.SimpleNamespace(device=device, lr_net=0.01, epoch_eval_train=args.eval_epochs, batch_train=256,
                                  model="ConvNet3D", eval_mode="SS")
    net = utils.get_network("ConvNet3D", 3, C, (args.size, args.size), frames=args.frames, dist=False).to(device)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _, acc_train, acc_test, _ = utils.evaluate_synset(0, net, syn, labels, loader, eargs, mode="none")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"epochs": args.eval_epochs + 1, "seconds": dt, "ms_per_epoch": dt / (args.eval_epochs + 1) * 1e3,
            "acc_train": float(acc_train), "acc_test": float(acc_test), "test_clips": int(idx.numel()),
            "note": "synthetic noise pool: test accuracy is chance by construction"}


def bench_dc(args, distill, geo, pool, device, rank, world):
    """Secondary line: gradient matching, one step = one `for it` iteration of
    distill.GMTrainer with get_loops(ipc) (ipc 1 / 5 -> outer 1, inner 1: no network update inside)."""
    from video_distillation_amd import utils
    outer, inner = utils.get_loops(args.ipc)
    ops = distill.HipGMOps(device, args.dis_metric)
    trainer = distill.GMTrainer(ops, pool, geo, args.classes, args.ipc, args.batch_real, lr_img=0.1, rank=rank, world=world,
                                outer_loop=outer, inner_loop=inner)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()
    for it in range(args.warmup):
        trainer.step(it)
    barrier()
    t0 = time.perf_counter()
    losses = [trainer.global_loss(trainer.step(it)) for it in range(args.warmup, args.warmup + args.steps)]
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
    if rank == 0:
        from video_distillation_amd import networks
        print(json.dumps({
            "metric": "distillation steps/sec (DC gradient matching '%s', IPC=%d)" % (args.dis_metric, args.ipc),
            "value": args.steps / dt, "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": networks.get_precision()["match"], "data": "synthetic",
            "config": {"workload": "gradient matching: C=%d classes x (%d real + %d syn) clips %dx%dx%d, fresh net per step, "
                                   "outer_loop %d inner_loop %d" % (args.classes, args.batch_real, args.ipc, args.size, args.size,
                                                                    args.frames, outer, inner),
                       "precision": networks.get_precision(),
                       "parallelism": "class-sharded x%d (owner-computes, no gradient exchange)" % world},
            "loss_last": float(losses[-1]) / args.classes}))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def bench_mtt(args, distill, geo, pool, device, rank, world):
    """Secondary line: one MTT iteration = syn_steps unrolled student steps on
    batch_syn synthetic clips + the reverse sweep to the pixels and syn_lr; expert buffer = random-walk
    parameter lists (11 epochs) shared by all ranks; every step's batch is split over the ranks."""
    from video_distillation_amd import networks
    C = args.classes
    gen = torch.Generator(device=device); gen.manual_seed(99)
    image_syn = torch.randn(C * args.ipc, args.frames, 3, args.size, args.size, device=device, generator=gen)
    label_syn = torch.arange(C, device=device).repeat_interleave(args.ipc)
    traj = [distill.fresh_full_network(5, C, device)]
    for e in range(11):
        traj.append([p + 0.01 * p.abs().mean() * torch.randn(p.shape, device=device, generator=gen) for p in traj[-1]])
    ops = distill.HipMTTOps(geo, C, device, dropout_p=0.5, batch_hint=min(args.batch_syn, C * args.ipc) // max(world, 1))
    tr = distill.MTTTrainer(ops, C, image_syn, label_syn, syn_lr=0.01, lr_img=1.0, lr_lr=1e-6, syn_steps=args.syn_steps,
                            batch_syn=min(args.batch_syn, C * args.ipc), expert_epochs=1, max_start_epoch=10, rank=rank, world=world)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()
    for it in range(args.warmup):
        tr.step(it, traj)
    barrier()
    t0 = time.perf_counter()
    losses = [tr.step(it, traj) for it in range(args.warmup, args.warmup + args.steps)]
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
    if rank == 0:
        print(json.dumps({
            "metric": "distillation steps/sec (MTT, syn_steps=%d)" % args.syn_steps,
            "value": args.steps / dt, "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": networks.get_precision()["match"], "data": "synthetic",
            "config": {"workload": "trajectory matching: C=%d classes, %d synthetic clips %dx%dx%d, syn_steps %d x batch_syn %d, "
                                   "expert_epochs 1, synthetic random-walk expert buffer" % (
                                       C, C * args.ipc, args.size, args.size, args.frames, args.syn_steps, tr.batch_syn),
                       "precision": networks.get_precision(),
                       "parallelism": "student batch split x%d, all-reduce of flat gradient + Hessian-vector product per inner step" % world},
            "grand_loss_last": float(losses[-1]), "syn_lr": tr.syn_lr}))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
