"""Late-regime parity of the SHIPPED precision mode against the ORACLE (no HIP-vs-HIP link in the chain).

The regime: class-patterned real clips (a base clip per class + 10 % noise) and synthetic clips initialised from a real
one, so that |mean f_real - mean f_syn| is only a few per cent of |f| -- where an absolute error on the real side's class
mean weighs most on the DM gradient 2 (mean f_syn - mean f_real) J (distill_baseline.py:344-355).  The HIP trainer runs
free in the mode bench.py times (real clips single-pass f16 with dithered weights, synthetic clips f16 hi+lo pairs, input
gradient per ``MODES``); at EVERY step the CPU oracle (``oracle.ref_cpu.dm_loss_and_grad``) is evaluated on the very state
the HIP step starts from -- same synthetic clips, same fresh network, same real batch -- in fp32 (what the reference
computes) and in fp64 (what it approximates), and the HIP loss and pixel gradient are compared with both.

Gradient criterion (flip tolerant): a max-pool window whose two largest entries tie to within rounding routes its gradient
elsewhere depending on summation order -- the fp32 oracle differs from the fp64 one for exactly that reason (`oracle32_vs_64`
below), and in this regime it happens in roughly one step out of five.  So every (step, class) entry is classified by the
pooling decisions the HIP forward of that synthetic clip RECORDED (its arg-max bytes) against the fp64 oracle's max_pool3d
decisions (tests/argmax_tools.py): an entry with no differing window at any level is CLEAN and must meet the bar at that very
step; an entry above the bar must show a differing window, every differing window must be a near-tie of the fp64 oracle
itself (margin below 2e-5 of the level's rms), and the error stays below the 5e-2 a last-level flip can cause (a first-level
window carrying a large gradient moves up to 1e-2).  (At 112x112x16 a clip has 800 k first-level windows and most steps have
one or two such ties; the bar binds on the entries that have none -- measured 0.74e-3 -- and the median over all entries is
recorded, not asserted: 1.1e-3 there, 0.73e-3 at 64x64x8 where 27 of 32 entries are clean.)

Round 5: the same bar on FIVE seeds per geometry and on 2 and 4 classes (``test_late_regime_seeds``): every seed draws other
class templates, noise, networks and real batches; every clean entry of every seed must meet the bar, and the maximum over
seeds is recorded.  ``VD_PARITY_FULL=1`` runs the full cross product with the step counts of the single-seed tests (the
record in profiles/r05_parity.json); the default sizes keep the driver's suite short.

Round 6 (suite time): the HIP trainer runs all its steps FIRST, recording the state every step started from, its loss, gradient
and arg-max bytes; the oracle evaluations of the recorded steps then run on a pool of host threads (four at a time, 32 cores
each: they are independent once the states are recorded), the fp32 oracle only where ``fp32=True`` -- the bar is stated against
fp64 -- i.e. in the two reference cases per geometry that replace round 5's single-seed tests (seed 1201 / C 2 at 64x64x8 with
16 steps and the all-hi+lo trainer beside the shipped one; seed 12 / C 2 at 112x112x16 with 5 steps).

Measured values go to gpurun_out/r06_parity.json (copied to profiles/)."""
import json
import os
import time

import numpy as np
import pytest
import torch

from oracle import ref_cpu as R
from tests import argmax_tools

pytestmark = pytest.mark.gpu

MODES = {"shipped": dict(prec_real="f16", prec_syn="f16x3", prec_bwd=None),            # HipBackend's defaults = bench.py's
         "x3": dict(prec_real="f16x3", prec_syn="f16x3", prec_bwd="f16x3")}


def _record(key, value):
    path = os.environ.get("VD_PARITY_LOG", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                        "gpurun_out", "r06_parity.json"))
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        data = json.load(open(path)) if os.path.exists(path) else {}
        data[key] = value
        json.dump(data, open(path, "w"), indent=1, sort_keys=True)
    except OSError:
        pass


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm())


def _oracle(params, reals, syn, dtype):
    """(loss, d loss / d syn, seconds) of the CPU oracle in ``dtype``.  (All 256 logical CPUs of the GPU box's host are 10x
    slower than 32 threads for these convolutions -- bench.py's calibration -- so the thread count is capped.)"""
    old = torch.get_num_threads()
    torch.set_num_threads(ORACLE_THREADS)
    try:
        t0 = time.perf_counter()
        loss, grad = R.dm_loss_and_grad([p.to(dtype) for p in params], [r.to(dtype) for r in reals], syn.to(dtype), ipc=1)
        return float(loss), grad, time.perf_counter() - t0
    finally:
        torch.set_num_threads(old)


# oracle evaluations in flight x threads each: 8 x 16 on the GPU boxes' 2 x 64-core hosts (3 TB of RAM; a recorded 112x112x16 step
# needs ~5 GB in fp64) -- the fp64 convolutions scale better over independent evaluations than over threads of one
ORACLE_WORKERS = max(1, min(8, (os.cpu_count() or 1) // 16))
ORACLE_THREADS = max(1, min(32, (os.cpu_count() or 1) // (2 * ORACLE_WORKERS)))


def _evaluate_step(st, C, fp32):
    """Everything the CPU does for one recorded step: the oracle in fp64 (and fp32), the feature gap, the decisions of the lead
    trainer's synthetic forward against the fp64 oracle's.  Runs on a worker thread (its own OpenMP team)."""
    torch.set_num_threads(ORACLE_THREADS)
    weights, reals, syn = st["weights"], st["reals"], st["syn"]
    out = {}
    out["l64"], out["g64"], out["t64"] = _oracle(weights, reals, syn, torch.float64)
    if fp32:
        out["l32"], out["g32"], out["t32"] = _oracle(weights, reals, syn, torch.float32)
    with torch.no_grad():      # how small the quantity the gradient is proportional to is, relative to the features
        gaps = []
        for c in range(C):
            fr = R.convnet3d_embed(reals[c], weights).mean(0)
            fs = R.convnet3d_embed(syn[c:c + 1], weights)[0]
            gaps.append(float((fr - fs).norm() / fr.norm()))
    out["gaps"] = gaps
    out["dec"] = argmax_tools.compare_decisions(syn, weights, st["am"])
    return out


def late_regime_run(geom, C, NP, B, steps, lr, seed, modes=("shipped",), backend_kw=None, noise=0.1, fp32=True):
    """-> record dict.  The first mode in ``modes`` runs free; the others are put on its state before every step.  ``fp32``:
    also evaluate the fp32 oracle (loss_vs_fp32, oracle32_vs_64); without it those entries stay empty."""
    from concurrent.futures import ThreadPoolExecutor
    ctx = late_regime_hip(geom, C, NP, B, steps, lr, seed, modes, backend_kw, noise)
    old_threads = torch.get_num_threads()
    try:
        with ThreadPoolExecutor(max_workers=ORACLE_WORKERS) as ex:
            evaluated = list(ex.map(lambda st: _evaluate_step(st, C, fp32), ctx["recorded"]))
    finally:
        torch.set_num_threads(old_threads)
    return late_regime_finish(ctx, evaluated, fp32)


def late_regime_hip(geom, C, NP, B, steps, lr, seed, modes=("shipped",), backend_kw=None, noise=0.1):
    """Phase A: the HIP trainers, step by step, recording what the oracle will need (state, network, real batch, loss, gradient,
    arg-max bytes of the lead trainer's synthetic forward).  -> context for ``late_regime_finish``."""
    from video_distillation_amd import distill, plan
    T, H, W = geom
    geo = plan.NetGeometry(T, H, W)
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(seed)
    base = torch.randn(C, 1, T, 3, H, W, device=dev, generator=g)
    clips = (base + noise * torch.randn(C, NP, T, 3, H, W, device=dev, generator=g)).reshape(C * NP, T, 3, H, W)
    pool = distill.RealPool(clips, [NP] * C, [c * NP for c in range(C)])
    syn0 = clips[::NP].clone()
    mu = 0.5
    trainers = {}
    for m in modes:
        kw = {k: v for k, v in MODES[m].items() if v is not None}
        kw.update(backend_kw or {})
        if m == "shipped" and os.environ.get("VD_PARITY_BWD"):      # (experiments: another input-gradient format for the lead trainer)
            kw["prec_bwd"] = os.environ["VD_PARITY_BWD"]
        be = distill.HipBackend(geo, dev, chunk=4096, **kw)
        trainers[m] = distill.DMTrainer(be, pool, C, 1, B, lr_img=lr, momentum=mu, image_syn=syn0.clone())
    lead = trainers[modes[0]]
    captured = {}
    orig_embed_syn = lead.be.embed_syn

    def spy_embed_syn(x, weights):          # the pooling decisions of the lead trainer's synthetic-clip forward
        f, handle = orig_embed_syn(x, weights)
        captured["am"] = [a.clone() for a in handle[0][2:5]]
        return f, handle
    lead.be.embed_syn = spy_embed_syn
    recorded = []
    for it in range(steps):
        state = (lead.image_syn.clone(), lead.buf.clone(), lead.steps_done)
        weights = [w.cpu() for w in lead.be.new_network(seed=it)]
        idx = distill.sample_real_indices(it, pool.counts, pool.offsets, B, list(range(C)))
        real = clips[torch.as_tensor(idx, device=dev)].cpu()
        st = {"weights": weights, "reals": [real[c * B:(c + 1) * B] for c in range(C)], "syn": state[0].cpu(), "hip": {}}
        for m in modes:
            tr = trainers[m]
            if tr is not lead:
                tr.image_syn.copy_(state[0]); tr.buf.copy_(state[1]); tr.steps_done = state[2]
            lt = float(tr.step(it))
            if hasattr(tr, "sync"):
                tr.sync()
            gt = (tr.buf - mu * state[1] if it > 0 else tr.buf.clone()).cpu()           # buf = mu * buf + g
            st["hip"][m] = (lt, gt)
            if tr is lead:
                st["am"] = [a.cpu() for a in captured["am"]]
        recorded.append(st)
    summ = {m: {"dither_groups": int(getattr(trainers[m].be, "_dither", 0)), "real_last": getattr(trainers[m].be, "real_last", None),
                "prec_bwd": trainers[m].be.eng_syn.prec_name if trainers[m].be.eng_syn.prec_bwd == trainers[m].be.eng_syn.prec
                else [k for k, v in trainers[m].be.hip.PREC.items() if v == trainers[m].be.eng_syn.prec_bwd][0]} for m in modes}
    return {"recorded": recorded, "modes": modes, "C": C, "summ": summ,
            "config": "C=%d classes x (%d real + 1 syn) clips %dx%dx%d, pool %d per class (base + %.0f %% noise), %d steps, lr_img %g" % (
                C, B, H, W, T, NP, noise * 100, steps, lr)}


def late_regime_finish(ctx, evaluated, fp32):
    """Phase B's bookkeeping: the record dict from the recorded HIP steps and their oracle evaluations (``_evaluate_step``)."""
    recorded, modes, C = ctx["recorded"], ctx["modes"], ctx["C"]
    rec = {m: {"loss_vs_fp32": [], "loss_vs_fp64": [], "grad_vs_fp32": [], "grad_vs_fp64": [], "grad_vs_fp64_per_class": [],
               "flipped_frac": []} for m in modes}
    rec["decisions"] = {"mismatch_per_class": [], "not_near_tie_per_class": [], "mismatch_per_level": []}
    rec.update({"oracle32_vs_64": {"loss": [], "grad": [], "grad_per_class": []}, "feature_gap_over_norm": [], "oracle_seconds": []})
    for st, ev in zip(recorded, evaluated):
        l64, g64 = ev["l64"], ev["g64"]
        rec["oracle_seconds"].append([ev.get("t32", 0.0), ev["t64"]])
        rec["feature_gap_over_norm"].append(ev["gaps"])
        if fp32:
            l32, g32 = ev["l32"], ev["g32"]
            rec["oracle32_vs_64"]["loss"].append(abs(l32 / l64 - 1))
            rec["oracle32_vs_64"]["grad"].append(_rel(g32, g64))
            rec["oracle32_vs_64"]["grad_per_class"].append([_rel(g32[c], g64[c]) for c in range(C)])
        for m in modes:
            lt, gt = st["hip"][m]
            r = rec[m]
            r["loss_vs_fp64"].append(abs(lt / l64 - 1))
            r["grad_vs_fp64"].append(_rel(gt, g64))
            r["grad_vs_fp64_per_class"].append([_rel(gt[c], g64[c]) for c in range(C)])
            if fp32:
                r["loss_vs_fp32"].append(abs(lt / l32 - 1)); r["grad_vs_fp32"].append(_rel(gt, g32))
            d = (gt.double() - g64).abs()
            r["flipped_frac"].append(float((d > 1e-2 * g64.abs().max()).double().mean()))
        dec = ev["dec"]
        rec["decisions"]["mismatch_per_class"].append([sum(v) for v in zip(*[d_["mismatch_per_clip"] for d_ in dec])])
        rec["decisions"]["not_near_tie_per_class"].append([sum(v) for v in zip(*[d_["not_near_tie_per_clip"] for d_ in dec])])
        rec["decisions"]["mismatch_per_level"].append([d_["mismatch"] for d_ in dec])
    flipped = np.asarray(rec["decisions"]["mismatch_per_class"]).reshape(-1) > 0
    for m in modes:
        per = np.asarray(rec[m]["grad_vs_fp64_per_class"]).reshape(-1)
        rec[m]["summary_clean"] = {"entries": int((~flipped).sum()), "of": int(flipped.size),
                                   "grad_vs_fp64_median": float(np.median(per[~flipped])) if (~flipped).any() else None,
                                   "grad_vs_fp64_max": float(per[~flipped].max()) if (~flipped).any() else None}
        rec[m]["summary"] = {"loss_vs_fp32_max": max(rec[m]["loss_vs_fp32"]) if fp32 else None, "loss_vs_fp64_max": max(rec[m]["loss_vs_fp64"]),
                             "grad_vs_fp64_median": float(np.median(per)), "grad_vs_fp64_p90": float(np.quantile(per, 0.9)),
                             "grad_vs_fp64_max": float(per.max()),
                             "grad_vs_fp32_median": float(np.median(rec[m]["grad_vs_fp32"])) if fp32 else None, **ctx["summ"][m]}
    if fp32:
        per = np.asarray(rec["oracle32_vs_64"]["grad_per_class"]).reshape(-1)
        rec["oracle32_vs_64"]["summary"] = {"loss_max": max(rec["oracle32_vs_64"]["loss"]), "grad_median": float(np.median(per)),
                                            "grad_p90": float(np.quantile(per, 0.9)), "grad_max": float(per.max())}
    else:
        rec["oracle32_vs_64"]["summary"] = None
    rec["config"] = ctx["config"]
    return rec


def _report(name, rec, modes):
    print(name, rec["config"])
    print("  feature gap / |f| per step (mean over classes):", ["%.3f" % float(np.mean(v)) for v in rec["feature_gap_over_norm"]])
    print("  fp32 oracle vs fp64 oracle:", rec["oracle32_vs_64"]["summary"])
    for m in modes:
        print("  %-8s vs oracle:" % m, rec[m]["summary"])
        print("  %-8s clean entries (pooling decisions equal to the fp64 oracle's at every level):" % m, rec[m]["summary_clean"])
    up = np.asarray(rec["decisions"]["mismatch_per_class"])
    print("  (step, class) entries with a differing near-tie window: %d of %d; mismatches per step and level: %s" % (
        int((up > 0).sum()), up.size, rec["decisions"]["mismatch_per_level"]))
    print("  oracle seconds per step (fp32, fp64): %.1f %.1f" % tuple(np.mean(rec["oracle_seconds"], axis=0)))


# bars of the shipped mode: loss 1e-3 (north_star); pixel gradient per synthetic clip vs the fp64 oracle 1e-3
GRAD_BAR = float(os.environ.get("VD_PARITY_GRAD_BAR", "1e-3"))


def _assert_shipped(rec):
    s = rec["shipped"]["summary"]
    # what bench.py times: the last level in hi+lo pairs -- with fp8 corrections ("c8") where the geometry has the one-clip program
    assert s["dither_groups"] == 8 and s["real_last"] in ("x3", "c8") and s["prec_bwd"] == "f16x3"
    assert (s["loss_vs_fp32_max"] is None or s["loss_vs_fp32_max"] < 1e-3) and s["loss_vs_fp64_max"] < 1e-3
    clean = rec["shipped"]["summary_clean"]
    assert clean["entries"] >= 2 and clean["grad_vs_fp64_median"] < GRAD_BAR, clean
    per = np.asarray(rec["shipped"]["grad_vs_fp64_per_class"])
    upper = np.asarray(rec["decisions"]["mismatch_per_class"])
    far = np.asarray(rec["decisions"]["not_near_tie_per_class"])
    assert int(far.sum()) == 0, "a pooling decision differs from the fp64 oracle's in a window that is no near-tie: %s" % far.tolist()
    for it in range(per.shape[0]):
        for c in range(per.shape[1]):
            if upper[it, c] == 0:          # same routing as the fp64 oracle at every level: the bar holds at this very step
                assert per[it, c] < GRAD_BAR, (it, c, per[it, c])
            else:                          # a near-tie routed the other way (the fp32 oracle does the same, oracle32_vs_64)
                assert per[it, c] < 5e-2, (it, c, per[it, c], int(upper[it, c]))


# ---- round 5: the bar on five seeds per geometry, 2 and 4 classes -------------------------------------------------------
FULL = os.environ.get("VD_PARITY_FULL") == "1"
SEEDS_64 = (1201, 7, 23, 101, 4242)
SEEDS_112 = (12, 5, 31, 77, 2026)
_SEED_CASES = ([("64", s, C) for s in SEEDS_64 for C in (2, 4)]
               + [("112", s, 2) for s in SEEDS_112]
               + [("112", s, 4) for s in (SEEDS_112 if FULL else SEEDS_112[:1])])
_seen = {}


def _assert_seed(rec):
    """The per-entry part of ``_assert_shipped`` (a short run may have few clean entries; the aggregate test counts them)."""
    s = rec["shipped"]["summary"]
    assert s["dither_groups"] == 8 and s["real_last"] in ("x3", "c8") and s["prec_bwd"] == "f16x3"
    assert (s["loss_vs_fp32_max"] is None or s["loss_vs_fp32_max"] < 1e-3) and s["loss_vs_fp64_max"] < 1e-3
    per = np.asarray(rec["shipped"]["grad_vs_fp64_per_class"])
    upper = np.asarray(rec["decisions"]["mismatch_per_class"])
    far = np.asarray(rec["decisions"]["not_near_tie_per_class"])
    assert int(far.sum()) == 0, "a pooling decision differs from the fp64 oracle's in a window that is no near-tie: %s" % far.tolist()
    for it in range(per.shape[0]):
        for c in range(per.shape[1]):
            if upper[it, c] == 0:
                assert per[it, c] < GRAD_BAR, (it, c, per[it, c])
            else:
                assert per[it, c] < 5e-2, (it, c, per[it, c], int(upper[it, c]))


REFERENCE_CASES = {("64", 1201, 2): 16, ("112", 12, 2): 5}      # the single-seed runs of rounds 3 - 5: full step counts, fp32 oracle too


def _case_setup(geom, seed, C):
    ref = (geom, seed, C) in REFERENCE_CASES
    if geom == "64":
        steps = REFERENCE_CASES.get((geom, seed, C), (16 if FULL else 4) if C == 2 else (8 if FULL else 2))
        return dict(geom=(8, 64, 64), C=C, NP=80, B=64, steps=steps, lr=50.0, seed=seed, modes=("shipped", "x3") if ref else ("shipped",)), ref
    steps = REFERENCE_CASES.get((geom, seed, C), (5 if FULL else 1) if C == 2 else (2 if FULL else 1))
    return dict(geom=(16, 112, 112), C=C, NP=72, B=64, steps=steps, lr=20.0, seed=seed, modes=("shipped",)), ref


_ALL = {}


def _all_cases():
    """Every case of the seed matrix in ONE pass (round 6): the HIP phases one after the other, then all recorded steps of all
    cases through one pool of oracle workers -- a one-step case alone keeps one worker busy, the matrix keeps all of them busy.
    Computed by the first test that asks; ``VD_PARITY_CASES=64-seed7-C2,...`` restricts the pass (single cases by hand)."""
    if _ALL:
        return _ALL
    from concurrent.futures import ThreadPoolExecutor
    only = os.environ.get("VD_PARITY_CASES")
    cases = [c for c in _SEED_CASES if not only or ("%s-seed%d-C%d" % c) in only.split(",")]
    ctxs, jobs = {}, []
    t0 = time.perf_counter()
    for case in cases:
        kw, ref = _case_setup(*case)
        ctxs[case] = (late_regime_hip(**kw), ref, kw)
        jobs += [(case, k) for k in range(len(ctxs[case][0]["recorded"]))]
    t1 = time.perf_counter()
    # longest evaluations first (112x112x16 steps take ~4x a 64x64x8 step; reference cases also run the fp32 oracle)
    jobs.sort(key=lambda j: -(ctxs[j[0]][2]["geom"][1] ** 2 * ctxs[j[0]][2]["geom"][0] * ctxs[j[0]][2]["C"] * (1.5 if ctxs[j[0]][1] else 1.0)))
    old_threads = torch.get_num_threads()
    try:
        with ThreadPoolExecutor(max_workers=ORACLE_WORKERS) as ex:
            futs = {j: ex.submit(_evaluate_step, ctxs[j[0]][0]["recorded"][j[1]], ctxs[j[0]][2]["C"], ctxs[j[0]][1]) for j in jobs}
            done = {j: f.result() for j, f in futs.items()}
    finally:
        torch.set_num_threads(old_threads)
    for case in cases:
        ctx, ref, kw = ctxs[case]
        _ALL[case] = (late_regime_finish(ctx, [done[(case, k)] for k in range(len(ctx["recorded"]))], ref), ref, kw)
        ctx["recorded"] = None
    print("late-regime matrix: %d cases, %d recorded steps; HIP phases %.0f s, oracle pool (%d workers) %.0f s" % (
        len(cases), len(jobs), t1 - t0, ORACLE_WORKERS, time.perf_counter() - t1))
    return _ALL


@pytest.mark.parametrize("geom,seed,C", _SEED_CASES, ids=["%s-seed%d-C%d" % c for c in _SEED_CASES])
def test_late_regime_seeds(geom, seed, C):
    """distill_baseline.py:344-355 in the shipped mode, another seed / class count: every clean entry within 1e-3 of the fp64
    oracle's pixel gradient, every loss within 1e-3 of the fp64 oracle's (and of the fp32 oracle's in the two reference cases,
    which also run the G12 step counts; the 64x64x8 one with the all-hi+lo trainer beside the shipped one)."""
    allc = _all_cases()
    if (geom, seed, C) not in allc:
        pytest.skip("not in VD_PARITY_CASES")
    rec, ref, kw = allc[(geom, seed, C)]
    steps, modes = kw["steps"], kw["modes"]
    if ref:
        _report("late regime %s (reference case)" % geom, rec, modes)
        _record("late_64x64x8" if geom == "64" else "late_112x112x16", rec)
        _assert_shipped(rec)
        if geom == "64":
            assert rec["shipped"]["summary_clean"]["entries"] >= rec["shipped"]["summary_clean"]["of"] // 2     # the per-step bar is not vacuous
            assert rec["x3"]["summary"]["grad_vs_fp64_median"] < 1e-4
    clean = rec["shipped"]["summary_clean"]
    print("late regime %s seed %d C %d:" % (geom, seed, C), clean, "loss max", rec["shipped"]["summary"]["loss_vs_fp64_max"])
    _seen[(geom, seed, C)] = {"clean": clean, "loss_vs_fp32_max": rec["shipped"]["summary"]["loss_vs_fp32_max"],
                              "loss_vs_fp64_max": rec["shipped"]["summary"]["loss_vs_fp64_max"],
                              "grad_vs_fp64_median_all": rec["shipped"]["summary"]["grad_vs_fp64_median"],
                              "grad_vs_fp32_median": rec["shipped"]["summary"]["grad_vs_fp32_median"],
                              "feature_gap_over_norm_mean": float(np.mean(rec["feature_gap_over_norm"])), "steps": steps}
    _record("seeds_%s_seed%d_C%d" % (geom, seed, C), rec)
    _assert_seed(rec)


def test_late_regime_seeds_aggregate():
    """Max over seeds of the clean-entry gradient error per geometry (the number DESIGN quotes), and that the per-entry bar
    was not vacuous: at least five clean entries per geometry over the seeds that ran."""
    if not _seen:
        pytest.skip("runs after test_late_regime_seeds in the same session")
    table = {}
    for geom in ("64", "112"):
        rows = {k: v for k, v in _seen.items() if k[0] == geom}
        if not rows:
            continue
        mx = [v["clean"]["grad_vs_fp64_max"] for v in rows.values() if v["clean"]["grad_vs_fp64_max"] is not None]
        med = [v["clean"]["grad_vs_fp64_median"] for v in rows.values() if v["clean"]["grad_vs_fp64_median"] is not None]
        n_clean = sum(v["clean"]["entries"] for v in rows.values())
        table[geom] = {"runs": len(rows), "clean_entries": n_clean, "entries": sum(v["clean"]["of"] for v in rows.values()),
                       "clean_grad_vs_fp64_max_over_seeds": max(mx) if mx else None,
                       "clean_grad_vs_fp64_median_of_medians": float(np.median(med)) if med else None,
                       "loss_vs_fp64_max_over_seeds": max(v["loss_vs_fp64_max"] for v in rows.values()),
                       "per_run": {"seed%d_C%d" % (k[1], k[2]): v for k, v in sorted(rows.items())}}
        assert n_clean >= 5, table[geom]
        assert max(mx) < GRAD_BAR
        # (round 6) with the weights of the hi+lo programs packed x 2^8 the synthetic forward's decisions equal the fp64 oracle's on every
        # entry of the record (126 of 126; round 5: 356 of 410): at least four fifths must stay clean, or the bar above binds on too few
        assert n_clean >= 0.8 * table[geom]["entries"], table[geom]
    print(json.dumps({g: {k: v for k, v in t.items() if k != "per_run"} for g, t in table.items()}, indent=1))
    _record("seeds_summary", table)
