"""The outer distillation loop of the DM hot path, MI355X-first.

What the reference does per iteration (distill_baseline.py:334-355; s2d twin
distill_s2d_ms.py:393-438): draw a fresh random ConvNet3D, and for every class embed 64
real clips (copied host->device per class) and the class's synthetic clips, sum the squared
distances of the mean embeddings, back-propagate to the synthetic pixels and take an
SGD-momentum step.

What this module does with the same arithmetic:
  * the real pool stays resident in HBM; a class batch is an index list and the gather is
    fused into the first kernel (no per-class host->device copy);
  * all classes of a rank go through each layer in a few large launches instead of 2*C
    small ones (the network is the same for every class of an iteration);
  * real clips (no gradient, 98.5 % of the FLOPs) use single-pass fp16 MFMA operands, the
    synthetic clips (which carry the gradient) the split-precision path;
  * classes are sharded over ranks in contiguous blocks; a rank owns its classes' synthetic
    clips and momentum, so the pixel gradients never cross xGMI (owner-computes).  Per step
    the only collective is a 4-byte all-reduce of the loss for logging; the s2d variant
    all-reduces the 327 hallucinator gradients.  Synthetic clips are all-gathered only for
    evaluation / saving.

The compute backend is injected (``HipBackend`` in production).  Tests drive the same
trainer logic with a CPU stand-in on ``gloo`` to check the sharding.
"""
from __future__ import annotations

import ctypes
import math
import collections
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import plan as P

PARAM_SHAPES = ((64, 3, 3, 7, 7), (64,), (128, 64, 3, 7, 7), (128,), (128, 128, 3, 7, 7), (128,))


COLLECTIVE_CALLS = {"all_reduce": 0, "all_gather": 0, "bytes": 0, "control_all_reduce": 0}      # issued by the trainers of this process (bench.py reports them; control: 4-byte decisions, not data)


def _all_reduce(t: torch.Tensor) -> None:
    """Sum-all-reduce of ``t`` in place over the default process group, issued on the CURRENT stream (RCCL under the ``nccl``
    backend: torch's process group runs the collective on its own communication stream, which waits for the current stream's
    work before it starts and which the current stream waits for afterwards)."""
    import torch.distributed as dist
    COLLECTIVE_CALLS["all_reduce"] += 1
    COLLECTIVE_CALLS["bytes"] += t.numel() * t.element_size()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)


def _all_reduce_max(t: torch.Tensor) -> None:
    """Max-all-reduce of a small decision flag (control path: counted apart from the data-path collectives)."""
    import torch.distributed as dist
    COLLECTIVE_CALLS["control_all_reduce"] += 1
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)


def _all_gather(parts: List[torch.Tensor], t: torch.Tensor) -> None:
    import torch.distributed as dist
    COLLECTIVE_CALLS["all_gather"] += 1
    COLLECTIVE_CALLS["bytes"] += t.numel() * t.element_size() * len(parts)
    dist.all_gather(parts, t)


def collectives_on(world: int) -> bool:
    """Whether a trainer issues its data-path collectives: always with more than one rank; on ONE rank only when
    ``VD_FORCE_COLLECTIVES=1`` and a process group exists -- so that the RCCL all_reduce / all_gather calls, on the trainers'
    own streams, execute on a one-GPU box (tests/test_gpu_collectives.py; a 1-rank collective is the identity)."""
    if world > 1:
        return True
    if os.environ.get("VD_FORCE_COLLECTIVES") == "1":
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized()
    return False


def class_range(num_classes: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of classes owned by ``rank`` (sizes differ by at most one)."""
    base, extra = divmod(num_classes, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def hybrid_partition(num_classes: int, rank: int, world: int) -> Tuple[List[int], List[int], List[int]]:
    """The hybrid decomposition of the DM class terms (distill_baseline.py:343-351) over ``world`` ranks: every rank takes
    ``num_classes // world`` WHOLE classes (a contiguous block, no exchange), and the ``num_classes % world`` classes left
    over are SPLIT: each rank embeds 1 / world of their real batches and the per-class feature sums are all-reduced
    (2048 floats per split class), their synthetic clips being owned round-robin.  50 classes on 8 ranks: 6 whole classes +
    1/8 of 2 = 400 real clips on every rank, where whole-class blocks give 7,7,6,... (448 / 384).
    -> (this rank's block, all split classes, the split classes this rank owns)."""
    nb = num_classes // world
    if world == 1 and os.environ.get("VD_HYBRID_FORCE_SPLIT"):     # test knob: split classes on ONE rank, so that the exchange executes
        nb = max(0, num_classes - int(os.environ["VD_HYBRID_FORCE_SPLIT"]))
    split = list(range(nb * world, num_classes))
    return list(range(rank * nb, (rank + 1) * nb)), split, [c for i, c in enumerate(split) if i % world == rank]


def choose_shard(num_classes: int, batch_real: int, world: int, method: str = "dm") -> str:
    """The decomposition ``bench.py --shard auto`` / ``run_dm`` pick for ``world`` ranks: whole-class blocks need no data-path
    collective but leave ranks idle when ``num_classes % world`` is large against ``num_classes // world`` (50 over 8: blocks
    of 7,7,6,..., ceiling 50/7 = 7.14x); the hybrid keeps whole classes and splits only the left-over ones' real batches over all
    ranks (one all-reduce of 2048 floats per split class).  The hybrid is chosen where the class blocks are more than 8 % uneven
    AND the real batch divides by the number of ranks (``batch_real % world == 0``: every rank must embed the same share of a
    split class -- worlds of 3, 5, 6, 7 with 64-clip batches fall back to whole-class blocks) and the method is DM."""
    blocks_uneven = world > 1 and (-(-num_classes // world)) * world > 1.08 * num_classes
    return "hybrid" if (blocks_uneven and batch_real % world == 0 and method == "dm") else "class"


def sample_real_indices(it: int, counts: Sequence[int], offsets: Sequence[int], batch_real: int,
                        classes: Sequence[int], allow_repeat: bool = False) -> np.ndarray:
    """Pool indices of the real batch of every class in ``classes`` for iteration ``it``: the
    on-device counterpart of ``np.random.permutation(indices_class[c])[:n]``
    (distill_baseline.py:85).  Seeded per (iteration, class) so that every sharding draws the
    same batches.  A class with fewer than ``batch_real`` clips: the reference takes the n it has and averages over
    n; the batched kernels need equal batches, so this raises unless ``allow_repeat`` (clips then repeat cyclically,
    which RE-WEIGHTS that class's mean -- an explicit opt-in, not a default)."""
    out = []
    for c in classes:
        rng = np.random.default_rng([it, c])
        perm = rng.permutation(counts[c])[:batch_real]
        if perm.size < batch_real:
            if not allow_repeat or perm.size == 0:
                raise ValueError("class %d has %d real clips, fewer than batch_real=%d (pass a smaller batch_real, or "
                                 "allow_repeat=True to repeat clips cyclically)" % (c, counts[c], batch_real))
            perm = np.resize(perm, batch_real)
        out.append(offsets[c] + perm)
    return np.concatenate(out).astype(np.int64) if out else np.zeros(0, dtype=np.int64)


def fresh_network_weights(seed: int, device) -> List[torch.Tensor]:
    """A fresh random ConvNet3D feature stack, drawn ON the device with PyTorch's default Conv3d
    distribution (kaiming-uniform a=sqrt(5) == U(+-1/sqrt(fan_in)) for weight and bias), from a
    seed shared by all ranks (no broadcast).  Stands in for ``get_network()`` at
    distill_baseline.py:334, which reseeds from the wall clock."""
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed))
    out = []
    for wi in range(0, len(PARAM_SHAPES), 2):
        fan_in = int(np.prod(PARAM_SHAPES[wi][1:]))
        bound = 1.0 / math.sqrt(fan_in)
        for shp in (PARAM_SHAPES[wi], PARAM_SHAPES[wi + 1]):
            out.append(torch.empty(shp, device=device, dtype=torch.float32).uniform_(-bound, bound, generator=gen))
    return out


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class HipBackend:
    """Production backend: EmbedEngine (MFMA kernels) + the small HIP kernels."""

    def __init__(self, geo: P.NetGeometry, device, prec_real: str = "f16", prec_syn: str = "f16x3", chunk: int = 512,
                 prec_bwd: Optional[str] = "f16x3", syn_batch_hint: Optional[int] = None, real_last: Optional[str] = None):
        """``real_last``: operand format of the real side's LAST conv level when ``prec_real`` is single-pass -- "x3" (hi+lo
        pairs: level 1 emits both planes of its output, level 2 runs three MFMAs per product on 5.6 % of the FLOPs), "c8" (the
        same hi+lo level with its two correction products on the block-scaled fp8 matrix instruction: two MFMA-equivalents per
        product, class means equal to x3's to 3 digits; round 4's default where the geometry has the one-clip last-level program,
        x3 elsewhere) or "x1"; default from ``VD_REAL_LAST`` (DESIGN sections 2 / 5.1c: error budget and cost of each)."""
        from . import engine, hip
        self.hip = hip
        self.device = torch.device(device)
        self.geo = geo
        if real_last is None:
            real_last = os.environ.get("VD_REAL_LAST", "c8")
        assert real_last in ("x1", "x3", "c8")
        self.real_last = real_last if prec_real in ("f16", "bf16") else "x1"
        if prec_syn == prec_real and not prec_bwd:
            self.real_last = "x1"       # one engine for both sides (kept arg-max forwards): a last_hilo engine has no such forward
        # "c8": the hi+lo last level with its two correction products on the fp8 matrix instruction (VD_PREC_F16C8: two
        # MFMA-equivalents per product instead of three, same accuracy on the class means); f16 and the one-clip last-level program
        # only -- other formats / geometries fall back to the three-MFMA form
        self.eng_real = None
        if self.real_last == "c8":
            try:
                self.eng_real = engine.EmbedEngine(geo, prec=prec_real, device=device, chunk=chunk, last_hilo="c8")
            except ValueError:
                self.real_last = "x3"
        if self.eng_real is None:
            self.eng_real = engine.EmbedEngine(geo, prec=prec_real, device=device, chunk=chunk, last_hilo=(self.real_last == "x3"))
        self.eng_syn = self.eng_real if (prec_syn == prec_real and not prec_bwd) else \
            engine.EmbedEngine(geo, prec=prec_syn, device=device, chunk=chunk, prec_bwd=prec_bwd, batch_hint=syn_batch_hint,
                               bwd0_small=os.environ.get("VD_SYN_BWD0_SMALL", "0") == "1")
        self.num_feat = geo.num_feat
        # Two HIP streams: the real-clip forward (98.5 % of the FLOPs, large launches) runs on one,
        # the synthetic-clip forward + backward + optimiser (small launches that cannot fill 256
        # CUs) on a higher-priority second one, so the small kernels overlap the big ones -- also
        # across iterations: the backward of step i runs under the real forward of step i+1.
        self.two_streams = (self.eng_syn is not self.eng_real)
        if self.two_streams:
            pr, ps = (int(v) for v in os.environ.get("VD_STREAM_PRIO", "0,-1").split(","))
            self.s_real = torch.cuda.Stream(device=self.device, priority=pr)
            self.s_syn = torch.cuda.Stream(device=self.device, priority=ps)
        self._ev_real = None
        self.s_prep = None                 # preparation stream (prepare_real_weights), created on first use
        self._real_done: Dict[int, "torch.cuda.Event"] = {}      # per buffer set: behind the launches that last read it
        self._prep_slot: Optional[int] = None
        # Mixed mode (single-pass real side + hi/lo synthetic side of the same 16-bit format).  The real side
        # multiplies by rn16(W): a SYSTEMATIC perturbation of mean f_real (~2e-4 |f|, it does not average out over
        # the 64 clips of a batch) that the exact-weight synthetic side does not share, so it lands undiminished in
        # mean f_real - mean f_syn (measured on fixture G12, where that difference is 4 % of |f|: 0.5 % of the
        # gradient).  Rounding the weights for ALL passes is no cure: the pooling decisions of rn16(W) differ from the
        # reference's (gradient off by 3e-2).  So the synthetic clips get TWO forwards: one with the exact weights,
        # whose ReLU / arg-max decisions route the gradient exactly as the reference does, and a "value pass" with
        # rn16(W), whose features enter the loss -- both sides of the difference then carry the same perturbation.
        self.weight_format = prec_real if (prec_real in ("f16", "bf16") and prec_syn == prec_real + "x3"
                                           and os.environ.get("VD_VALUE_PASS", "1") == "1") else None
        # The better remedy where the real batch of a class has >= 4 clips: DITHER the real side's weights instead.  The
        # class's clips are dealt to G launch groups; group g multiplies by weights rounded down / up such that every
        # weight's mean over the groups is the fp32 value to 1/(2G) ulp (vd_pack_weights_dither), so the perturbation of
        # the class MEAN cancels to first order: 3e-5 |f| at G = 8 against 1.9e-4 (plain) and 8e-5 (value pass; CPU
        # simulation tests/sim_dither_tool.py) -- and the synthetic clips need only their exact-weight forward.
        self.prec_real = prec_real
        self.dither_enabled = True      # (tests switch it off per backend to measure the other two remedies)
        self._dither = 0
        self.resident_rows = os.environ.get("VD_RESIDENT_ROWS", "1") == "1"
        self._pool_rows = None

    def new_network(self, seed: int):
        return fresh_network_weights(seed, self.device)

    def embed_syn(self, x: torch.Tensor, weights):
        """(features that enter the loss, handle for ``embed_backward``) of the synthetic clips; sets the synthetic
        engine's weights itself.  See ``weight_format`` above for the two forwards of the mixed mode."""
        eng = self.eng_syn
        eng.set_weights(weights)
        feats, handle = eng.forward(x, keep=True)
        if self.weight_format is not None and not self._dither:
            # value pass: the levels the (undithered) real side multiplies by plain rn16(W) -- with real_last = x3 its last level
            # runs on the exact hi+lo W2, so W2 stays exact here as well
            eng.set_weights(weights, quantize=self.weight_format, quantize_levels=(0, 1) if self.real_last in ("x3", "c8") else (0, 1, 2))
            feats = eng.forward(x)
        return feats, handle

    def set_weights(self, weights, per_class: int = 0) -> None:
        self.set_real_weights(weights, per_class)
        if self.eng_syn is not self.eng_real:
            self.eng_syn.set_weights(weights)

    def set_real_weights(self, weights, per_class: int = 0, slot: Optional[int] = None) -> None:
        """Weights of the real-clip engine for a step whose real batches hold ``per_class`` clips per class on this rank
        (0: unknown -> no dithering): packs the dithered operand sets when there are enough clips to deal them to.  ``slot``:
        which of the engine's packed-operand buffer sets to fill and launch from (``prepare_real_weights``)."""
        from . import engine
        self._dither = engine.dither_groups(int(per_class), self.prec_real) if (self.dither_enabled and self.eng_syn is not self.eng_real) else 0
        self.eng_real.set_weights(weights, dither=self._dither, slot=slot)

    def prepare_real_weights(self, weights, per_class: int, step: int, nclips: Optional[int] = None) -> None:
        """``set_real_weights`` on a PREPARATION stream, into the buffer set ``step % 2``: the operand packing of a step (0.2 - 0.3 ms
        of small launches: fp16 fragments, eight dithered sets for two levels, the fp8 fragments and their scale) used to sit on
        the real-clip stream between the last level of step i and the first level of step i + 1; with two buffer sets it runs
        under step i's launches -- it waits only for the launches of step i - 1, the previous readers of its buffers -- and
        the real-clip stream waits for an event that has long fired.  Called with the real-clip stream current."""
        # (only where the real side is long: with the 400 clips per rank of an 8-rank job the step is 5 ms and issue-bound -- there the
        #  extra stream's events cost up to 0.5 ms of it, `profiles/r04_rank_proxy_knobs.txt` -- so ``nclips`` below 1600 packs in line)
        if not self.two_streams or os.environ.get("VD_PREP_STREAM", "1") != "1" or \
                (nclips is not None and nclips < int(os.environ.get("VD_PREP_MIN_CLIPS", "1600"))):
            self._prep_slot = None
            return self.set_real_weights(weights, per_class)
        if self.s_prep is None:
            self.s_prep = torch.cuda.Stream(device=self.device, priority=-1)
        k = step % 2
        cur = torch.cuda.current_stream(self.device)                   # = the real-clip stream
        self.s_prep.wait_stream(torch.cuda.default_stream(self.device))          # the freshly drawn fp32 weights
        prev = self._real_done.get(k)
        if prev is not None:
            self.s_prep.wait_event(prev)                               # the launches that last read buffer set k
        for w in weights:
            w.record_stream(self.s_prep)
        with torch.cuda.stream(self.s_prep):
            self.set_real_weights(weights, per_class, slot=k)
            ev = torch.cuda.Event()
            ev.record(self.s_prep)
        cur.wait_event(ev)
        self._prep_slot = k

    def real_launches_done(self) -> None:
        """Record (on the current = real-clip stream) that the launches reading the current buffer set have been issued."""
        if self.s_prep is not None and self._prep_slot is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self._real_done[self._prep_slot] = ev

    # -- stream plumbing (no-ops for single-stream / CPU test backends) ----------------------
    def fork(self):
        """Make both work streams wait for whatever the caller's stream has queued."""
        if self.two_streams:
            cur = torch.cuda.current_stream(self.device)
            self.s_real.wait_stream(cur)
            self.s_syn.wait_stream(cur)

    def on_real(self):
        return torch.cuda.stream(self.s_real) if self.two_streams else _NullCtx()

    def on_syn(self):
        return torch.cuda.stream(self.s_syn) if self.two_streams else _NullCtx()

    def real_to_syn(self, *tensors):
        """The synthetic-side stream may consume `tensors` produced on the real-side stream."""
        if self.two_streams:
            self.s_syn.wait_stream(self.s_real)
            for t in tensors:
                t.record_stream(self.s_syn)

    C8_MIN_ABSMAX = 0.25        # (vd_hip.h, VdConvParams.range_stats: what the fp8 operand planes' fixed scalings were validated for)

    def real_range(self, reset: bool = True) -> Optional[dict]:
        """What the level-1 launches of the fp8-corrected real side saw since the last call: ``{"saturated": outputs that hit the
        +-1792 clamp, "absmax": max |output|}``; None when the real side does not run that format or has not run yet.  Reads
        two device words: the caller has synchronised (``DMTrainer.sync``)."""
        st = getattr(self.eng_real.fwd[1], "range_stats", None) if self.real_last == "c8" else None
        if st is None:
            return None
        sat, bits = (int(v) for v in st.cpu().numpy().view(np.uint32))
        if reset:
            st.zero_()
        return {"saturated": sat, "absmax": float(np.array([bits], dtype=np.uint32).view(np.float32)[0])}

    def check_real_range(self, force_fallback: bool = False) -> Optional[dict]:
        """The fp8 operand planes of the real side's last level use FIXED power-of-two scalings (outputs of level 1 clamped at
        1792, low parts x 2^9): exact to 2^-16 for magnitudes 2^-4 .. 1792, which covers PyTorch-default-initialised networks on
        standardised clips by orders of magnitude -- but not arbitrary weights.  If the launches since the last check left that
        range (something saturated, or the largest output was below ``C8_MIN_ABSMAX``), the backend warns and switches the real
        side to the fp16 hi+lo last level (``real_last = 'x3'``) for all later steps."""
        r = self.real_range()
        if r is None or not (self.range_is_bad(r) or force_fallback):
            return r
        return self.fall_back_to_hi_lo(r)

    def range_is_bad(self, r: Optional[dict]) -> bool:
        """Saturated (or non-finite: the launch counts NaN / Inf outputs as saturated) or too small for the low parts' scaling."""
        return r is not None and (r["saturated"] > 0 or not np.isfinite(r["absmax"]) or (0.0 < r["absmax"] < self.C8_MIN_ABSMAX))

    def fall_back_to_hi_lo(self, r: Optional[dict] = None) -> dict:
        r = dict(r or {"saturated": 0, "absmax": 0.0})
        import warnings
        warnings.warn("fp8-corrected last level outside its validated activation range (%d outputs of level 1 saturated at 1792, "
                      "max |output| %.3g): the real side's last level runs in fp16 hi+lo pairs from now on" % (r["saturated"], r["absmax"]))
        from . import engine
        old = self.eng_real
        self.eng_real = engine.EmbedEngine(self.geo, prec=old.prec_name, device=self.device, chunk=old.chunk, last_hilo=True)
        self.real_last = "x3"
        self._pool_rows = None
        self._prep_slot = None
        self._real_done = {}
        r["fallback"] = "x3"
        return r

    def join(self, *tensors):
        """The caller's stream waits for both work streams."""
        if self.two_streams:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_stream(self.s_real)
            cur.wait_stream(self.s_syn)
            for t in tensors:
                t.record_stream(cur)

    def embed_pool(self, pool: torch.Tensor, index: torch.Tensor, per_class: int = 0) -> torch.Tensor:
        """Features of pool[index].  The pool is static, so it is converted once to the first layer's
        16-bit operand rows and kept resident next to the fp32 clips (6 GB for the miniUCF101-sized
        pool); a real batch is then only an index list (what ``get_images`` + ``.to(device)`` +
        the cast inside the reference's conv do per class and step, distill_baseline.py:84-90).
        ``index`` is class-major with ``per_class`` clips per class; after ``set_real_weights(w, per_class)`` chose G
        dither groups, clip j of every class runs with operand set j mod G (one launch per layer, ``forward_sets``)."""
        segs = [(index.numel() // per_class, per_class)] if (per_class and index.numel() % per_class == 0) else None
        return self.embed_pool_segments(pool, index, segs)

    def embed_pool_segments(self, pool: torch.Tensor, index: torch.Tensor, segments) -> torch.Tensor:
        """As ``embed_pool`` for an index made of several class-major segments ``[(classes, clips per class), ...]`` (the hybrid
        decomposition: whole classes of 64 clips + 1/world slices of the split classes): clip j of every class goes to dither
        group j mod G, all groups in ONE launch per layer; features come back in the order of ``index``."""
        rows = None
        if self.resident_rows:
            key = (pool.data_ptr(), tuple(pool.shape))
            if self._pool_rows is None or self._pool_rows[0] != key:
                self._pool_rows = (key, self.eng_real.pool_rows(pool))
            rows = self._pool_rows[1]
        G = self._dither
        if G and segments and all(per % G == 0 for _, per in segments) and sum(n * per for n, per in segments) == index.numel():
            key = (tuple(segments), G, str(index.device))
            perm = self._seg_perm.get(key) if hasattr(self, "_seg_perm") else None
            if perm is None:     # group-major launch order [g][segment][class][q]: the clips of one weight set are consecutive
                pos, parts = 0, [[] for _ in range(G)]
                for n, per in segments:
                    blk = torch.arange(pos, pos + n * per).view(n, per // G, G)
                    for g in range(G):
                        parts[g].append(blk[:, :, g].reshape(-1))
                    pos += n * per
                perm = torch.cat([torch.cat(pg) for pg in parts]).to(index.device)
                if not hasattr(self, "_seg_perm"):
                    self._seg_perm = {}
                self._seg_perm[key] = perm
            f = self.eng_real.forward_sets(pool, index[perm], rows=rows)
            out = torch.empty_like(f)
            out[perm] = f
            return out
        return self.eng_real.forward(pool, index=index, rows=rows)

    def embed_keep(self, x: torch.Tensor):
        return self.eng_syn.forward(x, keep=True)

    def embed_backward(self, handle, g: torch.Tensor) -> torch.Tensor:
        return self.eng_syn.backward(handle, g)

    def dm_loss(self, f_real: torch.Tensor, f_syn: torch.Tensor, nclass: int):
        nreal, nsyn, dim = f_real.shape[0] // max(nclass, 1), f_syn.shape[0] // max(nclass, 1), self.num_feat
        loss = torch.empty(nclass, dtype=torch.float32, device=self.device)
        g = torch.empty_like(f_syn)
        hip = self.hip
        hip.check(hip.lib().vd_dm_loss(hip.ptr(f_real), hip.ptr(f_syn), nclass, nreal, nsyn, dim, hip.ptr(loss),
                                       hip.ptr(g), hip.stream_ptr(self.device)), "vd_dm_loss")
        return loss, g

    def group_sum(self, x: torch.Tensor, groups: int, per: int, scale: float) -> torch.Tensor:
        hip = self.hip
        out = torch.empty((groups, x.shape[1]), dtype=torch.float32, device=self.device)
        hip.check(hip.lib().vd_group_sum(hip.ptr(x), groups, per, x.shape[1], ctypes.c_float(scale), hip.ptr(out),
                                         hip.stream_ptr(self.device)), "vd_group_sum")
        return out

    def sgd(self, x: torch.Tensor, buf: torch.Tensor, g: torch.Tensor, lr: float, mu: float, first: bool) -> None:
        hip = self.hip
        hip.check(hip.lib().vd_sgd_momentum(hip.ptr(x), hip.ptr(buf), hip.ptr(g), ctypes.c_int64(x.numel()),
                                            ctypes.c_float(lr), ctypes.c_float(mu), int(first),
                                            hip.stream_ptr(self.device)), "vd_sgd_momentum")

    def hallucinate(self, static, dynamic, sidx, didx, w, b):
        hip = self.hip
        n, T, H, W = int(sidx.numel()), dynamic.shape[-4], dynamic.shape[-2], dynamic.shape[-1]
        out = torch.empty((n, T, 3, H, W), dtype=torch.float32, device=self.device)
        hip.check(hip.lib().vd_hallucinator_fwd(hip.ptr(static), hip.ptr(dynamic), hip.ptr(sidx), hip.ptr(didx),
                                                hip.ptr(w), hip.ptr(b), n, T, H, W, hip.ptr(out),
                                                hip.stream_ptr(self.device)), "vd_hallucinator_fwd")
        return out

    def hallucinate_backward(self, g_out, static, dynamic, sidx, didx, w, need_static: bool):
        hip = self.hip
        n, T, H, W = int(sidx.numel()), dynamic.shape[-4], dynamic.shape[-2], dynamic.shape[-1]
        g_dyn = torch.zeros_like(dynamic)
        g_stat = torch.zeros_like(static) if need_static else None
        g_w = torch.zeros(324, dtype=torch.float32, device=self.device)
        g_b = torch.zeros(3, dtype=torch.float32, device=self.device)
        hip.check(hip.lib().vd_hallucinator_bwd(hip.ptr(g_out), hip.ptr(static), hip.ptr(dynamic), hip.ptr(sidx),
                                                hip.ptr(didx), hip.ptr(w), n, T, H, W, hip.ptr(g_dyn),
                                                hip.ptr(g_stat), hip.ptr(g_w), hip.ptr(g_b),
                                                hip.stream_ptr(self.device)), "vd_hallucinator_bwd")
        return g_dyn, g_stat, g_w.view(3, 4, 3, 3, 3), g_b


class RealPool:
    """The rank's share of the real training clips, resident on the device:
    ``clips`` (N,T,3,H,W) fp32 ordered by class, ``counts[c]`` clips for global class c,
    ``offsets[c]`` = index of the class's first clip in ``clips`` (only owned classes used)."""

    def __init__(self, clips: torch.Tensor, counts: Sequence[int], offsets: Sequence[int]):
        self.clips, self.counts, self.offsets = clips, list(counts), list(offsets)

    @staticmethod
    def synthetic(num_classes: int, classes: Sequence[int], per_class: int, geo: P.NetGeometry, device, seed: int = 1234,
                  kind: str = "randn", noise: float = 2.0):
        """Synthetic stand-in for the real training clips (no dataset ships), generated on the device.  Every class is drawn
        from its own generator (seeded by ``seed`` and the class id) and standardised on its own, so a class's clips do not
        depend on which rank holds it: every sharding of a benchmark run works on the same data.

        ``kind='randn'``: independent N(0,1) pixels -- classes are indistinguishable, so any accuracy measured
        on it is chance.  ``kind='templates'``: clip = a smooth random spatio-temporal template of its class (9x9 box-filtered
        noise, amplitude 6) + ``noise`` x N(0,1): a LEARNABLE problem with the same value statistics after standardisation,
        on which ``evaluate_synset``'s top-1 says something about the distilled clips."""
        assert kind in ("randn", "templates")
        n = len(classes) * per_class
        clips = torch.empty((n, geo.frames, 3, geo.height, geo.width), dtype=torch.float32, device=device)
        counts = [per_class] * num_classes
        offsets = [0] * num_classes
        for k, c in enumerate(classes):
            gen = torch.Generator(device=device)
            gen.manual_seed(int(seed) * 1000003 + int(c))
            blk = clips[k * per_class:(k + 1) * per_class]
            blk.normal_(generator=gen)
            if kind == "templates":
                t = torch.randn((geo.frames * 3, 1, geo.height, geo.width), dtype=torch.float32, device=device, generator=gen)
                t = torch.nn.functional.avg_pool2d(t, 9, 1, 4).view(1, geo.frames, 3, geo.height, geo.width) * 6.0
                blk.mul_(float(noise)).add_(t)
            mean = blk.mean(dim=(0, 1, 3, 4), keepdim=True)
            std = blk.std(dim=(0, 1, 3, 4), keepdim=True)
            blk.sub_(mean).div_(std)
            offsets[c] = k * per_class
        return RealPool(clips, counts, offsets)

    @staticmethod
    def from_dataset(dataset, num_classes: int, classes: Sequence[int], device, workers: int = 8):
        """The ``--preload`` + ``indices_class`` bookkeeping of the reference (distill_baseline.py:36-45, 76-81) with the
        clips left in HBM: the items of the owned ``classes`` of a frame-folder dataset (``dataset.FrameFolderVideos``)
        are decoded once, class by class in dataset order, and normalised on the device (``dataset.preload``)."""
        from . import dataset as D
        per_class = D.indices_class(dataset.labels, num_classes)
        order = [i for c in classes for i in per_class[c]]
        clips, _ = D.preload(dataset, device, indices=order, workers=workers)
        counts = [len(per_class[c]) for c in range(num_classes)]
        offsets, o = [0] * num_classes, 0
        for c in classes:
            offsets[c] = o
            o += counts[c]
        return RealPool(clips, counts, offsets)


def check_real_range(trainer) -> None:
    """``sync()`` of the DM trainers: every ``VD_RANGE_CHECK_EVERY`` calls (default 1; 0 = never) look at the activation range the
    fp8-corrected real side recorded (``HipBackend.check_real_range``: a host read of two device words after a device
    synchronisation -- ``sync()`` is where callers read results anyway); the latest record is kept in ``trainer.real_range``."""
    be = trainer.be
    every = int(os.environ.get("VD_RANGE_CHECK_EVERY", "1"))
    if every <= 0 or not hasattr(be, "check_real_range") or getattr(be, "real_last", None) != "c8":
        return
    trainer._range_calls = getattr(trainer, "_range_calls", 0) + 1
    if trainer._range_calls % every:
        return
    torch.cuda.synchronize(be.device)
    # every rank must run the SAME last-level format into the exchanged sums / gathered clips: the decision is the maximum over
    # the ranks (a rank used to decide on its own), and the step it was taken at goes into the record
    force = False
    if collectives_on(getattr(trainer, "world", 1)):
        flag = torch.tensor([1.0 if be.range_is_bad(be.real_range(reset=False)) else 0.0], device=be.device)
        _all_reduce_max(flag)
        force = bool(flag.item() > 0)
    r = be.check_real_range(force_fallback=force)
    if r is not None and r.get("fallback"):
        r["fallback_at_sync"] = trainer._range_calls
        r["fallback_decided_by"] = "max over ranks" if force else "this rank"
    if r is not None and (r["absmax"] > 0.0 or r["saturated"] > 0 or getattr(trainer, "real_range", None) is None):
        trainer.real_range = r          # (a check with no launch since the previous one keeps the previous record)


class DMTrainer:
    """Baseline DM (distill_baseline.py DM branch, :292-361) over the classes owned by this rank."""

    def __init__(self, backend, pool: RealPool, num_classes: int, ipc: int, batch_real: int, lr_img: float,
                 momentum: float = 0.5, rank: int = 0, world: int = 1, image_syn: Optional[torch.Tensor] = None,
                 shard: str = "class", exchange: str = "owner", comm=None):
        """``exchange``: what happens to the pixel gradients of a step.  ``'owner'`` (default): a rank owns the synthetic clips
        and momentum of its classes, so nothing is exchanged (owner-computes).  ``'allreduce'``: the literal form of the
        task's "all-reduce of the matching-loss gradient" -- every rank scatters its rows into the FULL (C * ipc, T, 3, H, W)
        gradient tensor (zeros elsewhere; 120 MB at C = 50, 112 x 112 x 16), the tensor is sum-all-reduced (``comm``: a
        ``hip.Comm`` = RCCL through vd_comm_allreduce_f32 on the synthetic-clip stream; without one, torch.distributed's
        group) and the rank updates its rows from the reduced tensor -- the same update (the other ranks contribute zeros to
        a rank's rows), at the price of the exchange, which ``exchange_ms`` records per step (HIP events around the call).
        That is the tensor distill_baseline.py:353-355 would step on after nn.DataParallel's gather.

        ``shard='class'``: a rank embeds the whole real batch of its own classes (no data-path
        collective).  ``shard='batch'``: every rank embeds 1/world of EVERY class's real batch and
        the per-class feature sums (C x D fp32, 410 KB) are all-reduced; the synthetic clips stay
        class-owned.  The latter balances 50 classes over 8 ranks exactly (6.25 class-equivalents
        each instead of 7,7,6,...) at the price of one small all-reduce per step; the pool must
        then hold all classes on every rank."""
        assert shard in ("class", "batch", "hybrid")
        # (world == 1 normally has nothing to shard; VD_FORCE_BATCH_SHARD=1 keeps the collective path
        #  alive on one rank so that it can be smoke-tested on a single-GPU box)
        self.shard = shard if (world > 1 or os.environ.get("VD_FORCE_BATCH_SHARD") == "1") else "class"
        if self.shard in ("batch", "hybrid") and batch_real % world != 0:
            raise ValueError("shard=%r needs batch_real (%d) divisible by the number of ranks (%d): every rank embeds the same share of a "
                             "split class batch; use shard='class' (distill.choose_shard falls back to it)" % (self.shard, batch_real, world))
        self.be, self.pool = backend, pool
        self.num_classes, self.ipc, self.batch_real = num_classes, ipc, batch_real
        self.lr_img, self.momentum = float(lr_img), float(momentum)
        self.rank, self.world = rank, world
        self.c_lo, self.c_hi = class_range(num_classes, rank, world)
        self.classes = list(range(self.c_lo, self.c_hi))
        if self.shard == "hybrid":
            # ``shard='hybrid'`` (hybrid_partition): whole classes in equal blocks + the left-over classes' real batches split
            # over all ranks; a rank owns the synthetic clips of its block and of the split classes dealt to it
            self.block, self.split, self.split_owned = hybrid_partition(num_classes, rank, world)
            self.classes = self.block + self.split_owned
        if image_syn is None:   # --init real (distill_baseline.py:96-100): first ipc real clips of each class
            idx = np.concatenate([pool.offsets[c] + np.arange(ipc) % pool.counts[c] for c in self.classes]) \
                if self.classes else np.zeros(0, dtype=np.int64)
            image_syn = pool.clips[torch.as_tensor(idx, device=pool.clips.device, dtype=torch.int64)].clone()
        self.image_syn = image_syn.contiguous()
        self.buf = torch.zeros_like(self.image_syn)
        self.steps_done = 0
        assert exchange in ("owner", "allreduce")
        self.exchange, self.comm = exchange, comm
        self.exchange_events = collections.deque(maxlen=512)       # (start, end) HIP events of the latest exchanges (bench.py reads and clears)
        self._g_full = None
        self._pending = None
        # (default: with the fp8-corrected last level, whose short launch no longer absorbs the synthetic side's kernels -- they then
        #  land on the first level, where they cost most; measured -0.3 .. -0.5 ms per step together, nothing apart: DESIGN 5.1c)
        self.defer_backward = os.environ.get("VD_DEFER_BWD", "1" if getattr(backend, "real_last", None) == "c8" else "0") == "1"

    def _allreduce_pixel_grad(self, grad: torch.Tensor) -> torch.Tensor:
        """``exchange='allreduce'``: this rank's gradient rows through the all-reduced full tensor (see ``__init__``).  On one rank
        without a communicator there is nothing to exchange: the rows are returned as they are (bench.py's one-rank leg passes a
        ``hip.Comm`` to time the call anyway)."""
        if self.world == 1 and self.comm is None:
            return grad
        if self._g_full is None:
            self._g_full = torch.empty((self.num_classes * self.ipc,) + tuple(grad.shape[1:]), dtype=grad.dtype, device=grad.device)
            own = self.owned_classes()
            self._g_rows = (torch.as_tensor(own, device=grad.device, dtype=torch.int64).view(-1, 1) * self.ipc
                            + torch.arange(self.ipc, device=grad.device)).view(-1)
        full = self._g_full
        full.zero_()
        full.index_copy_(0, self._g_rows, grad)
        timed = full.is_cuda
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if self.comm is not None:
            COLLECTIVE_CALLS["all_reduce"] += 1
            COLLECTIVE_CALLS["bytes"] += full.numel() * full.element_size()
            self.comm.all_reduce(full)
        else:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                _all_reduce(full)
        if timed:
            e1.record()
            self.exchange_events.append((e0, e1))
        return full.index_select(0, self._g_rows)

    def owned_classes(self, rank: Optional[int] = None) -> List[int]:
        """Classes whose synthetic clips ``rank`` (default: this rank) owns, in the order of its ``image_syn`` rows."""
        rank = self.rank if rank is None else rank
        if self.shard == "hybrid":
            block, _, owned = hybrid_partition(self.num_classes, rank, self.world)
            return block + owned
        lo, hi = class_range(self.num_classes, rank, self.world)
        return list(range(lo, hi))

    def step(self, it: int, overlap: bool = False) -> torch.Tensor:
        """One distillation iteration over this rank's classes; returns the rank-local loss sum
        as a 0-dim device tensor (no host sync).  With ``overlap`` the work is left in flight on
        the backend's two streams (the caller must ``sync()`` / ``global_loss()`` before reading
        results); consecutive steps then overlap: backward(i) runs under the real forward(i+1)."""
        be, ncls = self.be, len(self.classes)
        if self.shard == "batch":
            per = self.batch_real // self.world
            idx = sample_real_indices(it, self.pool.counts, self.pool.offsets, self.batch_real, range(self.num_classes))
            idx = idx.reshape(self.num_classes, self.batch_real)[:, self.rank * per:(self.rank + 1) * per].reshape(-1)
        elif self.shard == "hybrid":        # whole batches of the block classes, then this rank's 1/world slice of every split class
            per = self.batch_real // self.world
            idx_s = sample_real_indices(it, self.pool.counts, self.pool.offsets, self.batch_real, self.split)
            idx_s = idx_s.reshape(len(self.split), self.batch_real)[:, self.rank * per:(self.rank + 1) * per].reshape(-1)
            idx = np.concatenate([sample_real_indices(it, self.pool.counts, self.pool.offsets, self.batch_real, self.block), idx_s])
        else:
            idx = sample_real_indices(it, self.pool.counts, self.pool.offsets, self.batch_real, self.classes)
        dev = self.image_syn.device
        idx_t = torch.as_tensor(idx, device=dev)
        weights = be.new_network(seed=it)
        fork, on_real, on_syn = getattr(be, "fork", None), getattr(be, "on_real", _NullCtx), getattr(be, "on_syn", _NullCtx)
        if fork:
            fork()
        if getattr(be, "two_streams", False):
            # tensors created on the caller's stream but consumed asynchronously on the work streams
            # must not be recycled by the caching allocator before those streams are done with them
            idx_t.record_stream(be.s_real)
            for w in weights:
                w.record_stream(be.s_real)
                w.record_stream(be.s_syn)
            defer = overlap and self.defer_backward
            ev_l0 = None
            with on_real():
                if hasattr(be, "prepare_real_weights"):
                    be.prepare_real_weights(weights, self._per_class(), self.steps_done, nclips=int(idx.size))
                else:
                    be.set_real_weights(weights, self._per_class())
                if defer:       # an event behind the first level's launch of THIS step's real side (see ``_flush_backward``)
                    ev_l0 = torch.cuda.Event()
                    be.eng_real.after_first_level = lambda: ev_l0.record(be.s_real)
                try:
                    f_real = self._real_features(idx_t)
                finally:
                    be.eng_real.after_first_level = None
                if hasattr(be, "real_launches_done"):
                    be.real_launches_done()
            with on_syn():
                # the PREVIOUS step's backward + SGD, if it was deferred: held back until this step's first level is done when this
                # step defers too; issued at once otherwise (a step that does not defer must not overwrite ``_pending``, and its
                # forward must see the updated clips)
                self._flush_backward(ev_l0 if defer else None)
                f_syn, handle = be.embed_syn(self.image_syn, weights)
                be.real_to_syn(f_real)
                f_real = self._exchange(f_real)      # (batch sharding) on the synthetic-clip stream: the real-clip stream is
                loss_c, g_syn = be.dm_loss(f_real, f_syn, ncls)     # free to start the next iteration's forward meanwhile
                self._pending = (handle, g_syn, self.steps_done == 0)
                if not defer:
                    self._flush_backward(None)
                loss = loss_c.sum()
            if not overlap:
                be.join(loss)
            self.steps_done += 1
            return loss
        self._flush_backward(None)          # (a backward deferred by an earlier overlapped step on a two-stream backend)
        if hasattr(be, "set_real_weights"):
            be.set_weights(weights, self._per_class())
        else:
            be.set_weights(weights)
        f_real = self._exchange(self._real_features(idx_t))
        f_syn, handle = be.embed_syn(self.image_syn, weights) if hasattr(be, "embed_syn") else be.embed_keep(self.image_syn)
        loss_c, g_syn = be.dm_loss(f_real, f_syn, ncls)
        grad = be.embed_backward(handle, g_syn)
        if self.exchange == "allreduce":
            grad = self._allreduce_pixel_grad(grad)
        be.sgd(self.image_syn, self.buf, grad, self.lr_img, self.momentum, first=(self.steps_done == 0))
        self.steps_done += 1
        return loss_c.sum()

    def _flush_backward(self, after: Optional["torch.cuda.Event"]) -> None:
        """Backward to the pixels + SGD of the latest class terms (on the CURRENT = synthetic-clip stream).  With
        ``defer_backward`` (overlapped steps only) step i's backward is issued at the START of step i + 1's synthetic side, behind an
        event that follows the first level of step i + 1's real side: the first-level kernel runs ONE eight-wave workgroup per CU
        (all 160 KB of LDS), so a synthetic-side workgroup that gets a CU keeps a whole first-level workgroup out for as long as
        it lives, while under levels 1 / 2 (two workgroups per CU) it merely replaces one of two -- the same work costs less there.
        Same arithmetic, same order per tensor: SGD(i) still precedes the forward of step i + 1."""
        if self._pending is None:
            return
        handle, g_syn, first = self._pending
        self._pending = None
        be = self.be
        if after is not None:
            torch.cuda.current_stream(self.image_syn.device).wait_event(after)
        grad = be.embed_backward(handle, g_syn)
        if self.exchange == "allreduce":
            grad = self._allreduce_pixel_grad(grad)
        be.sgd(self.image_syn, self.buf, grad, self.lr_img, self.momentum, first=first)

    def _real_features(self, idx_t: torch.Tensor) -> torch.Tensor:
        """This rank's contribution to the real side: all clips' features of the owned classes (class sharding), or --
        batch sharding -- the per-class sums of its 1/world slice of every class's batch, pre-scaled by 1/batch_real
        (C x D fp32, 410 KB; ``_exchange`` all-reduces them); hybrid: the class MEANS of the block classes followed by the
        pre-scaled partial sums of the split classes."""
        be = self.be
        if self.shard == "hybrid":
            nb, ns, per = len(self.block), len(self.split), self.batch_real // self.world
            if hasattr(be, "embed_pool_segments"):
                f = be.embed_pool_segments(self.pool.clips, idx_t, [(n, p) for n, p in ((nb, self.batch_real), (ns, per)) if n])
            else:
                f = be.embed_pool(self.pool.clips, idx_t)
            parts = []
            if nb:
                parts.append(be.group_sum(f[:nb * self.batch_real].contiguous(), nb, self.batch_real, 1.0 / self.batch_real))
            if ns:
                parts.append(be.group_sum(f[nb * self.batch_real:].contiguous(), ns, per, 1.0 / self.batch_real))
            return torch.cat(parts, 0)
        f = be.embed_pool(self.pool.clips, idx_t, self._per_class()) if hasattr(be, "set_real_weights") else \
            be.embed_pool(self.pool.clips, idx_t)
        if self.shard != "batch":
            return f
        return be.group_sum(f, self.num_classes, self.batch_real // self.world, 1.0 / self.batch_real)

    def _per_class(self) -> int:
        """Real clips per class in this rank's launches (the unit the real side's dither groups divide; hybrid: the smaller
        of its two segment sizes, which divides the other)."""
        return self.batch_real // self.world if self.shard in ("batch", "hybrid") else self.batch_real

    def _exchange(self, x: torch.Tensor) -> torch.Tensor:
        """Batch sharding: the one data-path collective of a DM step -- all-reduce of the per-class feature sums; returns the
        class MEANS of the owned classes in the form dm_loss() consumes (one row per class, a 'batch' of one).  Hybrid: only
        the split classes' rows (2048 floats each) are all-reduced."""
        if self.shard == "class":
            return x
        import torch.distributed as dist
        live = dist.is_available() and dist.is_initialized()
        if self.shard == "hybrid":
            nb = len(self.block)
            if not self.split:
                return x
            part = x[nb:].contiguous()
            if live:
                _all_reduce(part)
            if not self.split_owned:
                return x[:nb].contiguous()
            if getattr(self, "_own_idx", None) is None or self._own_idx.device != x.device:
                # (a device index built ONCE: indexing with a Python list would upload it on every step -- a blocking copy on the
                #  synthetic-clip stream, which has just been made to wait for the whole real forward: the host could no longer
                #  run ahead, +1.4 ms per step)
                self._own_idx = torch.as_tensor([self.split.index(c) for c in self.split_owned], dtype=torch.int64, device=x.device)
            return torch.cat([x[:nb], part.index_select(0, self._own_idx)], 0)
        if live:
            _all_reduce(x)
        return x[self.c_lo:self.c_hi].contiguous()

    def global_loss(self, local_loss: torch.Tensor) -> torch.Tensor:
        """Sum of the per-rank losses (== the reference's ``loss`` before /num_classes)."""
        be = self.be
        ctx = be.on_syn() if getattr(be, "two_streams", False) else _NullCtx()
        with ctx:
            if collectives_on(self.world):
                import torch.distributed as dist
                local_loss = local_loss.clone()
                _all_reduce(local_loss)
        return local_loss

    def sync(self) -> None:
        """Wait (on the caller's stream) for everything the trainer has in flight (a deferred backward is issued first)."""
        if getattr(self.be, "two_streams", False):
            if getattr(self, "_pending", None) is not None:
                with self.be.on_syn():
                    self._flush_backward(None)
            self.be.join()
        check_real_range(self)

    def mark(self):
        """A timing event behind the last kernel ISSUED by the latest step (bench.py: per-step times under overlap).  With a
        deferred backward that is the step's loss kernel: its backward + SGD are issued by the next step (or ``sync``), so a
        step's interval holds the previous step's backward instead of its own -- the same work, shifted by one step; the
        total over the timed region is exact once ``sync()`` has run."""
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(self.be.s_syn if getattr(self.be, "two_streams", False) else torch.cuda.current_stream(self.image_syn.device))
        return ev

    def gather_syn(self) -> torch.Tensor:
        """All synthetic clips in class order on every rank (evaluation / ``images_*.pt``).  Waits for the steps still
        in flight on the trainer's own streams first: with ``overlap=True`` the SGD write of the latest step may not have
        landed when the caller's stream starts copying."""
        if hasattr(self, "sync"):
            self.sync()
        if not collectives_on(self.world):
            return self.image_syn
        import torch.distributed as dist
        owners = [self.owned_classes(r) if hasattr(self, "owned_classes") else
                  list(range(*class_range(self.num_classes, r, self.world))) for r in range(self.world)]
        sizes = [len(o) * self.ipc for o in owners]
        mx = max(sizes)
        pad = torch.zeros((mx,) + tuple(self.image_syn.shape[1:]), dtype=self.image_syn.dtype, device=self.image_syn.device)
        pad[:self.image_syn.shape[0]] = self.image_syn
        parts = [torch.empty_like(pad) for _ in range(self.world)]
        _all_gather(parts, pad)
        out = torch.cat([p[:s] for p, s in zip(parts, sizes)], dim=0)
        order = [c for o in owners for c in o]                    # class of every gathered block of ipc rows
        if order != sorted(order):                                # (hybrid: the split classes sit behind each rank's block)
            pos = torch.as_tensor(np.argsort(order), device=out.device)
            out = out.view(len(order), self.ipc, *out.shape[1:])[pos].reshape(out.shape)
        return out


class S2DTrainer:
    """DM + static/dynamic memories (distill_s2d_ms.py DM branch, :393-438): the synthetic clip of
    (class, v) is hallucinator(static[static_idx], dynamic[label, dynamic_idx]); static memory
    frozen (``--no_train_static``) or trained; SGD(momentum .95) on dynamic memory and
    hallucinator.  Dynamic rows are owned by the class's rank; the 327 hallucinator gradients
    are summed over ranks."""

    def __init__(self, backend, pool: RealPool, num_classes: int, vpc: int, spc: int, dpc: int, batch_real: int,
                 static_syn: torch.Tensor, dynamic_syn: torch.Tensor, hal_w: torch.Tensor, hal_b: torch.Tensor,
                 lr_dynamic: float, lr_hal: float, lr_static: float = 0.0, train_static: bool = False,
                 momentum: float = 0.95, rank: int = 0, world: int = 1):
        self.be, self.pool = backend, pool
        self.num_classes, self.vpc, self.spc, self.dpc, self.batch_real = num_classes, vpc, spc, dpc, batch_real
        self.rank, self.world = rank, world
        self.c_lo, self.c_hi = class_range(num_classes, rank, world)
        self.classes = list(range(self.c_lo, self.c_hi))
        # local shards: static rows of owned classes, dynamic rows of owned classes (flattened (c,dpc))
        self.static = static_syn[self.c_lo * spc:self.c_hi * spc].contiguous()
        self.dynamic = dynamic_syn[self.c_lo:self.c_hi].reshape((-1,) + tuple(dynamic_syn.shape[2:])).contiguous()
        self.hal_w, self.hal_b = hal_w.clone().contiguous(), hal_b.clone().contiguous()
        self.lr_dynamic, self.lr_hal, self.lr_static = float(lr_dynamic), float(lr_hal), float(lr_static)
        self.train_static, self.momentum = train_static, float(momentum)
        self.buf_d = torch.zeros_like(self.dynamic)
        self.buf_w, self.buf_b = torch.zeros_like(self.hal_w), torch.zeros_like(self.hal_b)
        self.buf_s = torch.zeros_like(self.static) if train_static else None
        self.steps_done = 0
        self._pending = None
        self.last_grads = None
        self.defer_backward = os.environ.get("VD_DEFER_BWD", "1" if getattr(backend, "real_last", None) == "c8" else "0") == "1"

    def indices(self, it: int, draws: Optional[Tuple[np.ndarray, np.ndarray]] = None):
        """distill_s2d_ms.py:402-406 for the owned classes; the two randint(2) draws are seeded per
        (iteration, class) so that every sharding composes the same clips."""
        ncls = len(self.classes)
        label = np.repeat(np.arange(ncls), self.vpc)
        idx = np.tile(np.arange(self.vpc), ncls)
        if draws is None:
            dd = np.concatenate([np.random.default_rng([it, c, 1]).integers(0, 2, self.vpc) for c in self.classes]) if ncls else idx
            ds = np.concatenate([np.random.default_rng([it, c, 2]).integers(0, 2, self.vpc) for c in self.classes]) if ncls else idx
        else:
            dd, ds = draws[0][self.c_lo * self.vpc:self.c_hi * self.vpc], draws[1][self.c_lo * self.vpc:self.c_hi * self.vpc]
        dynamic_idx = label * self.dpc + 2 * idx + dd      # row in the flattened (class, dpc) dynamic shard
        static_idx = self.spc * label + 2 * idx + ds
        return static_idx.astype(np.int64), dynamic_idx.astype(np.int64)

    def step(self, it: int, draws=None, overlap: bool = False) -> torch.Tensor:
        """One s2d iteration over this rank's classes.  With a two-stream backend the real-clip forward runs on
        one stream and hallucinator + synthetic forward / backward + optimiser on the other (as in DMTrainer);
        ``overlap`` leaves the work in flight so that consecutive iterations overlap (``sync()`` before reading)."""
        be, ncls = self.be, len(self.classes)
        dev = self.dynamic.device
        weights = be.new_network(seed=it)
        sidx_np, didx_np = self.indices(it, draws)
        sidx, didx = torch.as_tensor(sidx_np, device=dev), torch.as_tensor(didx_np, device=dev)
        idx = sample_real_indices(it, self.pool.counts, self.pool.offsets, self.batch_real, self.classes)
        idx_t = torch.as_tensor(idx, device=dev)
        two = getattr(be, "two_streams", False)
        on_real, on_syn = (be.on_real, be.on_syn) if two else (_NullCtx, _NullCtx)
        if two:
            be.fork()
            idx_t.record_stream(be.s_real)
            for t in (sidx, didx):
                t.record_stream(be.s_syn)
            for w in weights:
                w.record_stream(be.s_real)
                w.record_stream(be.s_syn)
            defer = overlap and self.defer_backward      # (as DMTrainer.defer_backward: the previous step's backward behind THIS step's first level)
            ev_l0 = None
            with on_real():
                be.prepare_real_weights(weights, self.batch_real, self.steps_done, nclips=int(idx.size))      # (operand packing on the preparation stream)
                if defer:
                    ev_l0 = torch.cuda.Event()
                    be.eng_real.after_first_level = lambda: ev_l0.record(be.s_real)
                try:
                    f_real = be.embed_pool(self.pool.clips, idx_t, self.batch_real)
                finally:
                    if defer:
                        be.eng_real.after_first_level = None
                be.real_launches_done()
        elif hasattr(be, "set_real_weights"):
            defer = False
            be.set_weights(weights, self.batch_real)
            f_real = be.embed_pool(self.pool.clips, idx_t, self.batch_real)
        else:
            defer = False
            be.set_weights(weights)
            f_real = be.embed_pool(self.pool.clips, idx_t)
        with on_syn():
            self._flush_backward(ev_l0 if defer else None)      # (see DMTrainer.step: a non-deferring step issues a pending backward at once)
            image_syn = be.hallucinate(self.static, self.dynamic, sidx, didx, self.hal_w, self.hal_b)
            f_syn, handle = be.embed_syn(image_syn, weights) if hasattr(be, "embed_syn") else be.embed_keep(image_syn)
            if two:
                be.real_to_syn(f_real)
            loss_c, g_syn = be.dm_loss(f_real, f_syn, ncls)
            self._pending = (handle, g_syn, sidx, didx, self.steps_done == 0)
            if not defer:
                self._flush_backward(None)
            loss = loss_c.sum()
        if two and not overlap:
            be.join(loss, *self.last_grads)
        self.steps_done += 1
        return loss

    def _flush_backward(self, after) -> None:
        """Backward through the embedding and the hallucinator + the SGD steps of the latest class terms, on the current
        (synthetic-clip) stream; see ``DMTrainer._flush_backward`` for the deferred form."""
        if self._pending is None:
            return
        handle, g_syn, sidx, didx, first = self._pending
        self._pending = None
        be = self.be
        if after is not None:
            torch.cuda.current_stream(self.dynamic.device).wait_event(after)
        g_img = be.embed_backward(handle, g_syn)
        g_dyn, g_stat, g_w, g_b = be.hallucinate_backward(g_img, self.static, self.dynamic, sidx, didx, self.hal_w,
                                                          self.train_static)
        if collectives_on(self.world):   # the hallucinator is shared by all classes: one 1.3 KB all-reduce
            flat = torch.cat([g_w.reshape(-1), g_b.reshape(-1)])
            _all_reduce(flat)
            g_w, g_b = flat[:324].view_as(g_w), flat[324:]
        be.sgd(self.dynamic, self.buf_d, g_dyn, self.lr_dynamic, self.momentum, first)
        be.sgd(self.hal_w, self.buf_w, g_w.contiguous(), self.lr_hal, self.momentum, first)
        be.sgd(self.hal_b, self.buf_b, g_b.contiguous(), self.lr_hal, self.momentum, first)
        if self.train_static:
            be.sgd(self.static, self.buf_s, g_stat, self.lr_static, self.momentum, first)
        self.last_grads = (g_dyn, g_w, g_b)

    def sync(self) -> None:
        if getattr(self.be, "two_streams", False):
            if self._pending is not None:
                with self.be.on_syn():
                    self._flush_backward(None)
            self.be.join()
        check_real_range(self)

    global_loss = DMTrainer.global_loss

    def mark(self):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(self.be.s_syn if getattr(self.be, "two_streams", False) else torch.cuda.current_stream(self.dynamic.device))
        return ev


# ------------------------------------------------------------------------------------------------
# Gradient matching (DC): SURVEY section 8(f)-2, config 4
# ------------------------------------------------------------------------------------------------
def fresh_full_network(seed: int, num_classes: int, device) -> List[torch.Tensor]:
    """All 8 ConvNet3D tensors (feature stack + 1x1x1 logit conv), PyTorch default init, drawn on
    the device from a seed shared by all ranks (``fresh_network_weights`` plus the head)."""
    out = fresh_network_weights(seed, device)
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed) + (1 << 20))
    bound = 1.0 / math.sqrt(128.0)
    for shp in ((num_classes, 128, 1, 1, 1), (num_classes,)):
        out.append(torch.empty(shp, device=device, dtype=torch.float32).uniform_(-bound, bound, generator=gen))
    return out


class HipGMOps:
    """The device operations of the gradient-matching loop, on the HIP path."""

    def __init__(self, device, dis_metric: str = "ours"):
        import types
        from . import hip
        self.hip, self.device = hip, torch.device(device)
        self.args = types.SimpleNamespace(device=self.device, dis_metric=dis_metric)
        self._side = None

    def make_net(self, params: Sequence[torch.Tensor], geo: P.NetGeometry, num_classes: int):
        from . import networks
        net = networks.ConvNet3D(3, num_classes, 128, 3, 'relu', 'none', 'maxpooling', frames=geo.frames,
                                 im_size=(geo.height, geo.width))
        net = net.to(self.device)
        with torch.no_grad():
            for p, w in zip(net.parameters(), params):
                p.copy_(w)
        return net.train()

    def param_grads(self, net, x, labels, create_graph: bool, slot: int = 0):
        return net.param_grads(x, labels, create_graph=create_graph, slot=slot)[1]

    def param_grads_grouped(self, net, x, labels, groups: int, slot: int = 0):
        """Detached first-order parameter gradients of ``groups`` equal consecutive sub-batches (real batches of several classes):
        one forward over all clips, one backward per sub-batch."""
        return [[t.detach() for t in g] for _, g in net.param_grads_grouped(x, labels, groups, slot=slot)]

    def lane_streams(self, n: int):
        """Streams of the class lanes (GMTrainer): class terms are independent chains of ~150 small launches each, so several
        run concurrently, each on its own stream with its own engine slot."""
        if getattr(self, "_lanes", None) is None or len(self._lanes) != n:
            self._lanes = [torch.cuda.Stream(device=self.device) for _ in range(n)]
        return self._lanes

    def param_grads_async(self, net, x, labels):
        """First-order parameter gradients of a REAL batch on a side stream: the real batch of the next class
        is independent of the synthetic-clip passes of the current one, and neither fills the GPU alone
        (64-clip launches / 5-clip launches), so the trainer overlaps them.  Returns (gradients, event)."""
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        main = torch.cuda.current_stream(self.device)
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side):
            g = [t.detach() for t in net.param_grads(x, labels, create_graph=False)[1]]
            ev = torch.cuda.Event()
            ev.record(self._side)
        for t in g + [x, labels]:
            t.record_stream(self._side)
            t.record_stream(main)
        return g, ev

    def match_loss(self, gw_syn, gw_real):
        from . import utils
        return utils.match_loss(gw_syn, gw_real, self.args)

    def sgd(self, x, buf, g, lr, mu, first):
        hip = self.hip
        hip.check(hip.lib().vd_sgd_momentum(hip.ptr(x), hip.ptr(buf), hip.ptr(g), ctypes.c_int64(x.numel()), ctypes.c_float(lr),
                                            ctypes.c_float(mu), int(first), hip.stream_ptr(self.device)), "vd_sgd_momentum")

    def train_epoch(self, net, images, labels, optimizer, batch_train: int):
        import types
        from . import utils
        args = types.SimpleNamespace(device=self.device, model="ConvNet3D", eval_mode="SS")
        loader = torch.utils.data.DataLoader(utils.TensorDataset(images, labels), batch_size=batch_train, shuffle=True,
                                             num_workers=0)
        return utils.epoch('train', loader, net, optimizer, torch.nn.CrossEntropyLoss().to(self.device), args)


class GMTrainer:
    """Gradient matching over the classes owned by this rank, in the structure of the upstream DC
    loop the reference's ``--method DC`` arguments / ``get_loops`` / ``match_loss`` belong to
    (distill_baseline.py:22-26, 370-379; utils.py:634-709; the loop body itself is absent from the
    reference):

        net <- fresh network;  for ol in range(outer_loop):
            for every class c:  gw_real = dCE/dparams on a real batch (detached)
                                gw_syn  = dCE/dparams on the class's synthetic clips (create_graph)
                                loss   += match_loss(gw_syn, gw_real)
            pixels <- SGD(lr_img, momentum .5) on d loss / d pixels
            (unless last) train net on the synthetic set for inner_loop epochs, SGD(lr_net, m .5)

    Pixel gradients are class-disjoint, so classes shard over ranks with no gradient exchange; the
    network update needs every rank's synthetic clips (all-gather), after which all ranks take the
    same deterministic training step."""

    def __init__(self, ops, pool: RealPool, geo: P.NetGeometry, num_classes: int, ipc: int, batch_real: int, lr_img: float,
                 lr_net: float = 0.01, momentum: float = 0.5, rank: int = 0, world: int = 1,
                 image_syn: Optional[torch.Tensor] = None, outer_loop: int = 1, inner_loop: int = 1, batch_train: int = 256,
                 dropout_p: Optional[float] = None, net_init=None):
        self.ops, self.pool, self.geo = ops, pool, geo
        self.net_init = net_init        # it -> 8 tensors; default: fresh_full_network(it) drawn on the device
        self.num_classes, self.ipc, self.batch_real = num_classes, ipc, batch_real
        self.lr_img, self.lr_net, self.momentum = float(lr_img), float(lr_net), float(momentum)
        self.rank, self.world = rank, world
        self.outer_loop, self.inner_loop, self.batch_train = int(outer_loop), int(inner_loop), int(batch_train)
        self.dropout_p = dropout_p
        self.c_lo, self.c_hi = class_range(num_classes, rank, world)
        self.classes = list(range(self.c_lo, self.c_hi))
        if image_syn is None:
            idx = np.concatenate([pool.offsets[c] + np.arange(ipc) % pool.counts[c] for c in self.classes]) \
                if self.classes else np.zeros(0, dtype=np.int64)
            image_syn = pool.clips[torch.as_tensor(idx, device=pool.clips.device, dtype=torch.int64)].clone()
        self.image_syn = image_syn.contiguous()
        self.buf = torch.zeros_like(self.image_syn)
        self.steps_done = 0

    def step(self, it: int) -> torch.Tensor:
        """One ``for it`` iteration; returns the rank-local matching loss summed over the outer loop."""
        ops, dev = self.ops, self.image_syn.device
        params = self.net_init(it) if self.net_init is not None else fresh_full_network(it, self.num_classes, dev)
        net = ops.make_net([p.to(dev) for p in params], self.geo, self.num_classes)
        if self.dropout_p is not None:
            net.dropout.p = float(self.dropout_p)
        for p in net.parameters():
            p.requires_grad_(True)
        opt_net = torch.optim.SGD(net.parameters(), lr=self.lr_net, momentum=0.5)
        total = torch.zeros((), device=dev)
        for ol in range(self.outer_loop):
            idx = sample_real_indices(it * self.outer_loop + ol, self.pool.counts, self.pool.offsets, self.batch_real, self.classes)
            idx_t = torch.as_tensor(idx, device=dev).reshape(len(self.classes), -1)
            g_img = torch.zeros_like(self.image_syn)
            nlanes = int(os.environ.get("VD_GM_LANES", "8")) if hasattr(ops, "lane_streams") else 1
            if nlanes > 1 and len(self.classes) > 1:
                # class lanes: class k runs on stream k % nlanes with engine slot k % nlanes; the chains only meet in g_img
                # (disjoint rows) and in the loss sum
                lanes = ops.lane_streams(nlanes)
                main = torch.cuda.current_stream(dev)
                totals = [torch.zeros((), device=dev) for _ in lanes]
                for st in lanes:
                    st.wait_stream(main)
                # VD_GM_GROUP = g > 1: the real batches of g consecutive classes of a lane share ONE forward (256 clips fill every
                # level's launch; 64 leave the last level at a quarter of the chip), their backward passes run per class on slices
                # of it.  Off by default: measured null (272.8 / 267.3 / 274.1 / 275.2 ms per step for g = 1 / 2 / 4 / 8 -- the
                # other lanes already fill what a 64-clip launch leaves idle)
                gsz = max(1, int(os.environ.get("VD_GM_GROUP", "1"))) if hasattr(ops, "param_grads_grouped") else 1
                ncls = len(self.classes)
                grouped = {}
                for k, c in enumerate(self.classes):
                    lane = (k // gsz) % nlanes
                    with torch.cuda.stream(lanes[lane]):
                        if gsz > 1 and k % gsz == 0:
                            k1 = min(k + gsz, ncls)
                            real_all = self.pool.clips[idx_t[k:k1].reshape(-1)]
                            # (labels built ON the device: an upload here would be a blocking copy on a busy lane stream)
                            lab_all = torch.cat([torch.full((idx_t.shape[1],), cc, dtype=torch.int64, device=dev) for cc in self.classes[k:k1]])
                            grouped = dict(zip(range(k, k1), ops.param_grads_grouped(net, real_all, lab_all, k1 - k, slot=lane)))
                        if gsz > 1:
                            gw_real = grouped.pop(k)
                        else:
                            real = self.pool.clips[idx_t[k]]
                            lab_r = torch.full((real.shape[0],), c, dtype=torch.int64, device=dev)
                            gw_real = [t.detach() for t in ops.param_grads(net, real, lab_r, False, slot=lane)]
                        lab_s = torch.full((self.ipc,), c, dtype=torch.int64, device=dev)
                        syn = self.image_syn[k * self.ipc:(k + 1) * self.ipc].detach().clone().requires_grad_(True)
                        gw_syn = ops.param_grads(net, syn, lab_s, True, slot=lane)
                        loss = ops.match_loss(gw_syn, gw_real)
                        (g,) = torch.autograd.grad(loss, syn)
                        g_img[k * self.ipc:(k + 1) * self.ipc] = g
                        totals[lane] = totals[lane] + loss.detach()
                for st, tl in zip(lanes, totals):
                    main.wait_stream(st)
                    tl.record_stream(main)
                total = total + torch.stack(totals).sum()
            else:
                total = total + self._class_terms_serial(ops, net, idx_t, g_img, dev)
            ops.sgd(self.image_syn, self.buf, g_img, self.lr_img, self.momentum, first=(self.steps_done == 0))
            self.steps_done += 1
            if ol == self.outer_loop - 1:
                break
            syn_all = self.gather_syn().detach().clone()
            lab_all = torch.arange(self.num_classes, device=dev).repeat_interleave(self.ipc)
            for _ in range(self.inner_loop):
                ops.train_epoch(net, syn_all, lab_all, opt_net, self.batch_train)
        return total

    def _class_terms_serial(self, ops, net, idx_t, g_img, dev):
        """All class terms on the caller's stream; with a HIP backend the real batch of class k+1 is prefetched on a side
        stream under the synthetic-clip passes of class k."""
        total = torch.zeros((), device=dev)
        overlap = hasattr(ops, "param_grads_async") and os.environ.get("VD_GM_OVERLAP", "1") == "1"

        def real_side(k):
            c = self.classes[k]
            real = self.pool.clips[idx_t[k]]
            lab_r = torch.full((real.shape[0],), c, dtype=torch.int64, device=dev)
            if overlap:
                return ops.param_grads_async(net, real, lab_r)
            return [t.detach() for t in ops.param_grads(net, real, lab_r, False)], None
        pending = real_side(0) if self.classes else None
        for k, c in enumerate(self.classes):
            gw_real, ready = pending
            if k + 1 < len(self.classes):
                pending = real_side(k + 1)          # runs under this class's synthetic-clip passes
            if ready is not None:
                torch.cuda.current_stream(dev).wait_event(ready)
            lab_s = torch.full((self.ipc,), c, dtype=torch.int64, device=dev)
            syn = self.image_syn[k * self.ipc:(k + 1) * self.ipc].detach().clone().requires_grad_(True)
            gw_syn = ops.param_grads(net, syn, lab_s, True)
            loss = ops.match_loss(gw_syn, gw_real)
            (g,) = torch.autograd.grad(loss, syn)
            g_img[k * self.ipc:(k + 1) * self.ipc] = g
            total = total + loss.detach()
        return total

    def global_loss(self, local_loss: torch.Tensor) -> torch.Tensor:
        if collectives_on(self.world):
            import torch.distributed as dist
            local_loss = local_loss.clone()
            _all_reduce(local_loss)
        return local_loss

    gather_syn = DMTrainer.gather_syn


# ------------------------------------------------------------------------------------------------
# Trajectory matching (MTT): SURVEY section 8(f)-3, config 5
# ------------------------------------------------------------------------------------------------
FULL_SHAPES = lambda k: PARAM_SHAPES + ((k, 128, 1, 1, 1), (k,))     # noqa: E731


def flatten_params(params: Sequence[torch.Tensor]) -> torch.Tensor:
    """ReparamModule's flat parameter (reparam_module.py:51): parameters() order, concatenated."""
    return torch.cat([p.reshape(-1) for p in params], 0)


def unflatten_params(flat: torch.Tensor, num_classes: int) -> List[torch.Tensor]:
    out, o = [], 0
    for shp in FULL_SHAPES(num_classes):
        n = int(np.prod(shp))
        out.append(flat[o:o + n].view(shp))
        o += n
    return out


class HipMTTOps:
    """Device operations of the unrolled student loop on the HIP path (train.GradMatchEngine)."""

    def __init__(self, geo: P.NetGeometry, num_classes: int, device, dropout_p: float = 0.5,
                 batch_hint: Optional[int] = None):
        from . import hip, networks, train
        self.hip, self.device = hip, torch.device(device)
        pool = (2, 2, 2) if geo.height > 64 else (2, 1, 1)          # networks.py:733
        self.te = train.GradMatchEngine(geo, num_classes, pool, device, prec=networks.get_precision()["match"],
                                        batch_hint=batch_hint)
        self.dropout_p = float(dropout_p)

    def grads(self, params, x, labels):
        """-> ([8 gradients of the mean CE], handle for hvp)."""
        te, mask = self.te, None
        if self.dropout_p > 0:
            keep = 1.0 - self.dropout_p
            mask = torch.bernoulli(torch.full((x.shape[0], te.C, te.Tp), keep, device=self.device)) / keep
        _, _, g, state = te.param_grads(x, labels, params, mask)
        return g, (state, [p for p in params])

    def hvp(self, handle, v):
        """-> (d <v, g> / dx, [d <v, g> / dparams])."""
        state, params = handle
        return self.te.vjp(state, v, params, param_adjoint=True)

    sgd = HipGMOps.sgd
    hallucinate = HipBackend.hallucinate
    hallucinate_backward = HipBackend.hallucinate_backward


class MTTTrainer:
    """One MTT iteration (distill_baseline.py:192-275): ``syn_steps`` unrolled student updates
    theta_{s+1} = theta_s - syn_lr * dCE(x_s; theta_s)/dtheta from an expert's start epoch, the
    normalised distance to the expert's parameters ``expert_epochs`` later, and its gradient w.r.t.
    the synthetic clips and syn_lr.  The reverse sweep is explicit (no autograd tape):

        thetabar_N = 2 (theta_N - target) / |theta_0 - target|^2
        for s = N-1 .. 0:   v = -syn_lr * thetabar_{s+1};  lrbar -= <thetabar_{s+1}, g_s>
                            (xbar_s, Hv) = second-order pass of step s with adjoint v
                            thetabar_s = thetabar_{s+1} + Hv

    All per-step state (activations, arg-max, first-order gradients at the conv outputs) of the
    unrolled loop stays resident in HBM between the forward and the reverse sweep, and so do the scalars
    (syn_lr, its momentum, the two distances, d/d syn_lr): an iteration enqueues without a host sync.
    Multi-GPU: the synthetic batch of every student step is split over ranks (what the reference's
    DataParallel does, :238-241); the flat parameter gradient and the Hessian-vector product are
    all-reduced per inner step (2 x 14.6 MB), pixel gradients are rank-disjoint and summed once."""

    LR_MOMENTUM = 0.5           # optimizer_lr = SGD([syn_lr], lr_lr, momentum=0.5), distill_baseline.py:108

    def __init__(self, ops, num_classes: int, image_syn: torch.Tensor, label_syn: torch.Tensor, syn_lr: float,
                 lr_img: float, lr_lr: float, syn_steps: int, batch_syn: int, expert_epochs: int, max_start_epoch: int,
                 momentum: float = 0.5, rank: int = 0, world: int = 1):
        self.ops, self.num_classes = ops, num_classes
        self.image_syn, self.label_syn = image_syn.contiguous(), label_syn
        self.buf = torch.zeros_like(self.image_syn)
        dev = self.image_syn.device
        self.syn_lr = torch.tensor(float(syn_lr), dtype=torch.float32, device=dev)
        self.lr_buf = torch.zeros((), dtype=torch.float32, device=dev)
        self.lr_img, self.lr_lr, self.momentum = float(lr_img), float(lr_lr), float(momentum)
        self.syn_steps, self.batch_syn = int(syn_steps), int(batch_syn)
        self.expert_epochs, self.max_start_epoch = int(expert_epochs), int(max_start_epoch)
        self.rank, self.world = rank, world
        self.steps_done = 0
        self.last_grads = None
        self.keep_tape = False      # True: ``last_tape`` keeps the unrolled steps' handles (arg-max bytes of every student forward) after step()
        self.last_tape = None

    # -- what differs between raw synthetic clips and the s2d composition ---------------------------------------------
    def _num_items(self) -> int:
        return int(self.image_syn.shape[0])

    def _begin(self, dev) -> None:
        self._g_img = torch.zeros_like(self.image_syn)

    def _clips(self, batch: torch.Tensor, sel: slice, step: int, it: int):
        """-> (clips of this rank's share ``batch[sel]`` of the student batch, their labels, context for ``_push``)."""
        mine = batch[sel]
        return self.image_syn[mine], self.label_syn.to(mine.device)[mine], mine

    def _push(self, ctx, dx: torch.Tensor) -> None:
        self._g_img.index_add_(0, ctx, dx)

    def _finish(self, update: bool):
        g_img = self._g_img
        if collectives_on(self.world):
            import torch.distributed as dist
            _all_reduce(g_img)
        if update:
            self.ops.sgd(self.image_syn, self.buf, g_img, self.lr_img, self.momentum, first=(self.steps_done == 0))
        return (g_img,)

    # -------------------------------------------------------------------------------------------------------------------
    def _allreduce(self, flat: torch.Tensor) -> torch.Tensor:
        if collectives_on(self.world):
            import torch.distributed as dist
            _all_reduce(flat)
        return flat

    def step(self, it: int, trajectory, start_epoch: Optional[int] = None, index_chunks=None, update: bool = True):
        """``trajectory``: one expert (list over epochs of the 8 parameter tensors, as stored in
        ``replay_buffer_N.pt``, buffer.py:75-104).  Returns the grand loss as a 0-dim device tensor."""
        dev = self.image_syn.device
        rng = np.random.default_rng([it, 17])
        if start_epoch is None:
            start_epoch = int(rng.integers(0, self.max_start_epoch))
        start = [p.to(dev, torch.float32) for p in trajectory[start_epoch]]
        target = flatten_params([p.to(dev, torch.float32) for p in trajectory[start_epoch + self.expert_epochs]])
        if index_chunks is None:            # torch.randperm + split + pop() of distill_baseline.py:226-233, seeded per iteration
            index_chunks, pending = [], []
            for _ in range(self.syn_steps):
                if not pending:
                    perm = torch.as_tensor(rng.permutation(self._num_items()))
                    pending = list(torch.split(perm, self.batch_syn))
                index_chunks.append(pending.pop())
        self._begin(dev)
        theta0 = flatten_params(start)
        theta = theta0.clone()
        tape = []
        for step, idx in enumerate(index_chunks):
            idx = idx.to(dev)
            sel = slice(self.rank, None, self.world) if self.world > 1 else slice(None)    # this rank's share of the batch
            n_mine = len(range(*sel.indices(int(idx.numel()))))
            share = float(n_mine) / float(idx.numel())
            params = unflatten_params(theta, self.num_classes)
            if n_mine:
                x, labels, ctx = self._clips(idx, sel, step, it)
                g, handle = self.ops.grads(params, x, labels)
                g = flatten_params(g)
                if share != 1.0:
                    g = g * share
            else:
                g, handle, ctx = torch.zeros_like(theta), None, None
            g = self._allreduce(g)
            tape.append((ctx, share, handle, g))
            theta = theta - self.syn_lr * g
        dist0 = ((theta0 - target) ** 2).sum()
        grand = ((theta - target) ** 2).sum() / dist0
        # ---- reverse sweep ---------------------------------------------------------------------
        tbar = 2.0 * (theta - target) / dist0
        g_lr = torch.zeros((), dtype=torch.float32, device=dev)
        for ctx, share, handle, g in reversed(tape):
            g_lr = g_lr - (tbar * g).sum()
            if handle is not None:
                v = unflatten_params(tbar * (-self.syn_lr * share), self.num_classes)
                dx, hv = self.ops.hvp(handle, v)
                self._push(ctx, dx)
                hv = flatten_params(hv)
            else:
                hv = torch.zeros_like(tbar)
            tbar = tbar + self._allreduce(hv)
        self.last_grads = self._finish(update) + (g_lr,)
        self.last_tape = tape if self.keep_tape else None
        if update:
            mu = self.LR_MOMENTUM
            self.lr_buf = g_lr.clone() if self.steps_done == 0 else mu * self.lr_buf + g_lr
            self.syn_lr = torch.clamp(self.syn_lr - self.lr_lr * self.lr_buf, min=0.001)         # .clip(min=0.001), :269
            self.steps_done += 1
        return grand


class S2DMTTTrainer(MTTTrainer):
    """MTT over static + dynamic memories ("MTT+Ours", distill_s2d_ms.py:189-300; BASELINE config 5): the clip of item
    i = (class, v) of a student batch is hallucinator(static[spc*class + 2v + r_s], dynamic[class, 2v + r_d]) with two
    fresh randint(2) draws per item and step (:248-256); the grand loss is back-propagated through the unrolled
    student steps AND the hallucinator to the dynamic memories, the hallucinator's 327 parameters and (unless
    ``--no_train_static``) the static memories, each under SGD(momentum .95) (:107-110); syn_lr under SGD(lr_lr, momentum
    .9) (:110) with the .clip(min=0.001) of :291.  The student batch of a step is split over ranks like MTTTrainer's;
    memory / hallucinator gradients are summed over ranks once per iteration."""

    LR_MOMENTUM = 0.9

    def __init__(self, ops, num_classes: int, vpc: int, spc: int, dpc: int, static_syn: torch.Tensor, dynamic_syn: torch.Tensor,
                 hal_w: torch.Tensor, hal_b: torch.Tensor, syn_lr: float, lr_dynamic: float, lr_hal: float, lr_lr: float,
                 syn_steps: int, batch_syn: int, expert_epochs: int, max_start_epoch: int, lr_static: float = 0.0,
                 train_static: bool = False, momentum: float = 0.95, rank: int = 0, world: int = 1):
        self.vpc, self.spc, self.dpc = int(vpc), int(spc), int(dpc)
        self.static = static_syn.contiguous()
        self.dynamic = dynamic_syn.reshape((-1,) + tuple(dynamic_syn.shape[2:])).contiguous()      # rows (class, dpc) flattened
        super().__init__(ops, num_classes, self.dynamic, None, syn_lr, lr_dynamic, lr_lr, syn_steps, batch_syn, expert_epochs,
                         max_start_epoch, momentum=momentum, rank=rank, world=world)
        self.image_syn = self.dynamic                  # device / bookkeeping handle of the base class
        self.hal_w, self.hal_b = hal_w.clone().contiguous(), hal_b.clone().contiguous()
        self.lr_dynamic, self.lr_hal, self.lr_static = float(lr_dynamic), float(lr_hal), float(lr_static)
        self.train_static = bool(train_static)
        self.buf_d = self.buf
        self.buf_w, self.buf_b = torch.zeros_like(self.hal_w), torch.zeros_like(self.hal_b)
        self.buf_s = torch.zeros_like(self.static) if train_static else None
        self.draws = None        # optional fixed randint draws: [step] -> (dynamic (batch,), static (batch,)) for the WHOLE batch

    def _num_items(self) -> int:
        return self.num_classes * self.vpc

    def _begin(self, dev) -> None:
        self._g_dyn = torch.zeros_like(self.dynamic)
        self._g_stat = torch.zeros_like(self.static) if self.train_static else None
        self._g_w = torch.zeros_like(self.hal_w)
        self._g_b = torch.zeros_like(self.hal_b)
        self._pos = {}

    def indices(self, these: torch.Tensor, step: int, it: int):
        """distill_s2d_ms.py:248-252 for the items ``these`` of the student batch of ``step`` (the draws are made for the
        whole batch from a per-(iteration, step) seed, so every sharding composes the same clips)."""
        n = int(these.numel())
        if self.draws is not None:
            dd, ds = (torch.as_tensor(np.asarray(t), dtype=torch.int64) for t in self.draws[step])
        else:
            r = np.random.default_rng([it, step, 23])
            dd, ds = torch.as_tensor(r.integers(0, 2, n)), torch.as_tensor(r.integers(0, 2, n))
        label = these // self.vpc
        idx = these % self.vpc
        dynamic_idx = label * self.dpc + 2 * idx + dd.to(these.device)       # row of the flattened (class, dpc) memory
        static_idx = self.spc * label + 2 * idx + ds.to(these.device)
        return label, static_idx, dynamic_idx

    def _clips(self, batch: torch.Tensor, sel: slice, step: int, it: int):
        label, sidx, didx = (t[sel].contiguous() for t in self.indices(batch, step, it))
        x = self.ops.hallucinate(self.static, self.dynamic, sidx, didx, self.hal_w, self.hal_b)
        return x, label, (sidx, didx)

    def _push(self, ctx, dx: torch.Tensor) -> None:
        sidx, didx = ctx
        g_dyn, g_stat, g_w, g_b = self.ops.hallucinate_backward(dx.contiguous(), self.static, self.dynamic, sidx, didx, self.hal_w,
                                                                self.train_static)
        self._g_dyn += g_dyn
        self._g_w += g_w.reshape(self._g_w.shape)
        self._g_b += g_b
        if self.train_static:
            self._g_stat += g_stat

    def _finish(self, update: bool):
        if collectives_on(self.world):
            import torch.distributed as dist
            flat = torch.cat([self._g_w.reshape(-1), self._g_b.reshape(-1)])
            _all_reduce(flat)
            self._g_w, self._g_b = flat[:self._g_w.numel()].view_as(self._g_w).contiguous(), flat[self._g_w.numel():].contiguous()
            _all_reduce(self._g_dyn)
            if self.train_static:
                _all_reduce(self._g_stat)
        if update:
            first = self.steps_done == 0
            self.ops.sgd(self.dynamic, self.buf_d, self._g_dyn, self.lr_dynamic, self.momentum, first)
            self.ops.sgd(self.hal_w, self.buf_w, self._g_w, self.lr_hal, self.momentum, first)
            self.ops.sgd(self.hal_b, self.buf_b, self._g_b, self.lr_hal, self.momentum, first)
            if self.train_static:
                self.ops.sgd(self.static, self.buf_s, self._g_stat, self.lr_static, self.momentum, first)
        return (self._g_dyn, self._g_w, self._g_b, self._g_stat)
