"""Device-side execution of ConvNet3D.embed (forward and input gradient) on MI355X.

`EmbedEngine` owns, for one clip geometry / precision / device: the uploaded tile programs
(plan.py), the packed MFMA operands of the current network weights, and the activation
workspaces, and drives the C ABI (hip.py).  It mirrors what the reference's
``ConvNet3D.embed`` (networks.py:747-751) plus autograd's backward to the input do, for
parameters that are frozen (distill_baseline.py:336-337).
"""
from __future__ import annotations

import ctypes
import threading
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import hip
from . import plan as P


# Measurement hook (bench.py): a list here makes every tile-program launch append
# (program name, precision id, algorithmic FLOP of the launch, start event, end event) -- HIP events on the launch stream.
LAUNCH_PROFILE: Optional[list] = None


class _DevPlan:
    """A ConvPlan with its tables resident on the device and a reusable parameter block."""

    def __init__(self, plan: P.ConvPlan, device, prec: int):
        self.plan = plan
        desc, tables = plan.flat_tables()
        self.type_desc = torch.from_numpy(desc.copy()).to(device)
        self.tables = torch.from_numpy(tables.copy()).to(device)
        self.boxes = torch.from_numpy(plan.device_boxes().copy()).to(device)
        self.widx = torch.from_numpy(plan.widx.reshape(-1).copy()).to(device)   # empty for wgrad plans (B = packed dy)
        gt = plan.gather_table()
        self.gather = torch.from_numpy(gt.copy()).to(device)
        self.n_w = int(self.widx.numel())
        planes = 2 if hip.is_x3(prec) else 1
        self.prec = prec
        self.wpk = self._new_wpk(planes, device)
        self.wpk_d: Optional[torch.Tensor] = None          # dithered operand sets (pack_dither)
        self.range_stats: Optional[torch.Tensor] = None    # emit_lo = 2 launches: [saturated outputs, float bits of max |output|] (vd_hip.h)
        self._slots: Dict[int, tuple] = {}                 # buffer sets of use_slot()
        self._slot = 0
        p = hip.VdConvParams()
        p.type_desc = self.type_desc.data_ptr(); p.tables = self.tables.data_ptr(); p.boxes = self.boxes.data_ptr()
        p.gather = self.gather.data_ptr(); p.gather_stride = int(gt.shape[1])
        p.nbox = plan.nbox; p.ncl = plan.ncl
        p.CC, p.S = plan.CC, plan.S
        p.src_clip_stride4, p.src_chunk_stride4 = plan.clip_stride4, plan.chunk_stride4
        p.NT, p.MW, p.MTW, p.NTW = plan.NT, plan.MW, plan.MTW, plan.NTW
        mt_max = max(t.mt for t in plan.types)
        p.mt_valid = mt_max if (plan.NTW == 2 and mt_max < plan.MW * plan.MTW and os.environ.get("VD_SKIP_PAD", "1") == "1") else 0
        p.epi, p.pool_t, p.relu = plan.epi, plan.pool_t, int(plan.relu)
        p.n_out, p.n_stride = plan.n_out, plan.n_stride
        p.out_clip_stride = plan.out_clip_stride
        p.out_chunk_stride, p.out_t_stride = plan.out_chunk_stride, plan.out_t_stride
        p.pair_flip = plan.pair_flip
        p.lds_plane_bytes = int(gt.shape[1]) * 16
        p.ntypes = len(plan.types)
        p.persist = int(os.environ.get("VD_PERSIST", str(P.BOX_WALK_GENERATIONS)))
        if os.environ.get("VD_NO_ALT") == "1":       # A/B: hi+lo programs without the per-chunk sign alternation of their accumulation (conv_mfma.hip ALT)
            p.persist |= 0x100000
        p.tab_ofs[0], p.tab_ofs[1], p.tab_ofs[2] = int(desc[0][7]), int(desc[0][8]), int(desc[0][9])
        self.zero = torch.zeros(64, dtype=torch.uint8, device=device)
        p.zero_slot = self.zero.data_ptr()
        p.prec = prec
        p.wpk = self.wpk.data_ptr(); p.w_plane_stride = self.n_w
        p.w_box_stride = plan.w_box_stride; p.atomic = int(plan.atomic)
        self.col_off = None if plan.col_off is None else torch.from_numpy(plan.col_off.copy()).to(device)
        p.col_off = 0 if self.col_off is None else self.col_off.data_ptr()
        self.params = p
        # first layer, single-pass formats, 2 x 2-wave layout, pixel-row source: the kernel with the layer's B fragments resident in
        # registers (conv0_breg_kernel: one eight-wave workgroup per CU whose two groups alternate K loop / everything else).
        # VD_L0_BREG: 5 (default) = on frame-tile programs every A fragment is read from LDS once for the tiles it serves;
        # 4 = the plain K loop, one read per MFMA; 0 = the generic tile-program kernel.  All bitwise equal
        # (tests/test_gpu_embed.py::test_first_layer_kernel_variants_are_bitwise_equal).
        self.breg_variant = int(os.environ.get("VD_L0_BREG", "5"))
        self.breg_ok = bool(self.breg_variant in (4, 5) and not hip.is_x3(prec)
                            and plan.epi == P.EPI_POOL_CL and plan.pool_t == 1 and plan.CC == 1 and plan.ncl == 1 and plan.NTW == 1
                            and (plan.NT, plan.MW, plan.MTW, plan.S) == (2, 2, 4, 32) and len(plan.types) == 1
                            and plan.relu and plan.row_source()[1] > 0)
        if plan.row_source()[1] > 0:
            p.src_planes, p.src_rows = plan.row_source()

    def _new_wpk(self, planes: int, device) -> torch.Tensor:
        """[planes, n_w] 16-bit packed operands.  VD_PREC_F16C8 programs read their B operands through running pointers, up to six K
        steps past a channel chunk -- i.e. past the END of the buffer for the last chunk: 32 KB of (zeroed) slack behind it."""
        slack = 16384 if self.prec == hip.PREC["f16c8"] else 0
        flat = torch.zeros(planes * self.n_w + slack, dtype=torch.int16, device=device) if slack else \
            torch.empty(planes * self.n_w, dtype=torch.int16, device=device)
        return flat[:planes * self.n_w].view(planes, self.n_w)

    def use_slot(self, k: int) -> None:
        """Switch to the k-th set of packed-operand buffers (``wpk`` and the dithered sets ``wpk_d``), allocated on first use.  Two
        sets let the operands of step i + 1 be packed on a side stream while the launches of step i still read theirs
        (EmbedEngine.set_weights(slot=), distill.DMTrainer)."""
        self._slots[self._slot] = (self.wpk, self.wpk_d)
        if k not in self._slots:
            self._slots[k] = (self._new_wpk(self.wpk.shape[0], self.wpk.device), None)
        self.wpk, self.wpk_d = self._slots[k]
        self._slot = k

    def pack(self, w: torch.Tensor) -> None:
        assert w.dtype == torch.float32 and w.is_contiguous()
        lo = self.wpk[1] if self.wpk.shape[0] == 2 else None
        if _PACK.queue is not None and self.n_w > 0:          # inside ``batched_packs()``: one launch for all of them at its exit
            _PACK.queue.append((w, self.widx, self.n_w, self.wpk[0], lo, self.prec))
            return
        hip.check(hip.lib().vd_pack_weights(hip.ptr(w), hip.ptr(self.widx), ctypes.c_int64(self.n_w),
                                            hip.ptr(self.wpk[0]), hip.ptr(lo), self.prec, hip.stream_ptr(w.device)),
                  "vd_pack_weights")

    def pack_c8(self, w: torch.Tensor, scales: torch.Tensor) -> None:
        """Operand planes of a VD_PREC_F16C8 program (vd_pack_weights_c8): fp16 fragments + the fp8 fragments of W_hi and W_lo;
        ``scales`` (>= 6 floats of device scratch) receives the two E8M0 codes ``run(..., out_scale=scales)`` hands to the kernel."""
        assert w.dtype == torch.float32 and w.is_contiguous() and self.wpk.shape[0] == 2
        pl = self.plan
        sets = self.n_w // (pl.CC * pl.S * pl.NT * 512)       # (position-tile programs: one operand set per box, chunk-major inside)
        hip.check(hip.lib().vd_pack_weights_c8(hip.ptr(w), ctypes.c_int64(w.numel()), hip.ptr(self.widx), sets * pl.CC, pl.S, pl.NT,
                                               hip.ptr(self.wpk[0]), hip.ptr(self.wpk[1]), hip.ptr(scales), hip.stream_ptr(w.device)),
                  "vd_pack_weights_c8")

    def pack_dither(self, w: torch.Tensor, groups: int) -> None:
        """``groups`` dithered single-pass operand sets (vd_pack_weights_dither); ``run(..., group=g)`` multiplies by set g."""
        assert w.dtype == torch.float32 and w.is_contiguous() and self.wpk.shape[0] == 1
        if self.wpk_d is None or self.wpk_d.shape[0] != groups:
            self.wpk_d = torch.empty((groups, self.n_w), dtype=torch.int16, device=w.device)
        hip.check(hip.lib().vd_pack_weights_dither(hip.ptr(w), hip.ptr(self.widx), ctypes.c_int64(self.n_w), int(groups),
                                                   hip.ptr(self.wpk_d), self.prec, hip.stream_ptr(w.device)), "vd_pack_weights_dither")

    def run(self, src: torch.Tensor, src_plane_slots: int, bias: Optional[torch.Tensor], dst_ptr: int,
            dst_plane_stride: int, argmax: Optional[torch.Tensor], nclips: int, out_scale: Optional[torch.Tensor] = None,
            wpk: Optional[torch.Tensor] = None, w_plane_elems: int = 0, clip_index: Optional[torch.Tensor] = None,
            group: Optional[int] = None, set_clips: int = 0, emit_lo: bool = False, launch: bool = True) -> None:
        """``launch=False`` only fills the parameter block (``run_together`` then sends several programs out as one launch)."""
        p = self.params
        p.emit_lo = int(emit_lo)        # single-pass program, staged pooled epilogue: 1 = also write the fp16 low plane (dst_plane_stride
        #                                 behind), 2 = write fp8 low parts there instead (for a VD_PREC_F16C8 consumer)
        if int(emit_lo) == 2 and _RANGE_MONITOR:      # ... whose fixed scalings hold for a RANGE of output magnitudes: the launch records what it saw
            if self.range_stats is None:
                self.range_stats = torch.zeros(2, dtype=torch.int32, device=src.device)
            p.range_stats = self.range_stats.data_ptr()
        else:
            p.range_stats = 0
        p.clip_index = 0 if clip_index is None else clip_index.data_ptr()
        p.out_scale = 0 if out_scale is None else out_scale.data_ptr()
        p.w_set_clips = 0
        if wpk is not None:     # B operand supplied per call (weight-gradient programs)
            p.wpk = wpk.data_ptr(); p.w_plane_stride = w_plane_elems
        elif group is not None:  # one of the dithered operand sets of pack_dither
            p.wpk = self.wpk_d[group].data_ptr(); p.w_plane_stride = self.n_w
        elif set_clips > 0:      # all dithered sets in one launch: clips [s * set_clips, (s+1) * set_clips) multiply by set s
            assert set_clips % self.plan.ncl == 0
            p.wpk = self.wpk_d.data_ptr(); p.w_plane_stride = self.n_w; p.w_set_clips = int(set_clips)
        else:
            p.wpk = self.wpk.data_ptr(); p.w_plane_stride = self.n_w
        p.src = src.data_ptr(); p.src_plane_stride4 = src_plane_slots * 4
        p.bias = 0 if bias is None else bias.data_ptr()
        p.dst = dst_ptr; p.dst_plane_stride = dst_plane_stride
        p.argmax = 0 if argmax is None else argmax.data_ptr()
        p.nclips = nclips
        if not launch:
            return
        prof = LAUNCH_PROFILE
        if prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if self.breg_ok and argmax is None and (not p.dbg or os.environ.get("VD_BREG_DBG") == "1"):
            persist = p.persist
            if self.breg_variant == 4:
                p.persist = persist | 0x80000          # (VdConvParams.persist bit 19: the plain K loop on a frame-tile program)
            try:
                hip.check(hip.lib().vd_conv0_breg(ctypes.byref(p), hip.stream_ptr(src.device)), "vd_conv0_breg(%s)" % self.plan.name)
            finally:
                p.persist = persist
        else:
            hip.check(hip.lib().vd_conv_mfma(ctypes.byref(p), hip.stream_ptr(src.device)), "vd_conv_mfma(%s)" % self.plan.name)
        if prof is not None:
            e1.record()
            prof.append((self.plan.name, self.prec, 2.0 * self.plan.meta.get("macs_per_unit", 0) * nclips, e0, e1))


def GRAD_TARGET() -> float:
    """max |g| * scale of a gradient operand lies in [target / 2, target) (vd_absmax_scale) before its 16-bit split: fp16's largest
    number is 65504, an fp16 PAIR carries 22 bits for elements down to 2^-2, i.e. 2^-12 of the largest at 1024 (VD_GRAD_TARGET: A/B)."""
    return float(os.environ.get("VD_GRAD_TARGET", "1024"))


_RANGE_MONITOR = os.environ.get("VD_RANGE_MONITOR", "1") == "1"      # (0: A/B of the monitor's cost in the level-1 epilogue)


class _PackState(threading.local):       # per host thread: packs queued by one thread are flushed on that thread's current stream
    queue: Optional[list] = None


_PACK = _PackState()


class batched_packs:
    """``with batched_packs():`` -- the ``_DevPlan.pack`` calls inside are collected and issued as ONE launch per 24 of them at
    the exit (vd_pack_weights_multi): a training or trajectory-matching step packs the same few weight tensors for a dozen tile
    programs, each a 5 - 10 us launch of its own otherwise.  Bitwise the same operands.  The packed buffers must not be read
    before the exit; nests (the outermost exit launches).  ``VD_PACK_BATCH=0``: every pack launches by itself.  The queue is
    per host THREAD (two threads driving two engines do not flush each other's packs); within a thread the flush goes to the
    stream that is current at the outermost exit, which is where the packs themselves would have gone."""

    def __enter__(self):
        self.outer = _PACK.queue is not None or os.environ.get("VD_PACK_BATCH", "1") != "1"
        if not self.outer:
            _PACK.queue = []
        return self

    def __exit__(self, *exc):
        if self.outer:
            return False
        q, _PACK.queue = _PACK.queue, None
        if exc[0] is None:
            flush_packs(q)
        return False


def flush_packs(q) -> None:
    for i in range(0, len(q), hip.VD_PACK_MAX):
        part = q[i:i + hip.VD_PACK_MAX]
        b = hip.VdPackBatch()
        b.nseg = len(part)
        for k, (w, widx, n, hi, lo, prec) in enumerate(part):
            sg = b.seg[k]
            sg.w, sg.widx, sg.n = w.data_ptr(), widx.data_ptr(), int(n)
            sg.out_hi, sg.out_lo, sg.prec = hi.data_ptr(), (0 if lo is None else lo.data_ptr()), int(prec)
        hip.check(hip.lib().vd_pack_weights_multi(ctypes.byref(b), hip.stream_ptr(part[0][0].device)), "vd_pack_weights_multi")


def run_together(plans: Sequence["_DevPlan"], *args, **kwargs) -> None:
    """``dp.run(*args, **kwargs)`` for every plan of ``plans`` (the parity classes of one input-gradient pass: same source,
    same destination tensor, disjoint output positions), programs of the same instantiation in ONE launch (vd_conv_mfma_multi) --
    a parity class alone starts too few workgroups to fill the chip at small batches.  Bitwise the separate launches' results
    (tests/test_gpu_embed.py::test_parity_classes_in_one_launch_are_bitwise_equal); ``VD_MULTI_LAUNCH=0`` launches one by one."""
    plans = list(plans)
    if len(plans) < 2 or os.environ.get("VD_MULTI_LAUNCH", "1") != "1":
        for dp in plans:
            dp.run(*args, **kwargs)
        return
    groups: Dict[tuple, list] = {}
    for dp in plans:
        pl, p = dp.plan, dp.params
        so = bool(p.atomic or p.select or p.src_split_cc > 0)
        plain = pl.NTW in (0, 1) and pl.MTW in (2, 4, 7, 8) and not p.w_box_stride and not p.dbg and (not so or hip.is_x3(dp.prec))
        key = (dp.prec, pl.MTW, pl.NT, pl.MW, so) if plain else ("single", id(dp))
        groups.setdefault(key, []).append(dp)
    for key, grp in groups.items():
        for i in range(0, len(grp), 4):
            part = grp[i:i + 4]
            if len(part) == 1 or key[0] == "single":
                for dp in part:
                    dp.run(*args, **kwargs)
                continue
            for dp in part:
                dp.run(*args, launch=False, **kwargs)
            arr = (ctypes.POINTER(hip.VdConvParams) * len(part))(*[ctypes.pointer(dp.params) for dp in part])
            prof = LAUNCH_PROFILE
            if prof is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            dev = part[0].tables.device
            hip.check(hip.lib().vd_conv_mfma_multi(arr, len(part), hip.stream_ptr(dev)),
                      "vd_conv_mfma_multi(%s)" % ",".join(dp.plan.name for dp in part))
            if prof is not None:
                e1.record()
                nclips = int(part[0].params.nclips)
                name = os.path.commonprefix([dp.plan.name for dp in part]).rstrip("_") + "_x%d" % len(part)
                prof.append((name, part[0].prec, sum(2.0 * dp.plan.meta.get("macs_per_unit", 0) * nclips for dp in part), e0, e1))


def dither_groups(n: int, prec: str) -> int:
    """Number of dithered weight sets (a power of two, >= 4) a batch of ``n`` real clips whose MEAN feature is what matters
    is dealt to, or 0: single-pass formats only, ``VD_REAL_DITHER`` (default 8; < 4 turns it off) is the most, and the
    groups must divide the batch evenly.  Below 4 groups the value pass on the synthetic side is the better remedy
    (tests/sim_dither_tool.py: 2 groups leave 1.1e-4 |f|, the value pass 8e-5, 8 groups 3e-5)."""
    want = int(os.environ.get("VD_REAL_DITHER", "8"))
    if prec not in ("f16", "bf16") or want < 4 or n < 4:
        return 0
    g = 1
    while g * 2 <= min(want, n) and n % (g * 2) == 0:
        g *= 2
    return g if g >= 4 else 0


def round_weights(params: Sequence[torch.Tensor], prec: str, levels: Sequence[int] = (0, 1, 2)) -> List[torch.Tensor]:
    """[w0, b0, w1, b1, w2, b2, ...] with every conv WEIGHT replaced by (float) rn16(w) in the operand format
    ``prec`` (biases are added in fp32 and stay as they are).  A mixed-precision step -- real clips single-pass
    f16, synthetic clips hi+lo f16 pairs, f16 input gradient -- that is fed these weights multiplies by the SAME
    weights in every pass: it is the exact step of the network rn16(W), so weight rounding cannot appear as a
    bias between mean f_real and mean f_syn (that bias is what grows, relative to the gradient, as the
    distillation converges and the difference of the means shrinks)."""
    out, L = [], hip.lib()
    for i, p in enumerate(params):
        if i % 2 == 0 and p.dim() == 5 and i < 6 and (i // 2) in levels:     # (``levels``: conv levels whose weights are rounded)
            p = p.detach().to(torch.float32).contiguous()
            q = torch.empty_like(p)
            hip.check(L.vd_round_operand(hip.ptr(p), ctypes.c_int64(p.numel()), hip.PREC[prec], hip.ptr(q),
                                         hip.stream_ptr(p.device)), "vd_round_operand")
            out.append(q)
        else:
            out.append(p)
    return out


class EmbedEngine:
    def __init__(self, geo: P.NetGeometry, prec: str = "bf16x3", device="cuda:0", chunk: int = 256,
                 prec_bwd: Optional[str] = None, ntw0: Optional[int] = None, batch_hint: Optional[int] = None,
                 last_hilo: bool = False, bwd0_small: bool = False):
        """``last_hilo`` (single-pass engines, forward without kept arg-max): the LAST conv level runs in the hi+lo format of the
        same 16-bit type -- level 1's program also writes the low plane of its pooled outputs (VdConvParams.emit_lo) and level 2
        multiplies hi+lo activations by hi+lo weights (3 MFMAs per product on 5.6 % of the network's FLOPs).  That removes two
        of the six rounding sources of a single-pass forward (level 1's output rounding, level 2's weight rounding; DESIGN
        section 2, tests/sim_error_budget_tool.py)."""
        if not torch.cuda.is_available():
            raise RuntimeError("EmbedEngine needs a HIP device (no CPU fallback)")
        hip.lib()
        self.geo = geo
        self.prec_name = prec
        self.prec = hip.PREC[prec]
        self.planes = 2 if hip.is_x3(self.prec) else 1
        self.device = torch.device(device)
        self.chunk = int(chunk)
        # two N tiles per wave (half the LDS reads per MFMA): layer-1/2 forward programs whose boxes fill 8 M
        # tiles, and the first layer in the single-pass formats (measured: x1 +7 %, x3 -6 % -> x3 keeps one)
        self.ntw = int(os.environ.get("VD_NTW", "2"))
        if ntw0 is None:
            # (x1 formats: the register-resident-B kernel runs the 2x2-wave one-N-tile layout)
            breg = os.environ.get("VD_L0_BREG", "5") in ("4", "5")
            ntw0 = int(os.environ.get("VD_NTW0", "1" if (hip.is_x3(self.prec) or self.ntw != 2 or breg) else "2"))
        bal = (self.ntw == 2 and not hip.is_x3(self.prec) and os.environ.get("VD_BALANCED", "1") == "1")
        self.batch_hint = batch_hint      # typical clips per launch: small batches get latency-oriented programs
        net = P.plan_network(geo, ntw=self.ntw, ntw0=ntw0, balanced=bal, batch_hint=batch_hint, bwd0_small=bwd0_small)
        self.dims = net["dims"]
        self.fwd = [_DevPlan(pl, self.device, self.prec) for pl in net["fwd"]]
        self.fwd2x = None
        self.last_c8 = False
        self._c8_scale_slots: Dict[int, torch.Tensor] = {}    # set_weights(slot=): the fp8-corrected program's scale table per buffer set
        if last_hilo:
            if hip.is_x3(self.prec):
                raise ValueError("last_hilo is an option of the single-pass formats (%s already carries hi+lo planes)" % prec)
            import dataclasses
            pl2 = net["fwd"][2]
            if os.environ.get("VD_L2X_PLAN", "small") == "small" and pl2.NTW == 2 and pl2.MTW == 4 and pl2.ncl % 2 == 0:
                # The hi+lo program keeps two operand planes of its patch in LDS: with the single-pass program's boxes (two clips, 4 M
                # tiles x 2 N tiles per wave, 2 x 58 KB) only one workgroup fits a CU.  One clip per box (4 M tiles x 1 N tile per
                # wave, 2 x 29 KB) lets two share it: 0.74 -> 0.68 ms per 512 clips, 4.53 -> 4.27 ms per launch in the bench (same
                # box); same K order per output, bitwise the same features.
                d2 = self.dims[2]
                try:
                    alt = P.plan_forward_cl("fwd2", d2[0], d2[1], d2[2], d2[3], d2[4], d2[11], feat_out=True, ntw=1, mtw_options=(4,),
                                            step_multiple=4 if last_hilo == "c8" else 1)
                    if alt.rows_total <= pl2.rows_total and 2 * alt.gather_table().shape[1] * 16 <= 72 * 1024:
                        pl2 = alt
                except ValueError:
                    pass
            # ``last_hilo == "c8"`` (f16 only): the last level as VD_PREC_F16C8 -- fp16 main product, the two hi+lo correction products
            # on the block-scaled fp8 matrix instruction once per four K steps (two MFMA-equivalents per product instead of three);
            # level 1 then emits fp8 low parts (emit_lo = 2) instead of the fp16 low plane.  Needs the one-clip 4 x 1-tile program.
            self.last_c8 = (last_hilo == "c8")
            if self.last_c8:
                # POSITION TILES (plan.plan_forward_pos): rows = (clip, frame) pairs of one output position, so that the taps outside
                # the input grid -- half of them at 7 x 7, two thirds at 4 x 4 -- are skipped per tile; VD_C8_POS=0 keeps the
                # row-major program, which needs the one-clip 4 x 1-tile decomposition (not every geometry has it)
                d2 = self.dims[2]
                pos = None
                if prec == "f16" and os.environ.get("VD_C8_POS", "1") == "1":
                    try:
                        pos = P.plan_forward_pos("fwd2_pos", d2[0], d2[1], d2[2], d2[3], d2[4], d2[11])
                    except (ValueError, AssertionError):
                        pos = None
                if pos is not None:
                    pl2 = pos
                elif prec == "f16" and pl2.NTW == 1 and pl2.MTW == 4 and pl2.S % 4 == 0:
                    pl2 = P.with_skip_table(dataclasses.replace(pl2, types=[dataclasses.replace(t) for t in pl2.types]))
                else:
                    raise ValueError("last_hilo='c8' needs the f16 format and either position tiles (frames dividing 32) or the one-clip "
                                     "4 x 1-tile last-level program (geometry %s)" % (geo,))
            self.fwd2x = _DevPlan(dataclasses.replace(pl2, name="fwd2_c8" if self.last_c8 else "fwd2_hilo"), self.device,
                                  hip.PREC["f16c8"] if self.last_c8 else hip.PREC[prec + "x3"])
            if self.last_c8:
                self.c8_scales = torch.zeros(8, dtype=torch.float32, device=self.device)
                if pl2.epi == P.EPI_POS_FEAT and os.environ.get("VD_C8_BOXMAJOR", "0") == "1":
                    self.fwd2x.params.persist |= 0x80000       # window-major box order (VdConvParams.persist bit 19)
        # operand precision of the input-gradient passes (default: same as the forward)
        self.prec_bwd = hip.PREC[prec_bwd] if prec_bwd else self.prec
        self.planes_bwd = 2 if hip.is_x3(self.prec_bwd) else 1
        self.bwd = [[_DevPlan(pl, self.device, self.prec_bwd) for pl in layer] for layer in net["bwd"]]
        self.num_feat = geo.num_feat
        self._weights: Optional[List[torch.Tensor]] = None
        self._bwd_packed = False
        self._ws: Dict[str, torch.Tensor] = {}
        self.profile = None   # list -> (layer, clips, start_event, end_event) per forward launch

    # ------------------------------------------------------------------------------------
    def _buf(self, name: str, shape, dtype) -> torch.Tensor:
        t = self._ws.get(name)
        n = int(np.prod(shape))
        if t is None or t.numel() < n or t.dtype != dtype:
            t = torch.empty(n, dtype=dtype, device=self.device)
            self._ws[name] = t
        return t[:n].view(*shape)

    def set_weights(self, params: Sequence[torch.Tensor], quantize: Optional[str] = None, dither: int = 0,
                    quantize_levels: Sequence[int] = (0, 1, 2), slot: Optional[int] = None) -> None:
        """params = [w0, b0, w1, b1, w2, b2] fp32 on the device (ConvNet3D.features order).  ``quantize`` ('f16' /
        'bf16'): round the weight tensors of ``quantize_levels`` to that operand format first (``round_weights``; the value
        pass rounds exactly the levels the real side multiplies by plain rn16 weights -- not the last one when that runs in
        hi+lo pairs with exact weights).  ``dither`` = G >= 2 (single-pass engines): additionally pack G dithered operand
        sets, selected by ``forward(..., group=g)``."""
        ws = [p.detach().to(self.device, torch.float32).contiguous() for p in params[:6]]
        if quantize is not None:
            ws = round_weights(ws, quantize, quantize_levels)
        self._weights = ws
        if slot is not None:      # (forward operands only: the packed buffers of slot ``slot``; the launches that follow read them)
            for dp in self.fwd + ([self.fwd2x] if self.fwd2x is not None else []):
                dp.use_slot(slot)
            if self.last_c8:
                if slot not in self._c8_scale_slots:
                    self._c8_scale_slots[slot] = torch.zeros(8, dtype=torch.float32, device=self.device)
                self.c8_scales = self._c8_scale_slots[slot]
        with batched_packs():
            for li in range(3):
                if li == 2 and self.fwd2x is not None:      # the last level multiplies by the exact hi+lo weights: nothing to dither
                    if self.last_c8:
                        self.fwd2x.pack_c8(ws[4], self.c8_scales)
                    else:
                        self.fwd2x.pack(ws[4])
                    continue
                self.fwd[li].pack(ws[2 * li])
                if dither >= 2:
                    self.fwd[li].pack_dither(ws[2 * li], dither)
        self._dither = int(dither) if dither >= 2 else 0
        self._bwd_packed = False

    def _pack_bwd(self) -> None:
        if not self._bwd_packed:
            with batched_packs():
                for li in range(3):
                    for dp in self.bwd[li]:
                        dp.pack(self._weights[2 * li])
            if os.environ.get("VD_BWD_X2_SIM") == "w":
                # measurement knob: the NUMERICS of a two-MFMA input gradient (g_hi + g_lo) x W_hi -- the low
                # plane of the weights dropped -- at the cost of the three-MFMA program
                if _PACK.queue:            # (inside an outer batched_packs(): the queued packs must land before their low planes are cleared)
                    flush_packs(_PACK.queue)
                    del _PACK.queue[:]
                for li in range(3):
                    for dp in self.bwd[li]:
                        if dp.wpk.shape[0] == 2:
                            dp.wpk[1].zero_()
            self._bwd_packed = True

    # ------------------------------------------------------------------------------------
    def pool_rows(self, pool: torch.Tensor, step: int = 256) -> torch.Tensor:
        """The whole (static) clip pool converted ONCE to this engine's first-layer operand format
        (16-bit pixel rows, [planes][N * slots per clip][8]); ``forward(..., rows=)`` then reads batches
        straight out of it through the index -- no per-step conversion of the real clips."""
        g = self.geo
        N = int(pool.shape[0])
        rowp = P.pix_row_pitch(g.width)
        per = g.frames * 3 * g.height * (rowp // 8)
        rows = torch.empty((self.planes, N * per, 8), dtype=torch.int16, device=self.device)
        L, st = hip.lib(), hip.stream_ptr(self.device)
        for i in range(0, N, step):
            nb = min(step, N - i)
            lo = rows[1, i * per:] if self.planes == 2 else None
            hip.check(L.vd_pix2rows(hip.ptr(pool[i:]), hip.ptr(None), ctypes.c_int64(nb), g.frames, g.height, g.width,
                                     hip.ptr(rows[0, i * per:]), hip.ptr(lo), self.prec, st), "vd_pix2rows")
        return rows

    def forward(self, x: torch.Tensor, keep: bool = False, index: Optional[torch.Tensor] = None,
                rows: Optional[torch.Tensor] = None, group: Optional[int] = None):
        """x (B,T,3,H,W) fp32 on the device -> features (B, num_feat) fp32.  With ``keep`` the
        pooling arg-max of every layer is retained and returned as a handle for ``backward``
        (several forwards may be outstanding before their backwards, as in the reference's
        per-class loop, distill_baseline.py:344-354).  ``group``: multiply by dithered operand set g of ``set_weights(dither=G)``."""
        assert self._weights is not None, "set_weights() first"
        assert group is None or 0 <= group < getattr(self, "_dither", 0), "set_weights(dither=G) first"
        g = self.geo
        assert x.dim() == 5 and tuple(x.shape[1:]) == (g.frames, g.channel, g.height, g.width), x.shape
        x = x.detach().to(torch.float32).contiguous()
        B = x.shape[0] if index is None else int(index.numel())
        if index is not None:   # batch clip b = x[index[b]] (gather fused into the slot conversion)
            index = index.to(self.device, torch.int64).contiguous()
        feats = torch.empty((B, self.num_feat), dtype=torch.float32, device=self.device)
        saved = []
        L = hip.lib()
        st = hip.stream_ptr(self.device)
        rowp = P.pix_row_pitch(g.width)
        per1 = int(np.prod(self.fwd[0].plan.out_shape[:-1]))
        per2 = int(np.prod(self.fwd[1].plan.out_shape[:-1]))
        for c0 in range(0, B, self.chunk):
            nb = min(self.chunk, B - c0)
            n_slots0 = nb * g.frames * 3 * g.height * (rowp // 8)      # 16-byte units of the padded pixel rows
            cidx = None
            if rows is not None:      # batch = index into the resident, already converted pool
                assert index is not None and rows.shape[0] == self.planes
                slots0, n_slots0, cidx = rows, int(rows.shape[1]), index[c0:]
            else:
                slots0 = self._buf("slots0", (self.planes, n_slots0, 8), torch.int16)
                lo = slots0[1] if self.planes == 2 else None
                xin = x[c0:] if index is None else x
                hip.check(L.vd_pix2rows(hip.ptr(xin), hip.ptr(None if index is None else index[c0:]),
                                         ctypes.c_int64(nb), g.frames, g.height, g.width,
                                         hip.ptr(slots0[0]), hip.ptr(lo), self.prec, st), "vd_pix2rows")
            n1, n2 = nb * per1, nb * per2
            hilo = self.fwd2x is not None
            assert not (hilo and keep), "last_hilo engines have no kept-arg-max forward"
            act1 = self._buf("act1", (self.planes, n1, 8), torch.int16)
            act2 = self._buf("act2", (2 if hilo else self.planes, n2, 8), torch.int16)
            am0 = am1 = am2 = None
            if keep:
                am0 = torch.empty(n1 * 8, dtype=torch.uint8, device=self.device)
                am1 = torch.empty(n2 * 8, dtype=torch.uint8, device=self.device)
                am2 = torch.empty(nb * self.num_feat, dtype=torch.uint8, device=self.device)
            w = self._weights
            prof = self.profile
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if prof is not None else None
            if ev: ev[0].record()
            self.fwd[0].run(slots0, n_slots0, w[1], act1.data_ptr(), n1, am0, nb, clip_index=cidx, group=group)
            if ev: ev[1].record()
            self.fwd[1].run(act1, n1, w[3], act2.data_ptr(), n2, am1, nb, group=group, emit_lo=(2 if self.last_c8 else 1) if hilo else 0)
            if ev: ev[2].record()
            if hilo:
                self.fwd2x.run(act2, n2, w[5], feats[c0:].data_ptr(), 0, None, nb, out_scale=self.c8_scales if self.last_c8 else None)
            else:
                self.fwd[2].run(act2, n2, w[5], feats[c0:].data_ptr(), 0, am2, nb, group=group)
            if ev:
                ev[3].record()
                prof += [("fwd0", nb, ev[0], ev[1]), ("fwd1", nb, ev[1], ev[2]), ("fwd2", nb, ev[2], ev[3])]
            if keep:
                saved.append((c0, nb, am0, am1, am2))
        if keep:
            return feats, saved
        return feats

    def forward_sets(self, x: torch.Tensor, index: torch.Tensor, rows: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Features of x[index] with the clips in G = ``set_weights(dither=G)`` consecutive blocks of len(index) / G, block s
        multiplying by dithered operand set s (single-pass engines): ONE launch per layer, the set picked per box from the clip
        number (VdConvParams.w_set_clips; the first layer's register-resident-B kernel reloads its fragments when a
        workgroup's walk crosses into the next block)."""
        G = getattr(self, "_dither", 0)
        B = int(index.numel())
        assert G >= 2 and B % G == 0 and self.planes == 1, "set_weights(dither=G) on a single-pass engine first"
        g = self.geo
        per = B // G
        index = index.to(self.device, torch.int64).contiguous()
        feats = torch.empty((B, self.num_feat), dtype=torch.float32, device=self.device)
        rowp = P.pix_row_pitch(g.width)
        per1 = int(np.prod(self.fwd[0].plan.out_shape[:-1]))
        per2 = int(np.prod(self.fwd[1].plan.out_shape[:-1]))
        n1, n2 = B * per1, B * per2
        hilo = self.fwd2x is not None
        act1 = self._buf("act1", (1, n1, 8), torch.int16)
        act2 = self._buf("act2", (2 if hilo else 1, n2, 8), torch.int16)
        w = self._weights
        if rows is None:
            n_slots0 = B * g.frames * 3 * g.height * (rowp // 8)
            slots0 = self._buf("slots0", (1, n_slots0, 8), torch.int16)
            hip.check(hip.lib().vd_pix2rows(hip.ptr(x.detach().to(torch.float32).contiguous()), hip.ptr(index), ctypes.c_int64(B),
                                            g.frames, g.height, g.width, hip.ptr(slots0[0]), hip.ptr(None), self.prec,
                                            hip.stream_ptr(self.device)), "vd_pix2rows")
        if self.fwd[0].breg_ok:       # register-resident-B kernel: reloads its fragments when a workgroup's walk crosses into the next set
            if rows is not None:
                self.fwd[0].run(rows, int(rows.shape[1]), w[1], act1.data_ptr(), n1, None, B, clip_index=index, set_clips=per)
            else:
                self.fwd[0].run(slots0, n_slots0, w[1], act1.data_ptr(), n1, None, B, set_clips=per)
        else:
            for s in range(G):
                if rows is not None:
                    self.fwd[0].run(rows, int(rows.shape[1]), w[1], act1.data_ptr() + s * per * per1 * 16, n1, None, per,
                                    clip_index=index[s * per:], group=s)
                else:
                    per0 = g.frames * 3 * g.height * (rowp // 8)
                    self.fwd[0].run(slots0[:, s * per * per0:], n_slots0, w[1], act1.data_ptr() + s * per * per1 * 16, n1, None, per, group=s)
        if getattr(self, "after_first_level", None) is not None:      # (distill.DMTrainer: an event behind the first level's launch)
            self.after_first_level()
        for li, (src, n_src, per_src, dst_ptr, n_dst, per_dst_bytes) in enumerate((
                (act1, n1, per1, act2.data_ptr(), n2, per2 * 16), (act2, n2, per2, feats.data_ptr(), 0, self.num_feat * 4)), start=1):
            if li == 2 and hilo:                        # hi+lo weights: one set
                if os.environ.get("VD_L2_X2_SIM") == "act":
                    # measurement knob: the NUMERICS of a two-MFMA last level a_hi x (W_hi + W_lo) -- the low plane
                    # of the activations dropped -- at the cost of the three-MFMA program (what the parity of such a mode would be)
                    src[1].zero_()
                self.fwd2x.run(src, n_src, w[5], dst_ptr, 0, None, B, out_scale=self.c8_scales if self.last_c8 else None)
            elif per % self.fwd[li].plan.ncl == 0:       # a box never spans two sets: one launch
                self.fwd[li].run(src, n_src, w[2 * li + 1], dst_ptr, n_dst, None, B, set_clips=per,
                                 emit_lo=(2 if self.last_c8 else 1) if (li == 1 and hilo) else 0)
            else:
                for s in range(G):
                    self.fwd[li].run(src[:, s * per * per_src:], n_src, w[2 * li + 1], dst_ptr + s * per * per_dst_bytes, n_dst, None,
                                     per, group=s, emit_lo=(2 if self.last_c8 else 1) if (li == 1 and hilo) else 0)
        return feats

    def backward(self, saved, g_feat: torch.Tensor) -> torch.Tensor:
        """d loss / d x for the clips of a ``forward(..., keep=True)`` call (same weights)."""
        self._pack_bwd()
        g = self.geo
        g_feat = g_feat.detach().to(torch.float32).contiguous()
        B = g_feat.shape[0]
        dx = torch.empty((B, g.frames, g.channel, g.height, g.width), dtype=torch.float32, device=self.device)
        L = hip.lib()
        st = hip.stream_ptr(self.device)
        for c0, nb, am0, am1, am2 in saved:
            grad = g_feat[c0:c0 + nb]
            layout = 0
            for li, am in ((2, am2), (1, am1), (0, am0)):
                cin, cout, t, h, w, T, OH, OW, To, Ho, Wo, pt = self.dims[li]
                nslots = nb * (cout // 8) * T * OH * OW
                dy = self._buf("dy%d" % li, (self.planes_bwd, nslots, 8), torch.int16)
                lo = dy[1] if self.planes_bwd == 2 else None
                sc = inv = None
                if self.prec_bwd in (hip.PREC["f16"], hip.PREC["f16x3"]):
                    # fp16 operands (also the hi/lo split ones): bring this layer's gradient into fp16's
                    # exponent range with an exact power-of-two scale (DM gradients shrink by orders of
                    # magnitude per layer and would otherwise fall into fp16's subnormals)
                    scb = self._buf("gscale%d" % li, (4,), torch.float32)
                    hip.check(L.vd_absmax_scale(hip.ptr(grad), ctypes.c_int64(grad.numel()), ctypes.c_float(GRAD_TARGET()),
                                                hip.ptr(scb), st), "vd_absmax_scale")
                    sc, inv = scb, scb[1:]
                hip.check(L.vd_unpool_relu_bwd(hip.ptr(grad), hip.ptr(am), ctypes.c_int64(nb), cout, To, Ho, Wo, pt,
                                               T, OH, OW, layout, hip.ptr(dy[0]), hip.ptr(lo), self.prec_bwd, hip.ptr(sc), st),
                          "vd_unpool_relu_bwd")
                if lo is not None and os.environ.get("VD_BWD_X2_SIM") == "g":
                    lo.zero_()      # measurement knob: the numerics of g_hi x (W_hi + W_lo)
                if li == 0:
                    out = dx[c0:c0 + nb]
                else:
                    out = self._buf("dx%d" % li, (nb, t, h, w, cin), torch.float32)
                run_together(self.bwd[li], dy, nslots, None, out.data_ptr(), 0, None, nb, out_scale=inv)
                grad = out
                layout = 1
        return dx


class WgradOp:
    """dW of one Conv3d(k(3,7,7), s(1,2,2), p(1,3,3)) layer for a batch of `nclips` clips, as a tile
    program of the MFMA kernel (plan.plan_wgrad): x is re-laid out clip-minor, dy is packed into
    per-box B fragments, boxes of positions accumulate into dW with fp32 atomics."""

    def __init__(self, cin: int, cout: int, t: int, h: int, w: int, nclips: int, prec: str, device, ordered: bool = False):
        """``ordered``: one accumulation copy per box, folded in index order (plan.plan_wgrad; bitwise reproducible)."""
        self.cin, self.cout, self.t, self.h, self.w, self.nclips = cin, cout, t, h, w, nclips
        self.ordered = bool(ordered)
        self.prec = hip.PREC[prec]
        self.planes = 2 if hip.is_x3(self.prec) else 1
        self.device = torch.device(device)
        blk = os.environ.get("VD_WG_BLOCK")
        self.plan = P.plan_wgrad("wgrad%dx%d" % (cin, cout), cin, cout, t, h, w, nclips, block=tuple(int(v) for v in blk.split(",")) if blk else None,
                                 planes=self.planes, ordered=self.ordered)
        self.dp = _DevPlan(self.plan, self.device, self.prec)
        self.T, self.OH, self.OW = self.plan.meta["grid"]
        self.CCb = self.plan.CC
        self.npos_in = t * h * w
        self.xT = torch.empty((self.planes, cin * self.CCb * self.npos_in, 8), dtype=torch.int16, device=self.device)
        self.bp_elems = self.plan.nbox * self.plan.w_box_stride
        self.bp = torch.empty((self.planes, self.bp_elems), dtype=torch.int16, device=self.device)
        self.replicas = int(self.plan.meta["replicas"])
        self.rep = torch.empty((self.replicas, cin * 147, cout), dtype=torch.float32, device=self.device)

    def _stage_x(self, x_src: torch.Tensor, x_is_pixels: bool, x_plane_slots: int) -> None:
        L, st = hip.lib(), hip.stream_ptr(self.device)
        nb = self.nclips
        if x_is_pixels:
            lo = self.xT[1] if self.planes == 2 else None
            hip.check(L.vd_clip_minor_pix(hip.ptr(x_src), ctypes.c_int64(nb), self.t, self.h, self.w, hip.ptr(self.xT[0]),
                                          hip.ptr(lo), self.prec, st), "vd_clip_minor_pix")
        else:
            hip.check(L.vd_clip_minor_cl(hip.ptr(x_src), ctypes.c_int64(x_plane_slots), self.planes, ctypes.c_int64(nb), self.cin,
                                         ctypes.c_int64(self.npos_in), hip.ptr(self.xT), ctypes.c_int64(self.xT.shape[1]), st),
                      "vd_clip_minor_cl")

    def _accumulate(self, dw_out: torch.Tensor, out_scale: Optional[torch.Tensor]) -> None:
        """The tile program over the staged x and the packed dy: boxes accumulate into `replicas` cout-minor copies
        (coalesced atomics, no same-address pile-up), which are then folded into dw_out."""
        L, st = hip.lib(), hip.stream_ptr(self.device)
        self.rep.zero_()
        self.dp.params.replica_stride = dw_out.numel()
        self.dp.run(self.xT, self.xT.shape[1], None, self.rep.data_ptr(), 0, None, self.cin, out_scale=out_scale, wpk=self.bp,
                    w_plane_elems=self.bp_elems)
        hip.check(L.vd_replica_sum(hip.ptr(self.rep), self.replicas, self.cin * 147, self.cout, hip.ptr(dw_out), st), "vd_replica_sum")

    def run(self, x_src: torch.Tensor, x_is_pixels: bool, x_plane_slots: int, dy: torch.Tensor, dy_plane_slots: int,
            dw_out: torch.Tensor, out_scale: Optional[torch.Tensor] = None) -> None:
        """x_src: fp32 clips (B,T,3,H,W) if x_is_pixels else channels-last slots [planes][clip][C/8][npos][8];
        dy: dense slots [planes][clip][N/8][T][OH][OW][8]; dw_out (cout,cin,3,7,7) fp32 is ACCUMULATED into."""
        L, st = hip.lib(), hip.stream_ptr(self.device)
        self._stage_x(x_src, x_is_pixels, x_plane_slots)
        nt, noh, now = self.plan.meta["box"]
        hip.check(L.vd_pack_dy(hip.ptr(dy), ctypes.c_int64(dy_plane_slots), self.planes, ctypes.c_int64(self.nclips), self.cout, self.T,
                               self.OH, self.OW, nt, noh, now, hip.ptr(self.bp), ctypes.c_int64(self.bp_elems), st), "vd_pack_dy")
        self._accumulate(dw_out, out_scale)

    def run_pooled(self, x_src: torch.Tensor, x_is_pixels: bool, x_plane_slots: int, g_pooled: torch.Tensor, argmax: torch.Tensor,
                   g_layout: int, pooled: Tuple[int, int, int, int], scale: Optional[torch.Tensor], dw_out: torch.Tensor,
                   out_scale: Optional[torch.Tensor] = None) -> None:
        """As ``run`` for a layer whose dense dy has no other reader: dy = backward of ReLU + max-pool of the POOLED gradient
        ``g_pooled`` (layout / arg-max bytes as vd_unpool_relu_bwd takes them; ``pooled`` = (To, Ho, Wo, pool_t)) goes
        straight into the packed B operand (vd_unpool_relu_bwd_packed), bitwise what unpool + pack produce."""
        L, st = hip.lib(), hip.stream_ptr(self.device)
        self._stage_x(x_src, x_is_pixels, x_plane_slots)
        nt, noh, now = self.plan.meta["box"]
        To, Ho, Wo, pt = pooled
        lo = self.bp[1] if self.planes == 2 else None
        hip.check(L.vd_unpool_relu_bwd_packed(hip.ptr(g_pooled), hip.ptr(argmax), ctypes.c_int64(self.nclips), self.cout, To, Ho, Wo, pt,
                                              self.T, self.OH, self.OW, g_layout, nt, noh, now, hip.ptr(self.bp[0]), hip.ptr(lo),
                                              self.prec, hip.ptr(scale), st), "vd_unpool_relu_bwd_packed")
        self._accumulate(dw_out, out_scale)
