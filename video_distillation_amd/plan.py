"""Host-side planner for the table-driven MFMA convolution kernel (csrc/conv_mfma.hip).

The device kernel knows nothing about convolution geometry.  It executes a *tile program*:

  * a workgroup owns a *box* of output rows (<= MW*MTW*32) of ``ncl`` consecutive clips;
  * it stages the input *patch* of that box (16-byte slots = 8 x 16-bit values, zero filled
    outside the source grid) into LDS once per 8-wide input-channel chunk;
  * output row ``r`` of MFMA tile ``i`` reads its A fragment for tap ``p`` at LDS byte address
    ``a_off[i*32+r] + tap_off[p]``; two taps (lane halves) x 8 values form one K=16 step;
  * B fragments come pre-packed in fragment order (``widx`` is the gather table used to pack);
  * the epilogue is either max-pool over 8-row groups (+bias, ReLU, arg-max) or plain rows.

Everything index-heavy lives here in numpy, where tests/test_plan_emulation.py replays the
tile program on the CPU against the oracle before any GPU is involved.

Geometry follows the reference's ConvNet3D (networks.py:792-814: Conv3d k(3,7,7) s(1,2,2)
p(1,3,3) -> ReLU -> MaxPool3d) but nothing below is copied from it.
"""
from __future__ import annotations

import itertools
import dataclasses
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import os

import numpy as np

SLOT_BYTES = 16
EPI_POOL_CL = 0      # pooled, channels-last 16-bit slots out (next layer's source)
EPI_POOL_FEAT = 1    # pooled, fp32 features in (C,T,H,W) order (embed output)
EPI_ROWS = 2         # plain fp32 rows (input-gradient passes)

KT, KH, KW = 3, 7, 7
# ds_read_b128 services a wave in four 16-lane groups (MI355X_MICROARCH.md, LDS table)
_B128_GROUPS = (
    (0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27),
    (4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31),
)


@dataclass
class BoxType:
    pf: int
    ph: int
    pw: int
    pitch_h: int
    pitch_f: int
    pitch_c: int
    mt: int                      # MFMA row tiles in the workgroup (MW*MTW, padded)
    a_off: np.ndarray            # int32 [mt*32] LDS byte offsets of row origins
    out: np.ndarray              # int32 [mt*4] (pooled) or [mt*32] (rows); -1 = discard
    tap_off: np.ndarray          # int32 [2*S] LDS byte offsets per tap
    conflict_cycles: float = 0.0

    @property
    def patch_slots(self) -> int:
        return self.pitch_c


@dataclass
class ConvPlan:
    name: str
    # source slot grid per clip
    CC: int
    F: int
    H: int
    W: int
    # decomposition
    NT: int
    MW: int
    MTW: int
    S: int
    ncl: int
    boxes: np.ndarray            # int32 [nbox, 6]: type, f0, h0, w0, out_rel, clip_rel(always 0)
    types: List[BoxType]
    widx: np.ndarray             # int32 [CC, S, NT, 64, 8] gather into flat fp32 weights, -1 = 0
    # epilogue
    epi: int
    pool_t: int
    relu: bool
    n_out: int                   # valid output channels
    n_stride: int                # element stride between output channels (FEAT/ROWS)
    out_clip_stride: int         # slots (CL, per channel chunk grid) or elements per clip
    out_chunk_stride: int        # CL: slots per output channel chunk (Fo*Ho*Wo)
    out_shape: Tuple[int, ...]   # logical output grid (for allocation / tests)
    out_t_stride: int = 0        # pool_t == 1: distance between the two outputs of a row group
    # source addressing in 4-byte units (an LDS slot is filled from 16 bytes at any dword offset)
    w_step4: int = 4             # dwords between neighbouring w positions (4 = packed slots)
    row_pitch4: int = 0          # dwords between neighbouring h rows
    clip_stride4: int = 0        # dwords per clip (all channel chunks)
    chunk_stride4: int = 0       # dwords per 8-channel chunk
    col_off: Optional[np.ndarray] = None   # EPI_ROWS: element offset of output column n (else n*n_stride)
    atomic: bool = False         # EPI_ROWS: accumulate with fp32 atomics (several boxes add into the same rows)
    w_box_stride: int = 0        # 16-bit elements between the packed B operands of consecutive boxes (0 = shared)
    NTW: int = 1                 # N tiles per wave: wave grid = (NT / NTW) columns x MW rows
    pair_flip: int = 0           # pool_t == 1: bit q = the q-th row group of a tile has its second output out_t_stride BEFORE the first
    rows_total: int = 0
    rows_useful: int = 0
    meta: Dict = field(default_factory=dict)

    @property
    def nbox(self) -> int:
        return int(self.boxes.shape[0])

    @property
    def threads(self) -> int:
        return 64 * (self.NT // self.NTW) * self.MW

    @property
    def lds_slots(self) -> int:
        return max(t.pitch_c * self.ncl for t in self.types)

    def grid(self, nclips: int) -> int:
        return ((nclips + self.ncl - 1) // self.ncl) * self.nbox

    def gather_table(self) -> np.ndarray:
        """int32 [nbox, stride]: for every LDS slot of a box's patch the source slot it is
        filled from, relative to the (clip, channel-chunk) base, with the box-local clip index
        in bits 24..30; -1 = zero fill (conv padding, pitch padding, outside the grid)."""
        stride = -(-(self.lds_slots + 1) // 64) * 64       # whole LDS-DMA wave-instructions (64 slots)
        out = -np.ones((self.nbox, stride), dtype=np.int64)
        assert self.F * self.H * self.row_pitch4 < (1 << 24) and self.ncl < 128
        for bi, box in enumerate(self.boxes):
            t = self.types[int(box[0])]
            f0, h0, w0 = int(box[1]), int(box[2]), int(box[3])
            idx = np.arange(self.ncl * t.pitch_c)
            ci, r1 = np.divmod(idx, t.pitch_c)
            f, r2 = np.divmod(r1, t.pitch_f)
            h, w = np.divmod(r2, t.pitch_h)
            sf, sh, sw = f0 + f, h0 + h, w0 + w
            ok = (f < t.pf) & (h < t.ph) & (w < t.pw) & (sf >= 0) & (sf < self.F) & (sh >= 0) & (sh < self.H) \
                & (sw >= 0) & (sw < self.W)
            rel = (sf * self.H + sh) * self.row_pitch4 + sw * self.w_step4     # dword offset in the chunk
            out[bi, :idx.size] = np.where(ok, rel | (ci << 24), -1)
        return out.astype(np.int32)

    def row_source(self) -> Tuple[int, int]:
        """(planes, rows per plane) of the source clip when this is a first-level program over 16-bit pixel rows whose patch
        can be built from ALIGNED 16-byte row loads (csrc/conv_mfma.hip: conv0_breg3 / conv0_breg4): 2 x 2 waves of 4 M tiles,
        32 K steps, one box type of 8 output columns starting at a multiple of 4 dwords, at most 768 patch-row halves,
        box origins that fit 16 bits.  (0, 0) otherwise.  Exported in the program header (words 35, 36)."""
        t0 = self.types[0]
        ok = (self.w_step4 == 1 and self.CC == 1 and self.ncl == 1 and self.NTW in (0, 1) and len(self.types) == 1
              and (self.NT, self.MW, self.MTW, self.S) == (2, 2, 4, 32) and t0.pw == 8 and self.row_pitch4 % 4 == 0
              and all(int(b[3]) % 4 == 0 and int(b[3]) >= 0 for b in self.boxes) and 2 * t0.pf * t0.ph <= 768
              and all(abs(int(b[1])) < 32768 and abs(int(b[2])) < 32768 for b in self.boxes))
        return (int(self.F), int(self.H)) if ok else (0, 0)

    def device_boxes(self) -> np.ndarray:
        """int32 [nbox, 8] rows the kernel reads with one scalar load: offsets of the box type's
        a_off / out / tap tables inside flat_tables()[1], the output origin, the type id."""
        descs, _ = self.flat_tables()
        out = np.zeros((self.nbox, 8), dtype=np.int32)
        for bi, box in enumerate(self.boxes):
            ty = int(box[0])
            out[bi, :5] = (descs[ty][7], descs[ty][8], descs[ty][9], int(box[4]), ty)
            out[bi, 5] = int(box[5])              # replica of the accumulation target (weight-gradient programs), else 0
            # patch origin in source slot coordinates, (f0 << 16) | (h0 & 0xffff) and w0: read by the first-level kernel that
            # builds its patch from aligned row loads (conv0_breg3_kernel)
            out[bi, 6] = np.array((int(box[1]) << 16) | (int(box[2]) & 0xFFFF), dtype=np.int64).astype(np.int32)
            out[bi, 7] = int(box[3])
        return out

    def flat_tables(self):
        """Serialise box types into (type_desc int32 [ntypes,16], tables int32[...])."""
        descs, chunks, pos = [], [], 0
        for t in self.types:
            a_ofs = pos; chunks.append(t.a_off.astype(np.int32)); pos += t.a_off.size
            o_ofs = pos; chunks.append(t.out.astype(np.int32)); pos += t.out.size
            t_ofs = pos; chunks.append(t.tap_off.astype(np.int32)); pos += t.tap_off.size
            descs.append([t.pf, t.ph, t.pw, t.pitch_h, t.pitch_f, t.pitch_c, t.mt,
                          a_ofs, o_ofs, t_ofs, 0, 0, 0, 0, 0, 0])
        return np.asarray(descs, dtype=np.int32), np.concatenate(chunks).astype(np.int32)


# ----------------------------------------------------------------------------------------
# bank-conflict model for the A-fragment reads
PROGRAM_MAGIC = b"VDPROG01"
PROGRAM_HEADER_WORDS = 40


BOX_WALK_GENERATIONS = 16    # first-level kernels walk boxes: grid = this many generations of resident workgroups.  4 -> 8 in round 3 (a
#                              workgroup of the eight-wave first-level kernel holds its CU's LDS for half as long, so the synthetic-clip
#                              stream's kernels find CUs during the first level instead of piling onto level 1), 8 -> 16 in round 5 with
#                              the faster kernel: stand-alone 7.95 / 7.87 / 7.91 / 8.01 ms at 8 / 4 / 16 / 32 (noise), DM step 31.78 -> 31.67 ms
#                              (4: 31.95), two same-box pairs each


def export_program(plan: "ConvPlan", persist: int = BOX_WALK_GENERATIONS) -> bytes:
    """Serialise a tile program for ``vd_program_load`` (include/vd_hip.h): 40 int64 header words
    followed by the int32 arrays type_desc | tables | boxes | gather | widx | col_off."""
    desc, tables = plan.flat_tables()
    boxes = plan.device_boxes().reshape(-1)
    gt = plan.gather_table()
    widx = plan.widx.reshape(-1).astype(np.int32)
    col = np.zeros(0, dtype=np.int32) if plan.col_off is None else plan.col_off.astype(np.int32).reshape(-1)
    mt_max = max(t.mt for t in plan.types)
    h = np.zeros(PROGRAM_HEADER_WORDS, dtype=np.int64)
    h[0] = np.frombuffer(PROGRAM_MAGIC, dtype=np.int64)[0]
    h[1:28] = [plan.CC, plan.S, plan.NT, plan.MW, plan.MTW, plan.NTW, plan.epi, plan.pool_t, int(plan.relu), plan.n_out,
               plan.n_stride, plan.out_clip_stride, plan.out_chunk_stride, plan.out_t_stride, int(gt.shape[1]) * 16,
               len(plan.types), int(desc[0][7]), int(desc[0][8]), int(desc[0][9]), int(plan.atomic), plan.w_box_stride,
               plan.ncl, plan.nbox, int(gt.shape[1]), plan.clip_stride4, plan.chunk_stride4,
               mt_max if (plan.NTW == 2 and mt_max < plan.MW * plan.MTW) else 0]
    h[28:35] = [desc.size, tables.size, boxes.size, gt.size, widx.size, col.size, persist]
    h[35:37] = plan.row_source()
    h[37] = plan.pair_flip
    arrays = [desc.reshape(-1), tables, boxes, gt.reshape(-1), widx, col]
    return h.tobytes() + b"".join(np.ascontiguousarray(a, dtype=np.int32).tobytes() for a in arrays)


# ----------------------------------------------------------------------------------------
def _conflict_cycles(slots_per_row: np.ndarray) -> float:
    """Average LDS cycles per ds_read_b128 wave-instruction (4 = conflict-free) for one MFMA
    tile whose 32 rows start at the given 16-byte slot indices (both lane halves read the
    same rows at a tap-dependent but half-uniform offset, so taps do not matter)."""
    total = 0
    for grp in _B128_GROUPS:
        banks = slots_per_row[list(grp)] % 16
        total += np.bincount(banks, minlength=16).max()
    return 2.0 * total  # two lane halves


def _row_of(q: int, j: int) -> int:
    """MFMA 32x32 C/D layout: group q (0..3) and in-group index j (0..7) -> row of the tile.
    Rows of one group sit in ONE lane (registers 8*(q>>1)..+7 of lane half q&1)."""
    return (j & 3) + 4 * (q & 1) + 8 * (j >> 2) + 16 * (q >> 1)


def _choose_box(dims, group, rows_max, slot_fn, lds_budget):
    """Pick the box shape (na,nb,nc) minimising boxes (then patch size) under the row and
    LDS limits.  ``group`` is the (ga,gb,gc) granularity a box must respect."""
    RA, RB, RC = dims
    best = None
    cand_a = [a for a in range(group[0], RA + group[0], group[0])]
    cand_b = [b for b in range(group[1], RB + group[1], group[1])]
    cand_c = [c for c in range(group[2], RC + group[2], group[2])]
    for na, nb, nc in itertools.product(cand_a, cand_b, cand_c):
        if na > RA + group[0] - 1 or nb > RB + group[1] - 1 or nc > RC + group[2] - 1:
            continue
        if na * nb * nc > rows_max:
            continue
        slots = slot_fn(na, nb, nc)
        if slots > lds_budget:
            continue
        nbox = -(-RA // na) * -(-RB // nb) * -(-RC // nc)
        key = (nbox, nc < min(8, RC), slots)   # narrow patch rows coalesce badly
        if best is None or key < best[0]:
            best = (key, (na, nb, nc))
    if best is None:
        raise ValueError("no box shape fits %s rows<=%d lds<=%d" % (dims, rows_max, lds_budget))
    return best[1]


_PERMS4 = list(itertools.permutations(range(4)))
FRAME_TILE_FLIP = 0b0110       # frame-tile programs: row groups 1 and 2 of a tile list their second window first (VdConvParams.pair_flip)
FRAME_TILE_OUT_STEP = 2        # ... whose two windows are this many pooled columns apart (out_t_stride)


def _build_type(box, group, row_lin, tap_list, patch_ext, mt_pad, pooled, out_fn, valid_fn, ncl, pad_search=True,
                slot_cap=1 << 30, frame_tiles=False):
    """Pooled boxes: which four row groups (pool windows) share an MFMA tile is free -- the kernel is table-driven --
    so the enumeration order of the groups (which axis runs fastest) is searched too, next to the LDS pitches."""
    if not pooled:
        return _build_type_order(box, group, row_lin, tap_list, patch_ext, mt_pad, pooled, out_fn, valid_fn, ncl, pad_search,
                                 slot_cap, (0, 1, 2))
    if frame_tiles:          # one enumeration: tile = (set of four row groups of one frame), tiles of a wave row = consecutive frames
        return _build_type_order(box, group, row_lin, tap_list, patch_ext, mt_pad, pooled, out_fn, valid_fn, ncl, pad_search,
                                 slot_cap, None)
    best = None
    for order in ((0, 1, 2), (0, 2, 1), (1, 0, 2), (2, 0, 1), (1, 2, 0), (2, 1, 0)):
        bt = _build_type_order(box, group, row_lin, tap_list, patch_ext, mt_pad, pooled, out_fn, valid_fn, ncl, pad_search,
                               slot_cap, order)
        key = (round(bt.conflict_cycles, 3), bt.pitch_c)
        if best is None or key < best[0]:
            best = (key, bt)
        if bt.conflict_cycles <= 4.0 + 1e-9 and bt.pitch_c * ncl <= slot_cap:
            break
    return best[1]


def _build_type_order(box, group, row_lin, tap_list, patch_ext, mt_pad, pooled, out_fn, valid_fn, ncl, pad_search,
                      slot_cap, order):
    """Build one BoxType.  ``row_lin(a,b,c)`` -> (f,h,w) patch-relative slot coords of the row
    origin; ``tap_list`` -> (df,dh,dw); ``out_fn`` gives the output index of a group / row.
    Searches LDS pitches (and, for pooled tiles, which of a tile's four row groups sits in
    which lane-half/register-half) for the fewest ds_read_b128 bank conflicts."""
    na, nb, nc = box
    pf, ph, pw = patch_ext
    ga, gb, gc = group
    if pooled:
        axes = (range(0, na, ga), range(0, nb, gb), range(0, nc, gc))
        groups = []
        if order is None:
            # FRAME TILES (first level, pool (1,2,2), boxes of 8 x 8 conv rows per frame): a row group is two pool windows of ONE
            # frame, four conv columns apart -- (b, c) and (b, c + 4), c in {0, 2}; its second output lies out_t_stride = 2 pooled
            # columns from the first; an MFMA tile is the four groups of one column class c in one frame, and the tiles of a
            # wave row are the SAME four groups in consecutive frames -- so the A fragment of (frame t + kt, tap pair j) is the
            # same registers for tile t at kt and tile t + 1 at kt - 1 (csrc/conv_mfma.hip: conv0_breg_kernel<PREC, true>).
            # With pitch_h = 9 a window's four rows cover four consecutive LDS banks starting at 2 b + c (mod 16); a 16-lane
            # service group of ds_read_b128 holds the FIRST windows of a tile's groups 0 and 3 and the SECOND windows of groups
            # 1 and 2, so groups 1 and 2 list their windows right-to-left (``FRAME_TILE_FLIP``, VdConvParams.pair_flip): the
            # group then reads banks {2 b, 2 b + 4} x 4 -- no conflicts, which no left-to-right pairing achieves within the LDS
            # budget of the patch.
            assert (ga, gb, gc) == (1, 2, 8) and nc == 8 and nb == 8
            for ci in range(ncl):
                for c in (0, 2):
                    for a in axes[0]:
                        groups += [(ci, a, b, c) for b in axes[1]]
        else:
          for ci in range(ncl):           # `order` lists the axes from slowest to fastest
            for x in axes[order[0]]:
                for y in axes[order[1]]:
                    for z in axes[order[2]]:
                        abc = [0, 0, 0]
                        abc[order[0]], abc[order[1]], abc[order[2]] = x, y, z
                        groups.append((ci, abc[0], abc[1], abc[2]))
        ngr = len(groups)
        # coordinates of the 8 rows of every group: [ngr, 8, 4] = (ci, f, h, w)
        gcoord = np.zeros((ngr, 8, 4), dtype=np.int64)
        for gi, (ci, a, b, c) in enumerate(groups):
            for j in range(8):
                dt, dh, dw = (j >> 2) & 1, (j >> 1) & 1, j & 1
                if order is None:
                    dt, dw = 0, 4 * dt + dw
                f, h, w = row_lin(a + dt, b + dh, c + dw)
                gcoord[gi, j] = (ci, f, h, w)
        gout = np.array([out_fn(ci, a, b, c) if valid_fn(a, b, c) else -1 for ci, a, b, c in groups], dtype=np.int64)
        ntile_used = -(-ngr // 4)
        rows_of = np.array([[_row_of(q, j) for j in range(8)] for q in range(4)])
    else:
        rows = [(ci, a, b, c) for ci in range(ncl) for a in range(na) for b in range(nb) for c in range(nc)]
        rcoord = np.array([(ci,) + tuple(row_lin(a, b, c)) for ci, a, b, c in rows], dtype=np.int64)
        rout = np.array([out_fn(ci, a, b, c) if valid_fn(a, b, c) else -1 for ci, a, b, c in rows], dtype=np.int64)
        ntile_used = -(-len(rows) // 32)
    best = None
    for dph in (range(0, 16) if pad_search else [0]):
        pitch_h = pw + dph
        for dpf in (range(0, 16) if pad_search else [0]):
            pitch_f = ph * pitch_h + dpf
            pitch_c = pf * pitch_f
            if pitch_c * ncl > slot_cap and best is not None:
                continue
            a_off = np.zeros(mt_pad * 32, dtype=np.int64)
            if pooled:
                gslot = gcoord[:, :, 0] * pitch_c + gcoord[:, :, 1] * pitch_f + gcoord[:, :, 2] * pitch_h + gcoord[:, :, 3]
                out = -np.ones(mt_pad * 4, dtype=np.int64)
                cyc_sum = 0.0
                for tile in range(ntile_used):
                    gids = list(range(tile * 4, min(ngr, tile * 4 + 4)))
                    gids += [gids[0]] * (4 - len(gids))          # pad with a duplicate (discarded)
                    best_t = None
                    for perm in _PERMS4:
                        slots = np.zeros(32, dtype=np.int64)
                        for q in range(4):
                            gs = gslot[gids[perm[q]]]
                            if order is None and (FRAME_TILE_FLIP >> q) & 1:      # this group lists its second window first
                                gs = np.concatenate([gs[4:], gs[:4]])
                            slots[rows_of[q]] = gs
                        cyc = _conflict_cycles(slots)
                        if best_t is None or cyc < best_t[0]:
                            best_t = (cyc, perm, slots)
                        if cyc <= 4.0:
                            break
                    cyc, perm, slots = best_t
                    cyc_sum += cyc
                    a_off[tile * 32:(tile + 1) * 32] = slots
                    for q in range(4):
                        gi = tile * 4 + perm[q]
                        o = gout[gi] if gi < ngr else -1
                        if order is None and (FRAME_TILE_FLIP >> q) & 1 and o >= 0:
                            o += FRAME_TILE_OUT_STEP               # the first window of a flipped group is the right-hand one
                        out[tile * 4 + q] = o
                cyc = cyc_sum / ntile_used
            else:
                slots = rcoord[:, 0] * pitch_c + rcoord[:, 1] * pitch_f + rcoord[:, 2] * pitch_h + rcoord[:, 3]
                a_off[:slots.size] = slots
                out = -np.ones(mt_pad * 32, dtype=np.int64)
                out[:rout.size] = rout
                cyc = float(np.mean([_conflict_cycles(a_off[t * 32:(t + 1) * 32]) for t in range(ntile_used)]))
            key = (round(float(cyc), 3), pitch_c)
            if best is None or key < best[0]:
                tap_off = np.array([(df * pitch_f + dh * pitch_h + dw) * SLOT_BYTES for df, dh, dw in tap_list],
                                   dtype=np.int64)
                best = (key, BoxType(pf, ph, pw, pitch_h, pitch_f, pitch_c, mt_pad,
                                     (a_off * SLOT_BYTES).astype(np.int32), out.astype(np.int32),
                                     tap_off.astype(np.int32), float(cyc)))
            if cyc <= 4.0 + 1e-9:
                break
        if best is not None and best[0][0] <= 4.0 + 1e-9:
            break
    return best[1]


def _make_plan(name, src_grid, CC, row_dims, group, row_origin, row_stride, taps, widx_fn,
               n_out, NT, MW, mtw_options, epi, pool_t, relu, out_index, out_valid, n_stride,
               out_clip_stride, out_chunk_stride, out_shape, lds_budget, ncl_options=(1,), force_box=None, ntw=1, step_multiple=1,
               frame_tiles=False):
    """Generic planner.  Row (a,b,c) has its tap-(0,0,0) origin at source slot coords
    (row_stride[0]*a+row_origin[0], ...).  ``taps`` is a list of non-negative (df,dh,dw).  ``step_multiple``: pad the K steps
    (tap pairs) to a multiple of it with zero-weight taps (VD_PREC_F16C8 programs correct four steps at a time)."""
    F, H, W = src_grid
    ntaps = len(taps)
    S = -(-((ntaps + 1) // 2) // step_multiple) * step_multiple
    taps_p = list(taps) + [(0, 0, 0)] * (2 * S - ntaps)
    mdf = max(t[0] for t in taps); mdh = max(t[1] for t in taps); mdw = max(t[2] for t in taps)
    sa, sb, sc = row_stride
    pooled = epi in (EPI_POOL_CL, EPI_POOL_FEAT)

    def ext(na, nb, nc):
        return (sa * (na - 1) + mdf + 1, sb * (nb - 1) + mdh + 1, sc * (nc - 1) + mdw + 1)

    best = None
    for MTW in mtw_options:
        for ncl in ncl_options:
            rows_max = (MW * MTW * 32) // ncl
            if rows_max < group[0] * group[1] * group[2]:
                continue
            # programs with <= 4 accumulator tiles per wave run the box-walking instantiation, whose DMA register
            # budget is 14 groups of 64 slots per wave
            # (waves x 14 x 64 slots; the planned patch incl. pitch padding must stay below it)
            waves = (NT // ntw) * MW
            dma_cap = waves * (14 if MTW * ntw <= 4 else 17) * 64
            budget = min(lds_budget, dma_cap - 64)

            def slot_fn(na, nb, nc, ncl=ncl):
                e = ext(na, nb, nc)
                return int(ncl * e[0] * e[1] * e[2] * 1.06) + 16

            try:
                box = force_box or _choose_box(row_dims, group, rows_max, slot_fn, budget)
            except ValueError:
                continue
            nbox = -(-row_dims[0] // box[0]) * -(-row_dims[1] // box[1]) * -(-row_dims[2] // box[2])
            cost = nbox * MTW * MW / ncl
            if best is None or cost < best[0]:
                best = (cost, MTW, ncl, box)
    if best is None:
        raise ValueError("%s: no feasible decomposition" % name)
    _, MTW, ncl, box = best
    na, nb, nc = box
    mt_pad = MW * MTW
    if frame_tiles and not (na == MTW and (nb, nc) == (8, 8) and MW == 2 and ncl == 1):
        raise ValueError("%s: box %s does not split into frame tiles" % (name, (box,)))
    dma_cap_final = (NT // ntw) * MW * (14 if MTW * ntw <= 4 else 17) * 64

    def row_lin(a, b, c):
        return (sa * a, sb * b, sc * c)

    types: List[BoxType] = []
    type_key: Dict[Tuple, int] = {}
    boxes = []
    for a0 in range(0, row_dims[0], na):
        for b0 in range(0, row_dims[1], nb):
            for c0 in range(0, row_dims[2], nc):
                # validity pattern of this box (edge boxes may hang over the row grid)
                va = min(na, row_dims[0] - a0); vb = min(nb, row_dims[1] - b0); vc = min(nc, row_dims[2] - c0)
                key = (va, vb, vc)
                f0 = sa * a0 + row_origin[0]; h0 = sb * b0 + row_origin[1]; w0 = sc * c0 + row_origin[2]
                if key not in type_key:
                    def valid_fn(a, b, c, va=va, vb=vb, vc=vc):
                        if frame_tiles:       # (row dims are multiples of the box in c: a group's two windows, c and c + 4, always exist)
                            return a < va and b + 2 <= vb
                        if pooled:
                            return a + group[0] <= va and b + group[1] <= vb and c + group[2] <= vc
                        return a < va and b < vb and c < vc

                    def out_fn(ci, a, b, c):
                        return out_index(ci, a, b, c)  # relative to the box's out_rel

                    bt = _build_type(box, group, row_lin, taps_p, ext(na, nb, nc), mt_pad, pooled,
                                     out_fn, valid_fn, ncl, slot_cap=min(int(lds_budget * 1.12), dma_cap_final),
                                     frame_tiles=frame_tiles)
                    type_key[key] = len(types)
                    types.append(bt)
                boxes.append([type_key[key], f0, h0, w0, out_index(0, a0, b0, c0) - out_index(0, 0, 0, 0), 0])
    # out tables were built relative to (0,0,0) of the grid; shift so that they are relative to the box
    # origin: out_index is affine in (a,b,c), so out(ci,a0+a,..) = out(ci,a,..) + [out(0,a0,..)-out(0,0,0,0)].
    widx = widx_fn(CC, S, NT, taps_p, ntaps)
    rows_useful = row_dims[0] * row_dims[1] * row_dims[2]
    return ConvPlan(name=name, CC=CC, F=F, H=H, W=W, row_pitch4=W * 4, chunk_stride4=F * H * W * 4,
                    clip_stride4=CC * F * H * W * 4, NT=NT, MW=MW, MTW=MTW, S=S, ncl=ncl,
                    boxes=np.asarray(boxes, dtype=np.int32), types=types, widx=widx, epi=epi,
                    pool_t=pool_t, relu=relu, n_out=n_out, n_stride=n_stride,
                    out_clip_stride=out_clip_stride, out_chunk_stride=out_chunk_stride,
                    out_shape=tuple(out_shape), rows_total=len(boxes) * mt_pad * 32 // ncl,
                    rows_useful=rows_useful,
                    # algorithmic multiply-accumulates per clip: output rows x taps x input channels x output columns
                    meta={"box": box, "macs_per_unit": int(rows_useful) * ntaps * CC * 8 * int(n_out)})


# ----------------------------------------------------------------------------------------
# concrete layers
# ----------------------------------------------------------------------------------------
def conv_out_dim(n: int, k: int, s: int, p: int) -> int:
    return (n + 2 * p - k) // s + 1


def _lane_cols():
    lane = np.arange(64)
    return lane & 31, lane >> 5


def _memo(fn):
    """Per-process cache of a layer planner: engines with different precisions / batch hints ask for the same
    programs again and again, and the pitch / tile-assignment search is the expensive part."""
    cache = {}

    def wrapped(*args, **kwargs):
        key = (args, tuple(sorted(kwargs.items())), os.environ.get("VD_L0_BOX"), os.environ.get("VD_NTW2_MTW"), os.environ.get("VD_L0_FRAME_TILES"))
        if key not in cache:
            cache[key] = fn(*args, **kwargs)
        return cache[key]
    wrapped.__name__, wrapped.__doc__ = fn.__name__, fn.__doc__
    return wrapped


@_memo
def plan_forward_cl(name: str, cin: int, cout: int, t_in: int, h_in: int, w_in: int, pool_t: int,
                    feat_out: bool, lds_budget: int = 3700, mtw_options=(7, 8, 4, 2), ntw: int = 1, step_multiple: int = 1) -> ConvPlan:
    """Forward Conv3d(cin->cout) + ReLU + MaxPool(pool_t,2,2) over a channels-last chunked
    source [clip][cin/8][t][h][w] (slots of 8 channels).  ``ntw`` = N tiles per wave: with 2 the
    workgroup is 2 wave columns x 2 wave rows of 4 M tiles x 2 N tiles, and every A fragment read
    from LDS feeds two MFMAs."""
    assert cin % 8 == 0 and cout % 32 == 0
    if ntw == 2:
        assert cout == 128
        mtw_options = tuple(int(v) for v in os.environ.get("VD_NTW2_MTW", "4").split(","))
    CC = cin // 8
    T = conv_out_dim(t_in, KT, 1, 1); OH = conv_out_dim(h_in, KH, 2, 3); OW = conv_out_dim(w_in, KW, 2, 3)
    To, Ho, Wo = T // pool_t, OH // 2, OW // 2
    rows = (To * pool_t, Ho * 2, Wo * 2)           # conv rows that survive floor pooling
    taps = [(kt, kh, kw) for kt in range(KT) for kh in range(KH) for kw in range(KW)]
    NT = cout // 32
    MW = max(1, 4 // (NT // ntw))
    col, half = _lane_cols()

    def widx_fn(CC_, S, NT_, taps_p, ntaps):
        idx = -np.ones((CC_, S, NT_, 64, 8), dtype=np.int64)
        for s in range(S):
            for hh in range(2):
                p = 2 * s + hh
                if p >= ntaps:
                    continue
                kt, kh, kw = taps_p[p]
                lanes = np.where(half == hh)[0]
                for nt in range(NT_):
                    n = nt * 32 + col[lanes]                           # [32]
                    for cc in range(CC_):
                        c = cc * 8 + np.arange(8)                      # [8]
                        flat = (((n[:, None] * cin + c[None, :]) * KT + kt) * KH + kh) * KW + kw
                        idx[cc, s, nt, lanes, :] = flat
        return idx.astype(np.int32)

    if feat_out:
        epi = EPI_POOL_FEAT
        npos = To * Ho * Wo
        feat_stride = cout * npos

        def out_index(ci, a, b, c):
            return ci * feat_stride + ((a // pool_t) * Ho + b // 2) * Wo + c // 2
        n_stride, clip_stride, chunk_stride = npos, feat_stride, 0
        out_shape = (cout, To, Ho, Wo)
    else:
        epi = EPI_POOL_CL
        chunk_stride = To * Ho * Wo
        clip_stride = (cout // 8) * chunk_stride

        def out_index(ci, a, b, c):
            return ci * clip_stride + ((a // pool_t) * Ho + b // 2) * Wo + c // 2
        n_stride = 0
        out_shape = (cout // 8, To, Ho, Wo, 8)
    max_ncl = max(1, (8 * 32 * MW) // (rows[0] * rows[1] * rows[2]))
    ncl_options = sorted({1, max_ncl} | {n for n in (2, 4, 8) if n <= max_ncl})
    plan = _make_plan(name, (t_in, h_in, w_in), CC, rows, (2, 2, 2), (-1, -3, -3), (1, 2, 2), taps,
                      widx_fn, cout, NT, MW, mtw_options, epi, pool_t, True, out_index, None, n_stride,
                      clip_stride, chunk_stride, out_shape, lds_budget, ncl_options, ntw=ntw, step_multiple=step_multiple)
    plan.NTW = ntw
    return plan


EPI_POS_FEAT = 3     # position tiles (plan_forward_pos): the four M tiles of a wave are the four positions of ONE pool window, pooled
#                      across the tiles; fp32 features out


def with_skip_table(plan: ConvPlan) -> ConvPlan:
    """A VD_PREC_F16C8 program carries, behind the 2 S tap offsets of every box type, S / 4 words of TILE SKIP MASKS (bit i of word g:
    M tile i of a wave takes no part in K steps 4 g .. 4 g + 3).  Programs that skip nothing get zeros."""
    assert plan.S % 4 == 0
    for t in plan.types:
        if t.tap_off.size == 2 * plan.S:
            t.tap_off = np.concatenate([t.tap_off, np.zeros(plan.S // 4, dtype=np.int32)]).astype(np.int32)
    return plan


@_memo
def plan_forward_pos(name: str, cin: int, cout: int, t_in: int, h_in: int, w_in: int, pool_t: int) -> ConvPlan:
    """The LAST level's forward (features out) in POSITION TILES, for the fp8-corrected program (VD_PREC_F16C8).

    With a 7 x 7 input grid, stride 2 and a 7 x 7 kernel only 51 % of a position's spatial taps fall inside the grid, and a tile of
    32 consecutive positions cannot skip any of them: some row always needs the tap.  Here a tile's 32 rows are (clip, frame) pairs
    of ONE output position (32 / T clips per box), the four tiles of a wave are the four positions of one 2 x 2 pool window, a
    workgroup is one window x all output channels, and a box type per window lists ITS taps -- the union over the window's
    positions, grouped by which of the four tiles they are valid for (skip masks, ``with_skip_table``; groups padded to four K
    steps with zero-weight taps).  MFMAs per product: 152 instead of 304 per channel chunk at 7 x 7.  The patch holds only the
    window's real pixels (no conv padding in h / w; one zero frame at either end), the B operands are packed once per window
    (``w_box_stride``), and the pool runs ACROSS the four accumulator tiles (and over the frame pair inside a lane)."""
    if cin % 8 != 0 or cout != 128 or pool_t != 2:
        raise ValueError("%s: position tiles are planned for 128 output channels (four N tiles = four waves) and (2,2,2) pooling" % name)
    CC, NT = cin // 8, cout // 32
    T = conv_out_dim(t_in, KT, 1, 1); OH = conv_out_dim(h_in, KH, 2, 3); OW = conv_out_dim(w_in, KW, 2, 3)
    To, Ho, Wo = T // 2, OH // 2, OW // 2
    if T % 2 or 32 % T or T > 32 or Ho < 1 or Wo < 1:
        raise ValueError("%s: position tiles need an even frame count that divides 32" % name)
    ncl = 32 // T
    npos = To * Ho * Wo
    feat_stride = cout * npos
    col, half = _lane_cols()
    pf = T + 2
    win = []
    for a in range(Ho):
        for b in range(Wo):
            # tile order: 0 = the position with the most taps inside the grid in h AND w ("centre"), 1 = centre h / edge w,
            # 2 = edge h / centre w, 3 = edge / edge -- the skip masks are then 0, 0b1100, 0b1010, 0b1110 wherever the edge
            # position's valid taps are a subset of the centre's (7 x 7 grids: always), the patterns the kernel has bodies for
            nh = {oh: sum(0 <= 2 * oh + kh - 3 < h_in for kh in range(KH)) for oh in (2 * a, 2 * a + 1)}
            nw = {ow: sum(0 <= 2 * ow + kw - 3 < w_in for kw in range(KW)) for ow in (2 * b, 2 * b + 1)}
            hc, he = sorted(nh, key=lambda o: (-nh[o], o))
            wc, we = sorted(nw, key=lambda o: (-nw[o], o))
            pos = [(hc, wc), (hc, we), (he, wc), (he, we)]
            ok_h = [[0 <= 2 * oh + kh - 3 < h_in for kh in range(KH)] for oh, _ in pos]
            ok_w = [[0 <= 2 * ow + kw - 3 < w_in for kw in range(KW)] for _, ow in pos]
            hs = [2 * oh + kh - 3 for i, (oh, _) in enumerate(pos) for kh in range(KH) if ok_h[i][kh]]
            ws = [2 * ow + kw - 3 for i, (_, ow) in enumerate(pos) for kw in range(KW) if ok_w[i][kw]]
            h0, w0, ph, pw = min(hs), min(ws), max(hs) - min(hs) + 1, max(ws) - min(ws) + 1
            groups: Dict[int, list] = {}
            for kh in range(KH):
                for kw in range(KW):
                    skip = sum(1 << i for i in range(4) if not (ok_h[i][kh] and ok_w[i][kw]))
                    if skip != 15:
                        groups.setdefault(skip, []).extend((kt, kh, kw) for kt in range(KT))
            win.append(dict(a=a, b=b, pos=pos, h0=h0, w0=w0, ph=ph, pw=pw, groups=groups))
    # K steps: per window, mask groups in the order of falling tile count, each padded to a multiple of 8 taps (4 steps)
    for wd in win:
        taps, masks = [], []
        for skip in sorted(wd["groups"], key=lambda m: (bin(m).count("1"), m)):
            g = wd["groups"][skip]
            npad = -len(g) % 8
            taps += g + [None] * npad
            masks += [skip] * ((len(g) + npad) // 8)
        wd["taps"], wd["masks"] = taps, masks
    S = max(len(wd["taps"]) for wd in win) // 2
    ph_max, pw_max = max(wd["ph"] for wd in win), max(wd["pw"] for wd in win)
    # LDS pitches: a tile's rows are (clip, frame) -> slots ci * pitch_c + t * pitch_f: search the paddings for conflict-free b128 reads
    best = None
    for dpf in range(0, 16):
        pitch_f = ph_max * pw_max + dpf
        for dpc in range(0, 16):
            pitch_c = pf * pitch_f + dpc
            rows = np.array([ci * pitch_c + t * pitch_f for ci in range(ncl) for t in range(T)], dtype=np.int64)
            key = (_conflict_cycles(rows), pitch_c)
            if best is None or key < best[0]:
                best = (key, pitch_f, pitch_c)
    (cyc, _), pitch_f, pitch_c = best
    pitch_h = pw_max
    types, boxes, widx_all = [], [], []
    for wi, wd in enumerate(win):
        a_off = np.zeros(4 * 32, dtype=np.int64)
        for i, (oh, ow) in enumerate(wd["pos"]):
            for ci in range(ncl):
                for t in range(T):
                    r = ci * T + t          # row r of the tile: frames 2 tau, 2 tau + 1 are registers k, k + 1 of one lane
                    a_off[i * 32 + r] = ci * pitch_c + t * pitch_f + (2 * oh - 3 - wd["h0"]) * pitch_h + (2 * ow - 3 - wd["w0"])
        taps = wd["taps"] + [None] * (2 * S - len(wd["taps"]))
        masks = wd["masks"] + [15] * (S // 4 - len(wd["masks"]))
        tap_off = np.zeros(2 * S, dtype=np.int64)
        last = (0, 0, 0)
        for k, tp in enumerate(taps):       # a zero-weight tap reads where a tap of its group reads (any in-range slot would do)
            if tp is None:
                grp = [x for x in taps[(k // 8) * 8:(k // 8) * 8 + 8] if x is not None]
                tp_r = grp[0] if grp else None
            else:
                tp_r = tp
            if tp_r is None:
                tap_off[k] = 0
                continue
            kt, kh, kw = tp_r
            tap_off[k] = kt * pitch_f + kh * pitch_h + kw
        # groups nothing takes part in must still read inside the patch if a prefetch touches them: they are skipped by all tiles
        out = -np.ones(16, dtype=np.int64)          # entry e: rows 2 e, 2 e + 1 of every tile = (clip e // (T / 2), frame pair e % (T / 2))
        for ci in range(ncl):
            for tau in range(To):
                out[(ci * T + 2 * tau) // 2] = ci * feat_stride + tau * Ho * Wo
        tap_tab = np.concatenate([tap_off * SLOT_BYTES, np.array(masks, dtype=np.int64)])
        types.append(BoxType(pf, ph_max, pw_max, pitch_h, pitch_f, pitch_c, 4, (a_off * SLOT_BYTES).astype(np.int32),
                             out.astype(np.int32), tap_tab.astype(np.int32), float(cyc)))
        boxes.append([wi, -1, wd["h0"], wd["w0"], wd["a"] * Wo + wd["b"], 0])
        idx = -np.ones((CC, S, NT, 64, 8), dtype=np.int64)
        for s in range(S):
            for hh in range(2):
                tp = taps[2 * s + hh]
                if tp is None:
                    continue
                kt, kh, kw = tp
                lanes = np.where(half == hh)[0]
                for nt in range(NT):
                    n = nt * 32 + col[lanes]
                    for cc in range(CC):
                        c = cc * 8 + np.arange(8)
                        idx[cc, s, nt, lanes, :] = (((n[:, None] * cin + c[None, :]) * KT + kt) * KH + kh) * KW + kw
        widx_all.append(idx)
    widx = np.stack(widx_all).reshape(len(win) * CC, S, NT, 64, 8).astype(np.int32)
    ntaps_used = sum(len([t for t in wd["taps"] if t is not None]) for wd in win)
    plan = ConvPlan(name=name, CC=CC, F=t_in, H=h_in, W=w_in, row_pitch4=w_in * 4, chunk_stride4=t_in * h_in * w_in * 4,
                    clip_stride4=CC * t_in * h_in * w_in * 4, NT=NT, MW=1, MTW=4, S=S, ncl=ncl,
                    boxes=np.asarray(boxes, dtype=np.int32), types=types, widx=widx, epi=EPI_POS_FEAT, pool_t=2, relu=True,
                    n_out=cout, n_stride=npos, out_clip_stride=feat_stride, out_chunk_stride=0, out_shape=(cout, To, Ho, Wo),
                    w_box_stride=CC * S * NT * 64 * 8, rows_total=len(win) * 4 * 32 // ncl, rows_useful=T * 4 * Ho * Wo,
                    meta={"box": (T, 2, 2), "macs_per_unit": int(T * 4 * Ho * Wo) * KT * KH * KW * cin * cout, "pos_tiles": 1,
                          "mfma_per_chunk": [int(sum(4 * (4 - bin(m).count("1")) for m in wd["masks"])) for wd in win]})
    return plan


def pix_row_pitch(w: int) -> int:
    """Elements per 16-bit pixel row written by vd_pix2rows: 3 leading zeros + W pixels + zero
    tail, rounded up to a multiple of 8 (16 bytes)."""
    return -(-(w + 8) // 8) * 8


def l0_frame_tiles() -> bool:
    """VD_L0_FRAME_TILES (0: the round-4 layout, frame-pair row groups -- A/B measurements), read at every planning call like
    csrc/planner.cpp does, and part of the planners' memo key: both planners always see the same value."""
    return os.environ.get("VD_L0_FRAME_TILES", "1") == "1"


@_memo
def plan_forward_pix(name: str, cout: int, t_in: int, h_in: int, w_in: int, lds_budget: int = 3700,
                     mtw_options=(4,), ntw: int = 1) -> ConvPlan:
    """First layer: Conv3d(3->cout) + ReLU + MaxPool(1,2,2).  Source = 16-bit pixel rows made by
    vd_pix2rows: [clip][t*3+c][h][W+8] with 3 zero pixels in front (and >=5 behind), so that the
    'kw-slot' of output column ow -- x[t,c,h,2*ow-3 .. 2*ow+4], the 7 kernel columns plus one
    zero-weight tap -- is the 16 bytes at dword offset ow of the row: overlapping windows are
    read straight out of HBM/L2 by the LDS-DMA, nothing is duplicated in memory.

    FRAME TILES (round 5; whenever the pooled width is a multiple of 4 and the box is 4 x 8 x 8 conv rows): an MFMA tile holds 32
    conv rows of ONE frame (four row groups of two pool windows, ``_build_type_order``), the four tiles of a wave row are the same
    positions in four consecutive frames, and the K steps are ordered (tap pair j, kt) with both taps of a pair in the same kt plane.  The A
    fragment of (frame t + kt, pair j) then serves tile t at kt, tile t + 1 at kt - 1 and tile t + 2 at kt - 2: a kernel that
    knows this (conv0_breg5_kernel) reads 68 fragments per box and wave from LDS instead of 128, for the same 128 MFMAs in
    the same order per output as the generic kernel, which simply runs the steps in sequence."""
    cin = 3
    T = conv_out_dim(t_in, KT, 1, 1); OH = conv_out_dim(h_in, KH, 2, 3); OW = conv_out_dim(w_in, KW, 2, 3)
    Ho, Wo = OH // 2, OW // 2
    rows = (T, Ho * 2, Wo * 2)
    NT = cout // 32
    MW = max(1, 4 // (NT // ntw))
    if ntw == 2:                 # one wave column of 2 N tiles, 4 wave rows of 2 M tiles: same 8-tile box
        assert NT == 2
        mtw_options = (2,)
    col, half = _lane_cols()

    def widx_fn(CC_, S, NT_, taps_p, ntaps):
        idx = -np.ones((1, S, NT_, 64, 8), dtype=np.int64)
        for s in range(S):
            for hh in range(2):
                p = 2 * s + hh
                if p >= ntaps:
                    continue
                fc, kh, _ = taps_p[p]
                kt, c = divmod(fc, cin)
                lanes = np.where(half == hh)[0]
                for nt in range(NT_):
                    n = nt * 32 + col[lanes]
                    kw = np.arange(7)
                    flat = (((n[:, None] * cin + c) * KT + kt) * KH + kh) * KW + kw[None, :]
                    idx[0, s, nt, lanes, :7] = flat
        return idx.astype(np.int32)

    chunk_stride = T * Ho * Wo
    clip_stride = (cout // 8) * chunk_stride

    def out_index(ci, a, b, c):
        return ci * clip_stride + (a * Ho + b // 2) * Wo + c // 2
    force_box = tuple(int(v) for v in os.environ["VD_L0_BOX"].split(",")) if os.environ.get("VD_L0_BOX") else None
    plan = None
    if l0_frame_tiles() and ntw == 1 and NT == 2 and tuple(mtw_options) == (4,) and Wo % 4 == 0:
        # K order: (tap pair j, kt) for the 10 pairs of a kt plane's first 20 (c, kh) taps, then the three left-over taps
        q = [(c, kh) for c in range(cin) for kh in range(KH)]
        taps = [(kt * cin + q[2 * j + e][0], q[2 * j + e][1], 0) for j in range(10) for kt in range(KT) for e in range(2)]
        taps += [(kt * cin + q[20][0], q[20][1], 0) for kt in range(KT)]
        try:
            plan = _make_plan(name, (t_in * cin, h_in, OW), 1, rows, (1, 2, 8), (-cin, -3, 0), (cin, 2, 1), taps,
                              widx_fn, cout, NT, MW, mtw_options, EPI_POOL_CL, 1, True, out_index, None, 0,
                              clip_stride, chunk_stride, (cout // 8, T, Ho, Wo, 8), lds_budget, (1,), ntw=ntw,
                              force_box=force_box, frame_tiles=True)
            t0 = plan.types[0]
            plan.out_t_stride, plan.pair_flip = FRAME_TILE_OUT_STEP, FRAME_TILE_FLIP | (t0.pitch_h << 8) | (t0.pitch_f << 16)
            plan.meta["frame_tiles"] = 1
            fs = cin * t0.pitch_f * SLOT_BYTES
            for t in plan.types:          # what the frame-sharing kernel relies on (a violation falls back to the frame-pair layout)
                a = t.a_off.reshape(MW, 4, 32)
                if not all((a[:, i] - a[:, 0] == i * cin * t.pitch_f * SLOT_BYTES).all() for i in range(4)):
                    raise ValueError("frame tiles: tile i != tile 0 + i frames")
                tp = t.tap_off.reshape(-1, 2)
                if not all((tp[3 * j + kt] - tp[3 * j] == kt * fs).all() for j in range(10) for kt in range(KT)):
                    raise ValueError("frame tiles: tap 3 j + kt != tap 3 j + kt frames")
        except ValueError:
            plan = None
    if plan is None:
        assert T % 2 == 0, "first-layer planner pairs frames (pool window rows are 2x2x2 groups)"
        taps = [(kt * cin + c, kh, 0) for kt in range(KT) for c in range(cin) for kh in range(KH)]
        plan = _make_plan(name, (t_in * cin, h_in, OW), 1, rows, (2, 2, 2), (-cin, -3, 0), (cin, 2, 1), taps,
                          widx_fn, cout, NT, MW, mtw_options, EPI_POOL_CL, 1, True, out_index, None, 0,
                          clip_stride, chunk_stride, (cout // 8, T, Ho, Wo, 8), lds_budget, (1,), ntw=ntw,
                          force_box=force_box)
        plan.out_t_stride = Ho * Wo
    plan.NTW = ntw
    rowp = pix_row_pitch(w_in)
    plan.w_step4, plan.row_pitch4 = 1, rowp // 2
    plan.chunk_stride4 = plan.clip_stride4 = t_in * cin * h_in * (rowp // 2)
    plan.meta["macs_per_unit"] = T * OH * OW * cin * 3 * 7 * 7 * cout        # (the kw-slot carries an eighth, zero-weight tap)
    return plan


def dgrad_classes(h: int, w: int):
    """Parity classes of a stride-2 7x7 convolution's input positions."""
    return [(ph, pw) for ph in range(min(2, h)) for pw in range(min(2, w))]


@_memo
def plan_dgrad(name: str, cin: int, cout: int, t_in: int, h_in: int, w_in: int, ph: int, pw: int,
               pixel_out: bool, lds_budget: int = 3700, mtw_options=(7, 8, 4, 2)) -> ConvPlan:
    """Input gradient of Conv3d(cin->cout, k(3,7,7), s(1,2,2), p(1,3,3)) for the input positions
    (t, 2b+ph, 2c+pw).  Source: dense dy on the conv grid, channels-last chunks of cout.
    dx[t,h,w,ci] = sum_{kt,kh,kw,n} dy[t+1-kt, (h+3-kh)/2, (w+3-kw)/2, n] * W[n,ci,kt,kh,kw]."""
    assert cout % 8 == 0
    CC = cout // 8
    T = conv_out_dim(t_in, KT, 1, 1); OH = conv_out_dim(h_in, KH, 2, 3); OW = conv_out_dim(w_in, KW, 2, 3)
    khs = [kh for kh in range(KH) if (ph + 3 - kh) % 2 == 0]
    kws = [kw for kw in range(KW) if (pw + 3 - kw) % 2 == 0]
    # e = (p+3-k)/2 in [-1,2]; tap offset d = e+1 with row origin -1
    tap_k = [(kt, kh, kw) for kt in range(KT) for kh in khs for kw in kws]
    taps = [(2 - kt, (ph + 3 - kh) // 2 + 1, (pw + 3 - kw) // 2 + 1) for kt, kh, kw in tap_k]
    RB = (h_in - ph + 1) // 2; RC = (w_in - pw + 1) // 2
    rows = (t_in, RB, RC)
    n_pad = -(-cin // 32) * 32
    NT = n_pad // 32
    MW = max(1, 4 // NT)
    col, half = _lane_cols()

    def widx_fn(CC_, S, NT_, taps_p, ntaps):
        idx = -np.ones((CC_, S, NT_, 64, 8), dtype=np.int64)
        for s in range(S):
            for hh in range(2):
                p = 2 * s + hh
                if p >= ntaps:
                    continue
                kt, kh, kw = tap_k[p]
                lanes = np.where(half == hh)[0]
                for nt in range(NT_):
                    ci = nt * 32 + col[lanes]
                    ok = ci < cin
                    for cc in range(CC_):
                        n = cc * 8 + np.arange(8)
                        flat = (((n[None, :] * cin + ci[:, None]) * KT + kt) * KH + kh) * KW + kw
                        flat = np.where(ok[:, None], flat, -1)
                        idx[cc, s, nt, lanes, :] = flat
        return idx.astype(np.int32)

    if pixel_out:   # (T, C, H, W) fp32 pixels of one clip, C = cin
        clip_stride = t_in * cin * h_in * w_in
        n_stride = h_in * w_in

        def out_index(ci, a, b, c):
            return ci * clip_stride + (a * cin * h_in + (2 * b + ph)) * w_in + 2 * c + pw
        out_shape = (t_in, cin, h_in, w_in)
    else:           # fp32 [t][h][w][cin]
        clip_stride = t_in * h_in * w_in * cin
        n_stride = 1

        def out_index(ci, a, b, c):
            return ci * clip_stride + ((a * h_in + (2 * b + ph)) * w_in + 2 * c + pw) * cin
        out_shape = (t_in, h_in, w_in, cin)
    max_ncl = max(1, (8 * 32 * MW) // (rows[0] * rows[1] * rows[2]))
    ncl_options = sorted({1, max_ncl} | {n for n in (2, 4, 8) if n <= max_ncl})
    plan = _make_plan(name, (T, OH, OW), CC, rows, (1, 1, 1), (-1, -1, -1), (1, 1, 1), taps, widx_fn,
                      cin, NT, MW, mtw_options, EPI_ROWS, 0, False, out_index, None, n_stride,
                      clip_stride, 0, out_shape, lds_budget, ncl_options)
    plan.meta.update({"ph": ph, "pw": pw})
    return plan


@_memo
def plan_dgrad_pix(name: str, cin: int, cout: int, t_in: int, h_in: int, w_in: int, lds_budget: int = 3700,
                   mtw_options=(7, 8), bw: int = 2) -> ConvPlan:
    """Input gradient of the FIRST layer (cin = 3 pixel channels) with the stride-2 parity classes merged into the N
    dimension: one output row = the 2 x ``bw`` pixel block (t, 2b..2b+1, bw*c..bw*c+bw-1), its cin*2*bw values are the GEMM
    columns n = (ci*2 + ph)*bw + pw.  All of them read the same neighbourhood of dy, so K = taps x cout with structural zeros
    in B where a tap does not reach a pixel.
      bw = 2: 2x2 blocks, N = 12 of 32 columns, 3x4x4 = 48 taps;
      bw = 4: 2x4 blocks, N = 24 of 32 columns, 3x4x5 = 60 taps, half as many rows: 0.625 of the MFMA work per pixel (the
              dy window of a block grows by one column while the block doubles)."""
    assert bw in (2, 4) and cin * 2 * bw <= 32 and cout % 8 == 0 and h_in % 2 == 0 and w_in % bw == 0
    CC = cout // 8
    T = conv_out_dim(t_in, KT, 1, 1); OH = conv_out_dim(h_in, KH, 2, 3); OW = conv_out_dim(w_in, KW, 2, 3)
    taps = [(dt, dh, dw) for dt in range(3) for dh in range(4) for dw in range(bw // 2 + 3)]
    rows = (t_in, h_in // 2, w_in // bw)
    NT, MW = 1, 4
    col, half = _lane_cols()

    def widx_fn(CC_, S, NT_, taps_p, ntaps):
        idx = -np.ones((CC_, S, NT_, 64, 8), dtype=np.int64)
        for s in range(S):
            for hh in range(2):
                p = 2 * s + hh
                if p >= ntaps:
                    continue
                dt, dh, dw = taps_p[p]
                kt = 2 - dt
                lanes = np.where(half == hh)[0]
                n = col[lanes]
                ci, ph, pw = n // (2 * bw), (n // bw) % 2, n % bw
                kh = ph + 5 - 2 * dh          # from oh = b + (dh-1) = (2b+ph+3-kh)/2
                kw = pw + 5 - 2 * dw          # from ow = (bw/2)*c + (dw-1) = (bw*c+pw+3-kw)/2
                ok = (ci < cin) & (kh >= 0) & (kh < KH) & (kw >= 0) & (kw < KW)
                for cc in range(CC_):
                    nn = cc * 8 + np.arange(8)
                    flat = (((nn[None, :] * cin + ci[:, None]) * KT + kt) * KH + kh[:, None]) * KW + kw[:, None]
                    idx[cc, s, 0, lanes, :] = np.where(ok[:, None], flat, -1)
        return idx.astype(np.int32)

    clip_stride = t_in * cin * h_in * w_in

    def out_index(ci, a, b, c):
        return ci * clip_stride + (a * cin * h_in + 2 * b) * w_in + bw * c
    max_ncl = 1
    plan = _make_plan(name, (T, OH, OW), CC, rows, (1, 1, 1), (-1, -1, -1), (1, 1, bw // 2), taps, widx_fn,
                      cin * 2 * bw, NT, MW, mtw_options, EPI_ROWS, 0, False, out_index, None, 0,
                      clip_stride, 0, (t_in, cin, h_in, w_in), lds_budget, (max_ncl,))
    n = np.arange(32)
    plan.col_off = np.where(n < cin * 2 * bw, (n // (2 * bw)) * h_in * w_in + ((n // bw) % 2) * w_in + (n % bw), 0).astype(np.int32)
    plan.meta["macs_per_unit"] = T * OH * OW * cin * 3 * 7 * 7 * cout        # all four parity classes = the layer's forward count
    plan.meta["block_w"] = bw
    return plan


def bwd0_block_w(w_in: int) -> int:
    """Block width of the merged first-level input-gradient program: 4 where the clip width allows (VD_BWD0_WIDE=0: the 2 x 2
    blocks of rounds 1-3, for A/B runs; csrc/planner.cpp reads the same switch)."""
    return 4 if (w_in % 4 == 0 and os.environ.get("VD_BWD0_WIDE", "1") != "0") else 2


WGRAD_BLOCKS = ((8, 4, 4), (4, 4, 14), (8, 2, 14), (4, 7, 7), (2, 7, 14), (4, 2, 14), (2, 4, 14), (4, 4, 4), (2, 2, 14), (1, 4, 14), (2, 4, 4), (1, 2, 14))


def plan_wgrad(name: str, cin: int, cout: int, t_in: int, h_in: int, w_in: int, nclips: int,
               lds_budget: int = 3700, block=None, planes: int = 1, ordered: bool = False) -> ConvPlan:
    """``ordered``: every box of positions accumulates into its OWN copy (replicas = boxes) -- an fp32 atomic add onto a
    zeroed word with a single contributor is exact, and vd_replica_sum folds the copies in index order: the weight gradient
    becomes bitwise reproducible (the deterministic training step).  Same boxes, same tile program."""
    if ordered:
        import copy
        base = plan_wgrad(name, cin, cout, t_in, h_in, w_in, nclips, lds_budget, block, planes)
        pl = copy.copy(base)
        pl.boxes = base.boxes.copy()
        pl.boxes[:, 5] = np.arange(pl.boxes.shape[0])
        pl.meta = dict(base.meta, replicas=int(pl.boxes.shape[0]))
        return pl
    if block is None:
        # Among the blocks of positions whose patch fits the LDS budget and the DMA-group budget of the workgroup (cout/32
        # waves x 17 groups of 64 slots), the one with the smallest  (K steps incl. the padding of partial blocks) /
        # (resident waves per CU, counted up to 6 of the 8 the register budget allows -- 4 when the K loop is a single
        # clip chunk): residency is limited by the patch (`planes` 16-bit planes in LDS) and by how many workgroups the
        # launch has at all.  Ties: fewest boxes (atomic passes), longest K loop.  Measured per 64 clips 112x112x16,
        # hi+lo formats: layer 1 (8,2,14) 1.45 -> (4,2,14) 1.23 ms; layer 0 (4,2,14) 0.79 -> (2,2,14) 0.65 ms; layer 2
        # (8,4,4) 0.19 -> (2,4,4) 0.13 ms; 256 clips 64x64x8: layer 1 (2,7,8) 1.16 / (8,4,4) 0.82 / (4,4,4) 0.78 ms.
        best = None
        wg_waves = cout // 32
        want = 6 if nclips > 8 else 4
        for cand in WGRAD_BLOCKS:
            pl = _plan_wgrad(name, cin, cout, t_in, h_in, w_in, nclips, lds_budget, cand)
            bt = pl.types[0]
            groups = -(-bt.pitch_c // 64)
            if bt.pitch_c > lds_budget or groups > (cout // 32) * 17:
                continue
            lds = planes * bt.pitch_c * SLOT_BYTES + 2048
            resident = min(8, (160 * 1024 // lds) * wg_waves, cin * pl.nbox * wg_waves // 256)
            key = (pl.nbox * pl.S / max(1, min(resident, want)), pl.nbox, -pl.S)
            if best is None or key < best[0]:
                best = (key, pl)
        assert best is not None, "no weight-gradient block fits"
        return best[1]
    return _plan_wgrad(name, cin, cout, t_in, h_in, w_in, nclips, lds_budget, block)


@_memo
def _plan_wgrad(name: str, cin: int, cout: int, t_in: int, h_in: int, w_in: int, nclips: int,
                lds_budget: int, block) -> ConvPlan:
    """Weight gradient of Conv3d(cin->cout, k(3,7,7), s(1,2,2), p(1,3,3)) as a tile program of the SAME
    kernel, with the roles of the operands rotated:

        dW[n, c, tap] = sum_{clip, pos} dy[clip, pos, n] * x[clip, c, pos*stride + tap]

      * a 'clip' of the program is one INPUT CHANNEL c; its output rows are the 147 taps;
      * the K loop runs over (position in a block) x (8 clips per slot): the source is x in a
        clip-minor layout [c][clip/8][t][h][w][8 clips] (vd_clip_minor), the 'channel chunks' are
        the clip chunks, the program's taps are the block's positions (stride-2 offsets);
      * a box is a block of output positions; every box of a channel adds into the same 147 x cout
        rows (fp32 atomics), and its B operand -- dy of its positions, packed per box by
        vd_pack_dy -- sits at box * w_box_stride.
    The A address stays  a_off[row = tap] + tap_off[step position]  (both additive in the patch)."""
    assert cout % 32 == 0
    T = conv_out_dim(t_in, KT, 1, 1); OH = conv_out_dim(h_in, KH, 2, 3); OW = conv_out_dim(w_in, KW, 2, 3)
    nt, noh, now = block
    nt, noh, now = min(nt, T), min(noh, OH), min(now, OW)
    if (nt * noh * now) % 2:
        now += 1                                  # K steps take position pairs; the extra column is masked by zero dy
    CCb = -(-nclips // 8)
    NT = cout // 32
    MW = 1
    MTW = 5                                      # 147 taps -> 5 row tiles of 32
    pf, ph, pw = nt + 2, 2 * (noh - 1) + 7, 2 * (now - 1) + 7
    positions = [(dt, doh, dow) for dt in range(nt) for doh in range(noh) for dow in range(now)]
    S = len(positions) // 2
    best = None
    rows_tap = [(kt, kh, kw) for kt in range(KT) for kh in range(KH) for kw in range(KW)]
    for dph in range(0, 16):
        pitch_h = pw + dph
        for dpf in range(0, 16):
            pitch_f = ph * pitch_h + dpf
            pitch_c = pf * pitch_f
            if pitch_c > lds_budget * 1.12 and best is not None:
                continue
            a_off = np.zeros(MTW * 32, dtype=np.int64)
            for r, (kt, kh, kw) in enumerate(rows_tap):
                a_off[r] = kt * pitch_f + kh * pitch_h + kw
            cyc = float(np.mean([_conflict_cycles(a_off[t * 32:(t + 1) * 32]) for t in range(MTW)]))
            key = (round(cyc, 3), pitch_c)
            if best is None or key < best[0]:
                out = -np.ones(MTW * 32, dtype=np.int64)
                out[:len(rows_tap)] = np.arange(len(rows_tap)) * cout      # accumulation target [cin][tap][cout]: cout minor
                tap_off = np.array([(dt * pitch_f + 2 * doh * pitch_h + 2 * dow) * SLOT_BYTES for dt, doh, dow in positions])
                best = (key, BoxType(pf, ph, pw, pitch_h, pitch_f, pitch_c, MTW, (a_off * SLOT_BYTES).astype(np.int32),
                                     out.astype(np.int32), tap_off.astype(np.int32), cyc))
            if cyc <= 4.0 + 1e-9:
                break
        if best[0][0] <= 4.0 + 1e-9:
            break
    bt = best[1]
    boxes, blocks = [], []
    for t0 in range(0, T, nt):
        for oh0 in range(0, OH, noh):
            for ow0 in range(0, OW, now):
                boxes.append([0, t0 - 1, 2 * oh0 - 3, 2 * ow0 - 3, 0, 0])
                blocks.append((t0, oh0, ow0))
    # Every box of a channel adds into the same 147 x cout floats with fp32 atomics.  Two things make that affordable:
    # the target is laid out [cin][tap][cout] (a wave's 32 output channels share one 128-byte line; in dW's own layout
    # they are 441..18816 floats apart and the chip sustains only ~30 G scattered atomics/s: 0.4 ms of a 1.5 ms layer-1
    # launch), and boxes are dealt round-robin to `replicas` copies (box row word 5; same-address atomics serialise).
    # WgradOp folds the copies into dW afterwards (vd_replica_sum, a tiled transpose).
    replicas = max(1, min(16, len(boxes) // 28))
    for bi, box in enumerate(boxes):
        box[5] = bi % replicas
    plan = ConvPlan(name=name, CC=CCb, F=t_in, H=h_in, W=w_in, row_pitch4=w_in * 4, chunk_stride4=t_in * h_in * w_in * 4,
                    clip_stride4=CCb * t_in * h_in * w_in * 4, NT=NT, MW=MW, MTW=MTW, S=S, ncl=1,
                    boxes=np.asarray(boxes, dtype=np.int32), types=[bt], widx=np.zeros((0,), dtype=np.int32),
                    epi=EPI_ROWS, pool_t=0, relu=False, n_out=cout, n_stride=1,
                    out_clip_stride=KT * KH * KW * cout, out_chunk_stride=0, out_shape=(cin, KT, KH, KW, cout),
                    rows_total=len(boxes) * MTW * 32, rows_useful=len(boxes) * len(rows_tap), meta={"box": (nt, noh, now)})
    plan.atomic = True
    plan.w_box_stride = CCb * S * NT * 64 * 8
    plan.meta.update({"blocks": blocks, "positions": positions, "grid": (T, OH, OW), "nclips": nclips, "cout": cout, "replicas": replicas,
                      "macs_per_unit": T * OH * OW * KT * KH * KW * cout * nclips})       # unit = one input channel
    return plan


def wgrad_pack_index(plan: ConvPlan) -> np.ndarray:
    """Gather table of vd_pack_dy's output for tests: [nbox, CCb, S, NT, 64, 8] -> (clip, t, oh, ow, n)
    flat index into dy[clip][t][oh][ow][n] (channels-last logical order) or -1."""
    T, OH, OW = plan.meta["grid"]
    nclips, cout = plan.meta["nclips"], plan.meta["cout"]
    positions = plan.meta["positions"]
    lane = np.arange(64); col, half = lane & 31, lane >> 5
    out = -np.ones((plan.nbox, plan.CC, plan.S, plan.NT, 64, 8), dtype=np.int64)
    for bi, (t0, oh0, ow0) in enumerate(plan.meta["blocks"]):
        for s in range(plan.S):
            for hh in range(2):
                dt, doh, dow = positions[2 * s + hh]
                t, oh, ow = t0 + dt, oh0 + doh, ow0 + dow
                if t >= T or oh >= OH or ow >= OW:
                    continue
                lanes = np.where(half == hh)[0]
                for nt in range(plan.NT):
                    n = nt * 32 + col[lanes]
                    for cb in range(plan.CC):
                        clip = cb * 8 + np.arange(8)
                        flat = (((clip[None, :] * T + t) * OH + oh) * OW + ow) * cout + n[:, None]
                        out[bi, cb, s, nt, lanes, :] = np.where(clip[None, :] < nclips, flat, -1)
    return out


# ----------------------------------------------------------------------------------------
# whole network
# ----------------------------------------------------------------------------------------
@dataclass
class NetGeometry:
    frames: int
    height: int
    width: int
    channel: int = 3
    widths: Tuple[int, ...] = (64, 128, 128)
    pools_t: Tuple[int, ...] = (1, 2, 2)

    def layer_dims(self):
        """[(cin, cout, t_in, h_in, w_in, T, OH, OW, To, Ho, Wo, pool_t)] per layer."""
        out = []
        cin, t, h, w = self.channel, self.frames, self.height, self.width
        for cout, pt in zip(self.widths, self.pools_t):
            T = conv_out_dim(t, KT, 1, 1); OH = conv_out_dim(h, KH, 2, 3); OW = conv_out_dim(w, KW, 2, 3)
            To, Ho, Wo = T // pt, OH // 2, OW // 2
            out.append((cin, cout, t, h, w, T, OH, OW, To, Ho, Wo, pt))
            cin, t, h, w = cout, To, Ho, Wo
        return out

    @property
    def num_feat(self) -> int:
        d = self.layer_dims()[-1]
        return d[1] * d[8] * d[9] * d[10]


_PLAN_CACHE: Dict[Tuple, Dict[str, object]] = {}


SMALL_TILES_FIRST = (2, 4, 7, 8)
TINY_GRID = int(os.environ.get("VD_TINY_GRID", "128"))      # latency_variant: launches below this many workgroups ignore padding (0: off)


def latency_variant(pl: ConvPlan, batch_hint: Optional[int], make) -> ConvPlan:
    """For a launch of ``batch_hint`` clips that would start fewer workgroups than the chip has slots (2 per CU), the
    time is one workgroup's latency -- K steps x MFMAs per step: prefer the decomposition with the smallest M tile
    count per wave (same padded work, more and shorter workgroups).  ``make(mtw_options)`` rebuilds the program."""
    if not batch_hint or pl.grid(batch_hint) >= 512:
        return pl
    slack = float(os.environ.get("VD_LAT_SLACK", "1.05"))     # a few per cent of padding rows are cheaper than a long workgroup
    best = pl
    for opts in (SMALL_TILES_FIRST, (4,), (2,)):
        try:
            alt = make(opts)
        except ValueError:
            continue
        if alt.grid(batch_hint) <= best.grid(batch_hint):
            continue
        # (round 4) a TINY launch -- fewer workgroups than a quarter of the chip's 512 slots -- is one workgroup's latency whatever
        # it pads: fewer M tiles per wave win even where the rows do not fill them (the (1,1) parity class of the last level's
        # input gradient: 72 rows per clip, 7 tiles x 3 clips per box = 17 workgroups of 172 us for 50 clips; 2-tile boxes: 100
        # workgroups that also join the other classes' launch)
        tiny = best.grid(batch_hint) < TINY_GRID and alt.MTW < best.MTW and alt.grid(batch_hint) <= 512
        if alt.rows_total <= pl.rows_total * slack or tiny:
            best = alt
    return best


def plan_network(geo: NetGeometry, lds_budget: int = 3700, ntw: int = 1, ntw0: int = 1, balanced: bool = False,
                 batch_hint: Optional[int] = None, bwd0_small: bool = False) -> Dict[str, object]:
    """All tile programs of one ConvNet3D geometry: forward L0..L2 and the input-gradient
    passes (one per parity class per layer).  ``ntw`` / ``ntw0`` = N tiles per wave of the layer-1/2
    and of the first-layer forward programs; ``batch_hint`` = clips per launch the programs will typically see
    (small batches get latency-oriented decompositions, see ``latency_variant``)."""
    key = (geo.frames, geo.height, geo.width, geo.channel, geo.widths, geo.pools_t, lds_budget, ntw, ntw0, balanced, batch_hint, bwd0_small,
           bwd0_block_w(geo.width), l0_frame_tiles())
    if key in _PLAN_CACHE:
        return _PLAN_CACHE[key]
    dims = geo.layer_dims()
    assert geo.channel == 3 and geo.pools_t[0] == 1, "first layer planner assumes RGB clips and (1,2,2) pooling"
    fwd = [plan_forward_pix("fwd0", dims[0][1], dims[0][2], dims[0][3], dims[0][4],
                            int(os.environ.get("VD_L0_BUDGET", lds_budget)), ntw=ntw0)]
    for li in (1, 2):
        cin, cout, t, h, w = dims[li][:5]
        pl = plan_forward_cl("fwd%d" % li, cin, cout, t, h, w, dims[li][11], feat_out=(li == 2), lds_budget=lds_budget)
        if ntw == 2 and cout == 128:
            # two N tiles per wave halve the LDS reads per MFMA (measured +20 % on layer 2), but the workgroup then
            # has 8 M tiles: only worth it where the boxes fill them (layer 1's 7-tile boxes would pad 1/8)
            pl2 = plan_forward_cl("fwd%d" % li, cin, cout, t, h, w, dims[li][11], feat_out=(li == 2), lds_budget=lds_budget,
                                  ntw=2)
            if pl2.rows_total <= pl.rows_total:
                pl = pl2
            elif balanced and (pl.NT, pl.MW, pl.MTW) == (4, 1, 7) and all(t.mt == 7 for t in pl.types):
                # 7-tile boxes: same tables, but executed by 2 x 2 waves of 3 M tiles x 2 N tiles + one N tile of the
                # seventh M tile each (kernel template BAL): 7 MFMAs per K step for 4 A reads instead of 7
                pl = dataclasses.replace(pl, MW=2, MTW=3, NTW=2, meta=dict(pl.meta, balanced=1))
        pl = latency_variant(pl, batch_hint, lambda opts, li=li, cin=cin, cout=cout, t=t, h=h, w=w: plan_forward_cl(
            "fwd%d" % li, cin, cout, t, h, w, dims[li][11], feat_out=(li == 2), lds_budget=lds_budget, mtw_options=opts))
        fwd.append(pl)
    bwd = []
    for li in range(3):
        cin, cout, t, h, w = dims[li][:5]
        if li == 0 and h % 2 == 0 and w % 2 == 0:
            # ``bwd0_small`` (the engines of the training / second-order passes): boxes of 4 M tiles per wave within 1800 LDS slots --
            # a hi+lo program then holds 2 x 21 KB of patch and two or three workgroups share a CU (7-tile boxes: 2 x 53 KB, one
            # workgroup, one wave per SIMD).  Embed backward of 256 clips 64x64x8 2.29 -> 2.00 ms, second-order pass 10.0 -> 9.2 ms,
            # MTT+Ours 7.77 -> 8.23 it/s.  NOT for the synthetic-clip stream of DM, whose kernels run under the real side's: there the
            # extra resident workgroups take more from the real stream than they save (step 35.6 -> 36.2 ms, N = 8 proxy 4.9 -> 5.4).
            # 2 x 4 pixel blocks (N = 24 of 32 columns, 60 taps) where the width allows: 0.625 of the 2 x 2 blocks' MFMA work
            bw = bwd0_block_w(w)
            small = None
            if bwd0_small:
                try:
                    small = plan_dgrad_pix("bwd0_merged", cin, cout, t, h, w, lds_budget=min(lds_budget, 1800), mtw_options=(4,), bw=bw)
                except ValueError:
                    small = None
            bwd.append([small if small is not None else plan_dgrad_pix("bwd0_merged", cin, cout, t, h, w, lds_budget=lds_budget, bw=bw)])
            continue
        layer = []
        for ph, pw in dgrad_classes(h, w):
            def make(opts, li=li, cin=cin, cout=cout, t=t, h=h, w=w, ph=ph, pw=pw):
                return plan_dgrad("bwd%d_%d%d" % (li, ph, pw), cin, cout, t, h, w, ph, pw, pixel_out=(li == 0),
                                  lds_budget=lds_budget, mtw_options=opts)
            layer.append(latency_variant(make((7, 8, 4, 2)), batch_hint, make))
        bwd.append(layer)
    out = {"geo": geo, "dims": dims, "fwd": fwd, "bwd": bwd}
    _PLAN_CACHE[key] = out
    return out
