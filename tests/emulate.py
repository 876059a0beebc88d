"""numpy interpreter of the tile program that csrc/conv_mfma.hip executes on the GPU.

It consumes exactly the tables the device consumes (ConvPlan.flat_tables(), boxes, widx) and
follows the same dataflow (patch gather with zero fill -> per-row/tap A operand -> K-step
products -> pooled / row epilogue), in fp64, so that the planner can be checked on the CPU.
"""
import numpy as np

from video_distillation_amd import plan as P


def pack_weights(plan, w_flat):
    """[CC,S,NT,64,8] operand values (what vd_pack_weights writes, before 16-bit rounding)."""
    idx = plan.widx
    out = np.where(idx >= 0, w_flat[np.maximum(idx, 0)], 0.0)
    return out


def run_plan(plan, src, w_flat, bias, nclips, out):
    """src: float array [nclips, CC, ...] whose trailing dims flatten to the 16-bit ELEMENTS of one
    channel chunk (2 per dword); out: flat float64 array (pre-zeroed).
    Returns argmax array (same indexing as out) for pooled epilogues."""
    descs, tables = plan.flat_tables()
    wp = pack_weights(plan, w_flat)                       # [CC,S,NT,64,8]
    lane = np.arange(64)
    col, half = lane & 31, lane >> 5
    arg = {}
    gather = plan.gather_table()
    src_flat = src.reshape(src.shape[0], src.shape[1], -1)
    ngroups = -(-nclips // plan.ncl)
    for grp in range(ngroups):
        clip0 = grp * plan.ncl
        for bi, box in enumerate(plan.boxes):
            ty, f0, h0, w0, out_rel, _ = [int(v) for v in box]
            pf, ph, pw, pitch_h, pitch_f, pitch_c, mt, a_ofs, o_ofs, t_ofs = [int(v) for v in descs[ty][:10]]
            a_off = tables[a_ofs:a_ofs + mt * 32] // 16
            tap_off = tables[t_ofs:t_ofs + 2 * plan.S] // 16
            pos = plan.epi == P.EPI_POS_FEAT        # position tiles: skip masks behind the taps, B operands per box, pool across tiles
            skip = tables[t_ofs + 2 * plan.S:t_ofs + 2 * plan.S + plan.S // 4] if pos else None
            nout = mt * 4 if plan.epi != P.EPI_ROWS else mt * 32
            out_tab = tables[o_ofs:o_ofs + nout]
            acc = np.zeros((mt * 32, plan.NT * 32))
            for cc in range(plan.CC):
                patch = np.zeros((plan.ncl * pitch_c + 64, 8))
                gt = gather[bi]
                for idx in range(plan.ncl * pitch_c):
                    e = int(gt[idx])
                    if e < 0:
                        continue
                    b = clip0 + (e >> 24)
                    if b >= nclips:
                        continue
                    o2 = 2 * (e & 0xFFFFFF)                   # dword offset -> element offset
                    patch[idx] = src_flat[b, cc, o2:o2 + 8]
                # A[row, s, half, j]
                for s in range(plan.S):
                    for hh in range(2):
                        lanes = np.where(half == hh)[0]
                        if pos:
                            for i in range(mt):
                                if (int(skip[s // 4]) >> i) & 1:
                                    continue
                                sl = a_off[i * 32:(i + 1) * 32] + tap_off[2 * s + hh]
                                assert sl.min() >= 0 and sl.max() < plan.ncl * pitch_c, "position tile reads outside its patch"
                                for nt in range(plan.NT):
                                    acc[i * 32:(i + 1) * 32, nt * 32:(nt + 1) * 32] += patch[sl] @ wp[bi * plan.CC + cc, s, nt, lanes, :].T
                            continue
                        a = patch[a_off + tap_off[2 * s + hh]]            # [rows, 8]
                        for nt in range(plan.NT):
                            bmat = wp[cc, s, nt, lanes, :]                 # [32 cols, 8]
                            acc[:, nt * 32:(nt + 1) * 32] += a @ bmat.T
            n = np.arange(plan.NT * 32)
            nvalid = n < plan.n_out
            if bias is not None:
                acc = acc + np.where(nvalid, np.pad(bias, (0, max(0, plan.NT * 32 - bias.size)))[:plan.NT * 32], 0.0)[None, :]
            if plan.relu:
                acc = np.maximum(acc, 0.0)
            if pos:
                for e in range(16):
                    o = int(out_tab[e])
                    if o < 0 or clip0 + o // plan.out_clip_stride >= nclips:
                        continue
                    rows = [i * 32 + 2 * e + d for i in range(mt) for d in range(2)]
                    out[clip0 * plan.out_clip_stride + out_rel + o + n * plan.n_stride] = acc[rows].max(axis=0)
            elif plan.epi == P.EPI_ROWS:
                for r in range(mt * 32):
                    o = int(out_tab[r])
                    if o < 0:
                        continue
                    base = clip0 * plan.out_clip_stride + out_rel + o
                    ci = o // plan.out_clip_stride
                    if clip0 + ci >= nclips:
                        continue
                    coff = plan.col_off[n[nvalid]] if plan.col_off is not None else n[nvalid] * plan.n_stride
                    out[base + coff] = acc[r, nvalid]
            else:
                for t in range(mt):
                    for q in range(4):
                        o = int(out_tab[t * 4 + q])
                        if o < 0:
                            continue
                        ci = o // plan.out_clip_stride
                        if clip0 + ci >= nclips:
                            continue
                        rows = [t * 32 + P._row_of(q, j) for j in range(8)]
                        vals = acc[rows]                                   # [8, N]
                        sets = [(0, slice(0, 8))] if plan.pool_t == 2 else [(0, slice(0, 4)), (1, slice(4, 8))]
                        for dt, sl in sets:
                            v = vals[sl]
                            mx, am = v.max(axis=0), v.argmax(axis=0)
                            base = clip0 * plan.out_clip_stride + out_rel + o + dt * (-plan.out_t_stride if (plan.pair_flip >> q) & 1 else plan.out_t_stride)
                            if plan.epi == P.EPI_POOL_FEAT:
                                out[base + n * plan.n_stride] = mx
                                for k in n:
                                    arg[base + k * plan.n_stride] = am[k]
                            else:   # CL slots: chunk = n//8, element n%8
                                idxs = (base + (n // 8) * plan.out_chunk_stride) * 8 + (n % 8)
                                out[idxs] = mx
                                for k, ii in zip(n, idxs):
                                    arg[ii] = am[k]
    return arg


def pix_to_rows(x_btchw):
    """What vd_pix2rows produces: [B, 1, T*3, H, pitch] with 3 leading zero pixels per row."""
    B, T, C, H, W = x_btchw.shape
    pitch = P.pix_row_pitch(W)
    out = np.zeros((B, 1, T * C, H, pitch))
    out[..., 3:3 + W] = x_btchw.reshape(B, 1, T * C, H, W)
    return out
