"""Test tools: compare the pooling decisions a HIP forward recorded (arg-max bytes, csrc/conv_mfma.hip epilogue: bits 0-2 =
position in the window dt*4 + dh*2 + dw, bit 7 = ReLU-dead) with the fp64 oracle's max_pool3d decisions on the same
clips and weights, and measure how close to a tie every differing window is.  Imports the oracle: test infrastructure."""
import torch

from oracle import ref_cpu as R


def oracle_windows(z, pt):
    """z (B,C,T,H,W) conv output incl. bias -> (arg-max in the window [first maximum], top1 - top2, top1) per pooled element,
    windows of (pt,2,2) enumerated dt*4 + dh*2 + dw as the kernel does."""
    B, C, T, H, W = z.shape
    To, Ho, Wo = T // pt, H // 2, W // 2
    w = z[:, :, :To * pt, :Ho * 2, :Wo * 2].reshape(B, C, To, pt, Ho, 2, Wo, 2).permute(0, 1, 2, 4, 6, 3, 5, 7)
    w = w.reshape(B, C, To, Ho, Wo, pt * 4)
    top = w.topk(2, dim=-1).values
    return w.argmax(-1), top[..., 0] - top[..., 1], top[..., 0]


def decode_argmax(am, B, C, To, Ho, Wo, feat_layout):
    """arg-max bytes of one level -> (position index, dead flag), both (B,C,To,Ho,Wo)."""
    am = am.cpu()
    if feat_layout:          # last level: features in (C,T,H,W) order
        a = am.view(B, C, To, Ho, Wo)
    else:                    # channels-last slots [clip][C/8][t][h][w][8]
        a = am.view(B, C // 8, To, Ho, Wo, 8).permute(0, 1, 5, 2, 3, 4).reshape(B, C, To, Ho, Wo)
    a = a.to(torch.int64)
    return a & 7, (a & 0x80) != 0


def compare_decisions(x_btchw, params, am, tie_tol=2e-5):
    """fp64 oracle forward of clips x with `params` vs the HIP arg-max bytes `am` = (am0, am1, am2) of the same forward.
    -> per level {"windows", "mismatch", "not_near_tie", "worst_margin"}: a mismatch is a live window whose recorded
    position differs from the oracle's, or a window whose dead flag differs; it counts as a near-tie when the oracle's
    margin (top1 - top2, resp. |top1| for the dead flag) is below tie_tol x rms of the level's conv output."""
    p64 = [p.detach().double().cpu() for p in params[:6]]
    collect = []
    with torch.no_grad():
        R.feature_layers(x_btchw.detach().double().cpu().permute(0, 2, 1, 3, 4), p64, collect=collect)
    out = []
    for li, (_, pool) in enumerate(R.LAYER_SPECS):
        z = collect[3 * li]
        arg, margin, top1 = oracle_windows(z, pool[0])
        B, C, To, Ho, Wo = arg.shape
        pos, dead = decode_argmax(am[li], B, C, To, Ho, Wo, feat_layout=(li == 2))
        scale = float(z.pow(2).mean().sqrt())
        o_dead = top1 <= 0
        flag_diff = dead != o_dead
        pos_diff = (~dead) & (~o_dead) & (pos != arg)
        slack = torch.where(flag_diff, top1.abs(), margin) / scale
        mism = flag_diff | pos_diff
        far = mism & (slack > tie_tol)
        out.append({"windows": int(arg.numel()), "mismatch": int(mism.sum()), "not_near_tie": int(far.sum()),
                    "worst_margin": float(slack[mism].max()) if bool(mism.any()) else 0.0,
                    "mismatch_per_clip": [int(v) for v in mism.flatten(1).sum(1)],
                    "not_near_tie_per_clip": [int(v) for v in far.flatten(1).sum(1)]})
    return out


# ---- the oracle with IMPOSED pooling decisions (round 6) -----------------------------------------------------------------------
# Over MTT's ten unrolled student steps (distill_baseline.py:231-262) every gradient comparison is decided by a handful of pooling
# near-ties: one window routed the other way in an early step moves the median memory row by percents, and whether the HIP path,
# the fp32 oracle or neither has such a window is a matter of the seed (tools/parity_mtt10.py over seeds: HIP 50x closer to fp64
# than fp32 arithmetic, equal, or 8x further).  The arithmetic is therefore compared on the SAME piecewise-linear function: the
# oracle's forward takes the decisions the HIP forward recorded (its arg-max bytes) instead of its own max_pool3d, in fp64 and in
# fp32; and the decisions are compared separately -- every window where the fp64 values would have chosen otherwise must be a
# near-tie of those values.
def routes_from_argmax(am, x_shape, params):
    """HIP arg-max bytes (am0, am1, am2) of a forward of clips (B,T,3,H,W) -> per level (position index, dead flag), (B,C,To,Ho,Wo)."""
    import oracle.ref_cpu as R_
    B, T, _, H, W = x_shape
    routes = []
    for li, (cout, pool) in enumerate(R_.LAYER_SPECS):
        Tc, Hc, Wc = T, (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
        To, Ho, Wo = Tc // pool[0], Hc // 2, Wc // 2
        routes.append(decode_argmax(am[li], B, cout, To, Ho, Wo, feat_layout=(li == 2)))
        T, H, W = To, Ho, Wo
    return routes


def routed_feature_layers(x_bcthw, params, routes, stats=None, tie_tol=2e-5):
    """``oracle.ref_cpu.feature_layers`` with the ReLU + max-pool of every level replaced by the given routing: the selected window
    element, or zero where the window was recorded dead.  ``stats`` (a list) receives per level how many windows the values
    computed HERE would have routed otherwise, and how many of those are no near-tie (margin / rms above tie_tol)."""
    import torch.nn.functional as F_
    out = x_bcthw
    for li, (_, pool) in enumerate(R.LAYER_SPECS):
        z = F_.conv3d(out, params[2 * li], params[2 * li + 1], stride=R.CONV_STRIDE, padding=R.CONV_PAD)
        B, C, T, H, W = z.shape
        pt = pool[0]
        To, Ho, Wo = T // pt, H // 2, W // 2
        w = z[:, :, :To * pt, :Ho * 2, :Wo * 2].reshape(B, C, To, pt, Ho, 2, Wo, 2).permute(0, 1, 2, 4, 6, 3, 5, 7).reshape(B, C, To, Ho, Wo, pt * 4)
        pos, dead = routes[li]
        sel = w.gather(-1, pos.unsqueeze(-1)).squeeze(-1)
        out = torch.where(dead, torch.zeros_like(sel), sel)
        if stats is not None:
            with torch.no_grad():
                wd = w.detach()
                top = wd.topk(2, dim=-1).values
                arg, margin, top1 = wd.argmax(-1), top[..., 0] - top[..., 1], top[..., 0]
                scale = float(z.detach().pow(2).mean().sqrt())
                o_dead = top1 <= 0
                flag_diff = dead != o_dead
                pos_diff = (~dead) & (~o_dead) & (pos != arg)
                slack = torch.where(flag_diff, top1.abs(), margin) / scale
                mism = flag_diff | pos_diff
                stats.append({"level": li, "windows": int(arg.numel()), "mismatch": int(mism.sum()),
                              "not_near_tie": int((mism & (slack > tie_tol)).sum()),
                              "worst_margin": float(slack[mism].max()) if bool(mism.any()) else 0.0})
    return out


def routed_logits(x_btchw, params, routes, stats=None):
    import torch.nn.functional as F_
    feat = routed_feature_layers(x_btchw.permute(0, 2, 1, 3, 4), params, routes, stats)
    big = x_btchw.shape[-2] > 64
    feat = F_.avg_pool3d(feat, kernel_size=(2, 2, 2) if big else (2, 1, 1), stride=1)
    out = F_.conv3d(feat, params[6], params[7]).squeeze(3).squeeze(3)
    return out.max(dim=2).values      # (one pooled frame at 64x64x8: no decision; taller clips keep the oracle's own max over frames)


def mtt_step_routed(start, target, image_syn, label_syn, syn_lr, index_chunks, routes_per_step, dtype=torch.float64, stats=None):
    """``oracle.ref_cpu.mtt_step`` (distill_baseline.py:213-262) with the student forwards routed by ``routes_per_step`` (one list
    of per-level routes per unrolled step, for that step's clips in batch order).  -> (grand loss, d/d image_syn, d/d syn_lr)."""
    import torch.nn.functional as F_
    num_classes = start[6].shape[0]
    x = image_syn.detach().to(dtype).clone().requires_grad_(True)
    lr = torch.tensor(float(syn_lr), dtype=dtype, requires_grad=True)
    theta0 = R.flatten_params([p.detach().to(dtype) for p in start])
    tgt = R.flatten_params([p.detach().to(dtype) for p in target])
    theta = theta0.clone().requires_grad_(True)
    for s, idx in enumerate(index_chunks):
        st = None if stats is None else []
        logits = routed_logits(x[idx], R.unflatten_params(theta, 3, num_classes), routes_per_step[s], st)
        if stats is not None:
            stats.append(st)
        ce = F_.cross_entropy(logits, label_syn[idx])
        (g,) = torch.autograd.grad(ce, theta, create_graph=True)
        theta = theta - lr * g
    grand = ((theta - tgt) ** 2).sum() / ((theta0 - tgt) ** 2).sum()
    gx, glr = torch.autograd.grad(grand, [x, lr])
    return grand.detach(), gx, glr
