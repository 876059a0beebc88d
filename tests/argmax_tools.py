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
