"""CPU simulation (oracle ops) of what an fp8-CORRECTED level-1 forward (VD_PREC_F16C8: fp16 main product + the two hi+lo correction
products on the block-scaled fp8 instruction -- what the real side's LAST level runs in DM) would do to gradient matching (config 4)
if the REAL batch's level-1 forward used it instead of fp16 hi+lo pairs: one class term at the configuration's geometry (64 real + 5
synthetic clips 112x112x16, 51-way head, 'ours'), the real forward's level-1 pre-activations replaced by

    exact + [ variant(a0, w1) - exact ].detach()        variant = c8 | f16 (single pass) | w16 (hi+lo activations x ONE rn16 weight
                                                         plane: two products) | a16 (ONE rn16 activation plane x hi+lo weights: two products)

so that the decisions (ReLU, pooling arg-max) and values downstream are the variant's while every derivative is the exact one --
the effect of the operand format alone.  Prints the relative change of gw_real per parameter and of the pixel gradient of
match_loss per synthetic clip.  The c8 numerics are those of csrc/aux_kernels.hip (pack_weights_c8_kernel: W_hi x s and W_lo x 2048 s
as e4m3, s = the power of two that brings max|W| into [128, 256)) and csrc/conv_mfma.hip (vd_c8_lo_byte: a_lo x 2^9 as e4m3;
vd_c8_hi_byte: a_hi / 4 as e4m3).  Lives with the tests' tools: it imports the oracle.

usage: python tests/sim_dc_c8_level1_tool.py [real_clips=64] [seed=404] [levels=1]     (levels: "1", "0", "01": which levels run the variant)"""
import math
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, '/root/repo')
from oracle import ref_cpu as R

torch.set_num_threads(8)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 404
levels = sys.argv[3] if len(sys.argv) > 3 else "1"
K, ipc, T, S = 51, 5, 16, 112
KINDS = ("c8", "f16", "w16", "a16")
DT = torch.float32        # base arithmetic (its own 1e-7 is common to every variant: the base is shared, only the delta differs)


def rn16(t):
    return t.half().to(t.dtype)


def e4m3(t):
    return t.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(t.dtype)


def conv(a, w):
    return F.conv3d(a, w, None, stride=R.CONV_STRIDE, padding=R.CONV_PAD)


def variant_preact(a, w, kind):
    """conv(a, w) in the operand format ``kind`` (no bias)."""
    a_hi, w_hi = rn16(a), rn16(w)
    if kind == "w16":                      # exact activations x rn16 weights: hi+lo activation pairs against ONE weight plane (two products)
        return conv(a, w_hi)
    if kind == "a16":                      # rn16 activations x exact weights: ONE activation plane against hi+lo weight pairs (two products)
        return conv(a_hi, w)
    main = conv(a_hi, w_hi)
    if kind == "f16":
        return main
    s = 2.0 ** (7 - math.floor(math.log2(float(w.abs().max()))))          # max|W| * s in [128, 256)
    a_lo8 = e4m3((a - a_hi) * 512.0) / 512.0
    a_hi8 = e4m3(a_hi.clamp(-1792.0, 1792.0) / 4.0) * 4.0
    w_hi8 = e4m3(w_hi * s) / s
    w_lo8 = e4m3((w - w_hi) * (s * 2048.0)) / (s * 2048.0)
    return main + conv(a_lo8, w_hi8) + conv(a_hi8, w_lo8)


def logits_with(x_btchw, params, kind):
    """convnet3d_logits with the listed levels' pre-activations carrying the variant's values (derivatives: exact)."""
    out = x_btchw.permute(0, 2, 1, 3, 4)
    for li, (_, pool) in enumerate(R.LAYER_SPECS):
        z = F.conv3d(out, params[2 * li], params[2 * li + 1], stride=R.CONV_STRIDE, padding=R.CONV_PAD)
        if kind is not None and str(li) in levels:
            with torch.no_grad():
                delta = variant_preact(out.detach(), params[2 * li].detach(), kind) - conv(out.detach(), params[2 * li].detach())
            z = z + delta
        out = F.max_pool3d(torch.relu(z), kernel_size=pool, stride=pool)
    feat = F.avg_pool3d(out, kernel_size=(2, 2, 2), stride=1)
    o = F.conv3d(feat, params[6], params[7]).squeeze(3).squeeze(3)
    return o.max(dim=2).values


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


g = torch.Generator().manual_seed(seed)
base = torch.randn(T, 3, S, S, generator=g)
real = (base[None] + 0.7 * torch.randn(B, T, 3, S, S, generator=g)).to(DT)
syn = (base[None] + 0.7 * torch.randn(ipc, T, 3, S, S, generator=g)).to(DT)
params = [q.to(DT).requires_grad_(True) for q in R.init_params(4040, 3, K)]
lab_r, lab_s = torch.full((B,), 7), torch.full((ipc,), 7)

t0 = time.time()
gw = {}
for kind in (None,) + KINDS:
    gw[kind] = [t.detach() for t in torch.autograd.grad(F.cross_entropy(logits_with(real, params, kind), lab_r), params)]
    print("real side %-5s done (%.0f s)" % (kind or "exact", time.time() - t0), flush=True)
names = ["w0", "b0", "w1", "b1", "w2", "b2", "wh", "bh"]
for kind in KINDS:
    print("gw_real, level(s) %s in %-3s vs exact: %s" % (levels, kind, "  ".join("%s %.1e" % (n, rel(a, b)) for n, a, b in zip(names, gw[kind], gw[None]))))

xs = syn.clone().requires_grad_(True)
gw_syn = torch.autograd.grad(F.cross_entropy(R.convnet3d_logits(xs, params), lab_s), params, create_graph=True)
gx = {}
for kind in (None,) + KINDS:
    loss = R.match_loss(gw_syn, gw[kind], "ours")
    (gx[kind],) = torch.autograd.grad(loss, xs, retain_graph=True)
    print("match_loss with gw_real %-5s: %.6f" % (kind or "exact", float(loss)), flush=True)
for kind in KINDS:
    per = [rel(gx[kind][i], gx[None][i]) for i in range(ipc)]
    print("pixel gradient of match_loss, real level(s) %s in %-3s vs exact: per synthetic clip %s, all %.2e" % (
        levels, kind, ["%.1e" % v for v in per], rel(gx[kind], gx[None])))
print("(%.0f s)" % (time.time() - t0))
