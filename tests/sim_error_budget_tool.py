"""CPU simulation (oracle ops, fp64) of WHERE the class-mean feature error of the single-pass f16 real side comes from:
64 real clips of a class, weights dithered over G = 8 clip groups exactly as vd_pack_weights_dither does, operands rounded to
f16 at one place at a time -- pixels, the pooled outputs of layers 0 and 1, the weights of layers 0, 1 and 2 -- and all of them
together.  Prints |mean_j (f_variant(x_j) - f(x_j))| / |mean_j f(x_j)| per source; the squares of independent sources add.
Used for DESIGN section 2 (the error budget of the shipped mixed mode).  Lives with the tests' tools: it imports the oracle.

usage: python tests/sim_error_budget_tool.py [similar|independent] [nclips]"""
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, '/root/repo')
from oracle import ref_cpu as R
from tests.sim_dither_tool import dither        # noqa: E402  (the dither rule of vd_pack_weights_dither, restated there)

torch.set_num_threads(8)


def rn16(t):
    return t.float().half().double()


def embed(x, params, sets, G, round_pix=False, round_act=(False, False), wsrc=None):
    """fp64 ConvNet3D.embed of clips x (B,T,3,H,W); clip j multiplies by weight set j % G of `sets[layer]` (None: exact)."""
    B = x.shape[0]
    out = torch.empty(B, 0)
    feats = []
    for gi in range(G):
        sel = torch.arange(gi, B, G)
        a = x[sel].permute(0, 2, 1, 3, 4).double()
        if round_pix:
            a = rn16(a)
        for li, (_, pool) in enumerate(R.LAYER_SPECS):
            w = params[2 * li].double() if sets[li] is None else sets[li][gi]
            a = F.conv3d(a, w, params[2 * li + 1].double(), stride=R.CONV_STRIDE, padding=R.CONV_PAD)
            a = F.max_pool3d(torch.relu(a), kernel_size=pool, stride=pool)
            if li < 2 and round_act[li]:
                a = rn16(a)
        feats.append((sel, a.reshape(a.shape[0], -1)))
    out = torch.empty(B, feats[0][1].shape[1], dtype=torch.float64)
    for sel, f in feats:
        out[sel] = f
    return out


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "similar"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    G = 8
    g = torch.Generator().manual_seed(5)
    if kind == "similar":
        x = torch.randn(8, 3, 64, 64, generator=g) + 0.1 * torch.randn(n, 8, 3, 64, 64, generator=g)
    else:
        x = torch.randn(n, 8, 3, 64, 64, generator=g)
    params = R.init_params(1234)
    dsets = [[dither(params[2 * li], gi, G).double() for gi in range(G)] for li in range(3)]
    none = [None, None, None]
    f0 = embed(x, params, none, G)
    m0 = f0.mean(0)
    fn = float(m0.norm())

    def report(name, f):
        d = f - f0
        print("%-34s per-clip %.2e   class mean %.2e" % (name, float((d.norm(dim=1) / f0.norm(dim=1)).mean()), float(d.mean(0).norm()) / fn),
              flush=True)
        return float(d.mean(0).norm()) / fn

    tot = 0.0
    tot += report("pixels rn16", embed(x, params, none, G, round_pix=True)) ** 2
    tot += report("act0 rn16", embed(x, params, none, G, round_act=(True, False))) ** 2
    tot += report("act1 rn16", embed(x, params, none, G, round_act=(False, True))) ** 2
    for li in range(3):
        s = list(none); s[li] = dsets[li]
        tot += report("W%d dithered (G=8)" % li, embed(x, params, s, G)) ** 2
    print("root sum of squares of the six sources: %.2e" % tot ** 0.5)
    report("all six (the shipped real side)", embed(x, params, dsets, G, round_pix=True, round_act=(True, True)))
    report("all but act1 + W2 (layer 2 in hi+lo)", embed(x, params, [dsets[0], dsets[1], None], G, round_pix=True, round_act=(True, False)))
    report("all but pixels + W0 (layer 0 in hi+lo)", embed(x, params, [None, dsets[1], dsets[2]], G, round_act=(True, True)))
    plain = [[params[2 * li].half().double()] * G for li in range(3)]
    report("plain rn16(W), all roundings", embed(x, params, plain, G, round_pix=True, round_act=(True, True)))


if __name__ == "__main__":
    main()
