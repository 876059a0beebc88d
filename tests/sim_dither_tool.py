"""CPU simulation (oracle ops) of DITHERED weight rounding for the real side: the real clips of a class are dealt to G groups
and group g multiplies by weights rounded up or down to neighbouring f16 values such that the mean over the groups equals
the fp32 weight to 1/(2G) ulp.  Compares |mean over clips of (f(x; W_g(clip)) - f(x; W))| / |f| with plain rn16(W)
(|mean d|) and with the value-pass remedy (|mean d - d_syn|).  Lives with the tests' tools: it imports the oracle."""
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from oracle import ref_cpu as R
torch.set_num_threads(8)


def neighbours(w):
    """(lo, hi): the two f16 values bracketing w (lo <= w <= hi; equal when w is representable)."""
    q = w.half()
    qf = q.float()
    up = torch.nextafter(q, torch.full_like(q, float("inf"))).float()
    dn = torch.nextafter(q, torch.full_like(q, float("-inf"))).float()
    lo = torch.where(qf <= w, qf, dn)
    hi = torch.where(qf >= w, qf, up)
    return lo, hi


def bitrev(g, G):
    b = G.bit_length() - 1
    return int(format(g, "0%db" % b)[::-1], 2) if b else 0


def dither(w, g, G):
    lo, hi = neighbours(w)
    lam = torch.where(hi > lo, (w - lo) / (hi - lo), torch.zeros_like(w))            # fraction of the way up
    n = w.numel()
    rot = (torch.arange(n, dtype=torch.int64) * 2654435761 % 4294967296 >> 7) % G     # per-element rotation of the group order
    slot = (bitrev(g, G) + rot.view(w.shape)) % G
    t = (slot.float() + 0.5) / G
    return torch.where(t < lam, hi, lo)


def main():
    for name in ("similar", "independent"):
        g = torch.Generator().manual_seed(5)
        nreal = 32
        if name == "similar":
            base = torch.randn(8, 3, 64, 64, generator=g)
            x = base + 0.1 * torch.randn(nreal + 1, 8, 3, 64, 64, generator=g)
        else:
            x = torch.randn(nreal + 1, 8, 3, 64, 64, generator=g)
        params = R.init_params(1234)
        f0 = R.convnet3d_embed(x, params)
        fn = float(f0[0].norm())
        p_rn = [p.half().float() if p.dim() == 5 else p for p in params[:6]]
        d_rn = R.convnet3d_embed(x, p_rn) - f0
        print("%-12s rn16:      |mean d|/|f| %.2e   value pass |mean d - d_syn|/|f| %.2e" % (
            name, float(d_rn[:nreal].mean(0).norm()) / fn, float((d_rn[:nreal].mean(0) - d_rn[nreal]).norm()) / fn))
        for G in (2, 4, 8, 16, 32):
            d = torch.zeros(nreal, f0.shape[1])
            for gi in range(G):
                pg = [dither(p, gi, G) if p.dim() == 5 else p for p in params[:6]]
                sel = torch.arange(gi, nreal, G)
                d[sel] = R.convnet3d_embed(x[sel], pg) - f0[sel]
            print("%-12s dither G=%-2d |mean d|/|f| %.2e" % (name, G, float(d.mean(0).norm()) / fn))


if __name__ == "__main__":
    main()
