"""CPU simulation (oracle ops) of what rounding the conv WEIGHTS to f16 does to embed() features: per-clip perturbation,
its mean over a batch (the systematic part), and what is left after subtracting the synthetic clip's own perturbation
(the value pass); plain rounding vs error-diffusion rounding along the taps / along K.  Lives with the tests' tools:
it imports the oracle.  Output of the run recorded in DESIGN.md section 2."""
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from oracle import ref_cpu as R
torch.set_num_threads(8)
torch.manual_seed(0)

def rn16(w): return w.half().float()

def diffuse(w):
    # error-diffusion rounding along the taps of each (cout, cin) pair: carry the residual to the next tap
    o, i = w.shape[:2]
    flat = w.reshape(o * i, -1).double()
    out = torch.empty_like(flat)
    carry = torch.zeros(o * i, dtype=torch.float64)
    for t in range(flat.shape[1]):
        v = flat[:, t] + carry
        q = v.float().half().double()
        carry = v - q
        out[:, t] = q
    return out.float().reshape(w.shape)

def diffuse_all(w):
    # diffusion along the whole K = (cin, taps) of each output channel
    o = w.shape[0]
    flat = w.reshape(o, -1).double()
    out = torch.empty_like(flat)
    carry = torch.zeros(o, dtype=torch.float64)
    for t in range(flat.shape[1]):
        v = flat[:, t] + carry
        q = v.float().half().double()
        carry = v - q
        out[:, t] = q
    return out.float().reshape(w.shape)

def feats(x, params):
    return R.convnet3d_embed(x, params)

for name in ("similar", "independent"):
    g = torch.Generator().manual_seed(5)
    if name == "similar":
        base = torch.randn(8, 3, 64, 64, generator=g)
        x = base + 0.1 * torch.randn(33, 8, 3, 64, 64, generator=g)
    else:
        x = torch.randn(33, 8, 3, 64, 64, generator=g)
    params = R.init_params(1234)
    f0 = feats(x, params)
    fn = float(f0[0].norm())
    for mode, fn_round in (("rn16", rn16), ("diffuse-taps", diffuse), ("diffuse-K", diffuse_all)):
        p2 = [fn_round(p) if p.dim() == 5 else p for p in params[:6]]
        f1 = feats(x, p2)
        d = f1 - f0
        real = d[:32]; syn = d[32]
        print("%-12s %-13s per-clip |d|/|f| %.2e   |mean32 d|/|f| %.2e   |mean32 d - d_syn|/|f| %.2e" % (
            name, mode, float(d.norm(dim=1).mean()) / fn, float(real.mean(0).norm()) / fn, float((real.mean(0) - syn).norm()) / fn))
