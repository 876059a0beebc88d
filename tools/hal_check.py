"""Hallucinator backward (vd_hallucinator_bwd) against torch autograd of the same Conv3d on the GPU (a checker, not the product
path) + its time, for the two forms: VD_HAL_FUSED=1 (default: one fused kernel) / 0 (data + parameter kernels).
usage: python tools/hal_check.py [clips] [T] [H] [W]"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from video_distillation_amd import hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
T, H, W = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (16, 112, 112)
g = torch.Generator(device="cuda").manual_seed(5)
ns, nd = max(2, n // 2), max(2, n - 3)
stat = torch.randn(ns, 3, H, W, device="cuda", generator=g)
dyn = torch.randn(nd, T, 1, H, W, device="cuda", generator=g)
sidx = torch.randint(0, ns, (n,), device="cuda", generator=g)
didx = torch.randint(0, nd, (n,), device="cuda", generator=g)            # (memories shared by several clips: the scatter-add matters)
w = (torch.rand(3, 4, 3, 3, 3, device="cuda", generator=g) - 0.5) * 0.2
b = (torch.rand(3, device="cuda", generator=g) - 0.5) * 0.2
go = torch.randn(n, T, 3, H, W, device="cuda", generator=g)
# reference: out[i] = conv3d(cat(static[sidx[i]] repeated over T, dynamic[didx[i]]), w, b)  (utils.py:1186-1197)
st_r, dy_r, w_r, b_r = (t.clone().requires_grad_(True) for t in (stat, dyn, w, b))
x = torch.cat([st_r[sidx].unsqueeze(2).expand(-1, -1, T, -1, -1), dy_r[didx].permute(0, 2, 1, 3, 4)], 1)      # (n, 4, T, H, W)
out = F.conv3d(x, w_r, b_r, padding=1)                                                                           # (n, 3, T, H, W)
out.backward(go.permute(0, 2, 1, 3, 4))
L = hip.lib()
def run():
    g_dyn = torch.zeros_like(dyn); g_stat = torch.zeros_like(stat); g_w = torch.zeros_like(w); g_b = torch.zeros_like(b)
    hip.check(L.vd_hallucinator_bwd(hip.ptr(go), hip.ptr(stat), hip.ptr(dyn), hip.ptr(sidx), hip.ptr(didx), hip.ptr(w), n, T, H, W,
                                    hip.ptr(g_dyn), hip.ptr(g_stat), hip.ptr(g_w), hip.ptr(g_b), hip.stream_ptr("cuda")), "vd_hallucinator_bwd")
    return g_dyn, g_stat, g_w, g_b
got = run()
torch.cuda.synchronize()
rel = lambda a, r: float((a - r).norm() / r.norm())
print("VD_HAL_FUSED=%s  %d clips %dx%dx%d: rel-L2 vs autograd  g_dyn %.2e  g_stat %.2e  g_w %.2e  g_b %.2e" % (
    os.environ.get("VD_HAL_FUSED", "1"), n, H, W, T, rel(got[0], dy_r.grad), rel(got[1], st_r.grad), rel(got[2], w_r.grad), rel(got[3], b_r.grad)))
ts = []
for _ in range(8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g_dyn = torch.zeros_like(dyn); g_stat = torch.zeros_like(stat); g_w = torch.zeros_like(w); g_b = torch.zeros_like(b)
    e0.record()
    L.vd_hallucinator_bwd(hip.ptr(go), hip.ptr(stat), hip.ptr(dyn), hip.ptr(sidx), hip.ptr(didx), hip.ptr(w), n, T, H, W,
                          hip.ptr(g_dyn), hip.ptr(g_stat), hip.ptr(g_w), hip.ptr(g_b), hip.stream_ptr("cuda"))
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ms = sorted(ts)[len(ts) // 2]
algo = (go.numel() + dyn.numel() * 2 + stat.numel() * 2) * 4        # read g_out, dyn, stat once; write g_dyn, g_stat
print("   median %.3f ms  -> %.2f TB/s of %.0f MB algorithmic bytes (%.1f %% of 8 TB/s)" % (ms, algo / ms / 1e9, algo / 1e6, algo / ms / 1e9 / 8 * 100))
