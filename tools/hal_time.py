import ctypes, sys, os
sys.path.insert(0, os.getcwd())
import torch
from video_distillation_amd import hip
dev = torch.device("cuda:0")
n, T, H, W = 50, 16, 112, 112
static = torch.randn(n, 3, H, W, device=dev); dynamic = torch.randn(n, T, 1, H, W, device=dev)
up = torch.randn(n, T, 3, H, W, device=dev); w = torch.randn(324, device=dev) * 0.1; b = torch.randn(3, device=dev)
out = torch.empty(n, T, 3, H, W, device=dev)
g_dyn = torch.zeros_like(dynamic); g_stat = torch.zeros_like(static); g_w = torch.zeros(324, device=dev); g_b = torch.zeros(3, device=dev)
L = hip.lib(); st = hip.stream_ptr(dev)
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b_.record(); torch.cuda.synchronize()
    return a.elapsed_time(b_) / reps
print("fwd kernel      %.3f ms" % timeit(lambda: L.vd_hallucinator_fwd(hip.ptr(static), hip.ptr(dynamic), None, None, hip.ptr(w), hip.ptr(b), n, T, H, W, hip.ptr(out), st)))
print("bwd data only   %.3f ms" % timeit(lambda: L.vd_hallucinator_bwd(hip.ptr(up), hip.ptr(static), hip.ptr(dynamic), None, None, hip.ptr(w), n, T, H, W, hip.ptr(g_dyn), None, None, None, st)))
print("bwd data+stat   %.3f ms" % timeit(lambda: L.vd_hallucinator_bwd(hip.ptr(up), hip.ptr(static), hip.ptr(dynamic), None, None, hip.ptr(w), n, T, H, W, hip.ptr(g_dyn), hip.ptr(g_stat), None, None, st)))
print("bwd data+param  %.3f ms" % timeit(lambda: L.vd_hallucinator_bwd(hip.ptr(up), hip.ptr(static), hip.ptr(dynamic), None, None, hip.ptr(w), n, T, H, W, hip.ptr(g_dyn), None, hip.ptr(g_w), hip.ptr(g_b), st)))
print("zeros_like dyn  %.3f ms" % timeit(lambda: torch.zeros_like(dynamic)))
