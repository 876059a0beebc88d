#!/usr/bin/env python3
"""HBM-bound side kernels of the path: achieved GB/s of algorithmic bytes vs the
8 TB/s HBM3E peak -- hallucinator fwd/bwd, match_loss fwd/bwd, DM loss, pixel SGD, pix2rows.
Prints one JSON object."""
import ctypes, json, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_distillation_amd import hip, utils, distill, plan


def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


def main():
    dev = torch.device("cuda:0")
    out = {}
    PEAK = 8000.0
    # hallucinator: 50 clips 112x112x16 (config 3: C=50, vpc=1)
    n, T, H, W = 50, 16, 112, 112
    hal = utils.Conv3DNet().to(dev)
    static = torch.randn(n, 3, H, W, device=dev)
    dynamic = torch.randn(n, T, 1, H, W, device=dev, requires_grad=True)
    up = torch.randn(n, T, 3, H, W, device=dev)
    t = timeit(lambda: hal(static, dynamic))
    byt = n * (3 * H * W + T * H * W + 3 * T * H * W) * 4
    out["hallucinator_fwd"] = {"ms": t * 1e3, "GBps": byt / t / 1e9, "frac_hbm": byt / t / 1e9 / PEAK, "bytes": byt}

    def bwd():
        o = hal(static, dynamic)
        o.backward(up)
        dynamic.grad = None; hal.zero_grad()
    tb = timeit(bwd, 10) - t
    byt_b = n * (3 * T * H * W * 2 + T * H * W * 2 + 3 * H * W) * 4
    out["hallucinator_bwd"] = {"ms": tb * 1e3, "GBps": byt_b / tb / 1e9, "frac_hbm": byt_b / tb / 1e9 / PEAK, "bytes": byt_b}
    # match_loss on a full ConvNet3D gradient pair (3 647 666 elements each)
    shapes = [(64, 3, 3, 7, 7), (64,), (128, 64, 3, 7, 7), (128,), (128, 128, 3, 7, 7), (128,), (50, 128, 1, 1, 1), (50,)]
    gr = [torch.randn(s, device=dev) for s in shapes]
    gs = [torch.randn(s, device=dev, requires_grad=True) for s in shapes]
    nel = sum(g.numel() for g in gr)
    for metric in ("ours", "mse", "cos"):
        args = types.SimpleNamespace(device=dev, dis_metric=metric)
        tf = timeit(lambda: utils.match_loss(gs, gr, args))

        def fb():
            v = utils.match_loss(gs, gr, args)
            v.backward()
            for g in gs:
                g.grad = None
        tfb = timeit(fb, 10)
        out["match_loss_%s_fwd" % metric] = {"ms": tf * 1e3, "GBps": 2 * nel * 4 / tf / 1e9, "frac_hbm": 2 * nel * 4 / tf / 1e9 / PEAK}
        out["match_loss_%s_fwd_bwd" % metric] = {"ms": tfb * 1e3, "GBps": 5 * nel * 4 / tfb / 1e9, "frac_hbm": 5 * nel * 4 / tfb / 1e9 / PEAK}
    # DM loss, SGD, pix2rows
    be = distill.HipBackend(plan.NetGeometry(8, 64, 64), dev)
    fr = torch.randn(50 * 64, 2048, device=dev); fs = torch.randn(50, 2048, device=dev)
    be.num_feat = 2048
    t = timeit(lambda: be.dm_loss(fr, fs, 50))
    out["dm_loss"] = {"ms": t * 1e3, "GBps": (fr.numel() + 2 * fs.numel()) * 4 / t / 1e9, "frac_hbm": (fr.numel() + 2 * fs.numel()) * 4 / t / 1e9 / PEAK}
    x = torch.randn(50, 16, 3, 112, 112, device=dev); buf = torch.zeros_like(x); g = torch.randn_like(x)
    t = timeit(lambda: be.sgd(x, buf, g, 0.1, 0.5, False))
    out["sgd_momentum"] = {"ms": t * 1e3, "GBps": x.numel() * 20 / t / 1e9, "frac_hbm": x.numel() * 20 / t / 1e9 / PEAK}
    xx = torch.randn(512, 16, 3, 112, 112, device=dev)
    rows = torch.empty(512 * 48 * 112 * 15, 8, dtype=torch.int16, device=dev)
    L = hip.lib()
    t = timeit(lambda: L.vd_pix2rows(hip.ptr(xx), None, ctypes.c_int64(512), 16, 112, 112, hip.ptr(rows), None, 1, hip.stream_ptr(dev)))
    byt = xx.numel() * 4 + rows.numel() * 2
    out["pix2rows_f16"] = {"ms": t * 1e3, "GBps": byt / t / 1e9, "frac_hbm": byt / t / 1e9 / PEAK}
    # decoded frames -> clips (dataset preload): 256 clips x 16 frames 112x112, 3 B read + 12 B written per pixel
    from video_distillation_amd import dataset as D
    del xx, rows
    u8 = torch.randint(0, 256, (256 * 16, 112, 112, 3), dtype=torch.uint8, device=dev)
    clips = torch.empty((256 * 16, 3, 112, 112), device=dev)
    t = timeit(lambda: D.frames_normalize(u8, clips, D.IMAGENET_MEAN, D.IMAGENET_STD))
    byt = u8.numel() * 5
    out["frames_normalize"] = {"ms": t * 1e3, "GBps": byt / t / 1e9, "frac_hbm": byt / t / 1e9 / PEAK, "bytes": byt}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
