"""Measured dense f16 MFMA rate of this device (vd_mfma_peak), both instruction shapes, after `--seconds` of
back-to-back launches on pseudo-random operands (the clock settles under load)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_distillation_amd import hip

ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=3.0); a = ap.parse_args()
dev = torch.device("cuda:0")
blocks, iters = 256 * 8, 4000
out = torch.empty(blocks * 256, dtype=torch.float32, device=dev)
for shape, name in ((0, "v_mfma_f32_32x32x16_f16"), (1, "v_mfma_f32_16x16x32_f16"), (0, "v_mfma_f32_32x32x16_f16"), (1, "v_mfma_f32_16x16x32_f16")):
    t0 = time.perf_counter(); times = []
    while time.perf_counter() - t0 < a.seconds:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            hip.check(hip.lib().vd_mfma_peak(blocks, iters, shape, hip.ptr(out), hip.stream_ptr(dev)), "vd_mfma_peak")
        e1.record(); torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) / 10)
    tf = blocks * 4 * iters * 8 * 32768.0 / (min(times[-5:]) * 1e-3) / 1e12
    tf0 = blocks * 4 * iters * 8 * 32768.0 / (times[0] * 1e-3) / 1e12
    print("%s: first launches %.0f TFLOP/s, after %.1f s %.0f TFLOP/s (%.1f %% of 2500)" % (name, tf0, a.seconds, tf, tf / 25.0))
