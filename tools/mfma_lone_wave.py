import sys, time
sys.path.insert(0, "/root/repo")
import torch
from video_distillation_amd import hip
dev = torch.device("cuda:0")
for blocks, iters in ((256, 32000), (512, 16000), (2048, 4000), (256, 32000)):
    out = torch.empty(blocks * 256, dtype=torch.float32, device=dev)
    ts = []
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 2.0:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            hip.check(hip.lib().vd_mfma_peak(blocks, iters, 0, hip.ptr(out), hip.stream_ptr(dev)), "peak")
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5)
    tf = blocks * 4 * iters * 8 * 32768.0 / (min(ts[-3:]) * 1e-3) / 1e12
    print("blocks %d (%.1f waves/SIMD): %.0f TFLOP/s" % (blocks, blocks * 4 / 1024.0, tf))
