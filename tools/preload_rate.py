"""Throughput of dataset.preload on this box: the JPEG fixture's training items, repeated, decoded on N host threads and
normalised on the device.  usage: python tools/preload_rate.py [repeats] [workers ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, random, torch
from video_distillation_amd import dataset as D
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "frames", "UCF101")
rep = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ds = D.UCF101(root, "train")
idx = list(range(len(ds))) * rep
D.preload(ds, "cuda:0", indices=idx[:6], workers=4)            # warm-up (library load, first allocations)
for workers in [int(v) for v in sys.argv[2:]] or [1, 8, 32, 64]:
    np.random.seed(0); random.seed(0)
    t0 = time.perf_counter()
    clips, labels = D.preload(ds, "cuda:0", indices=idx, workers=workers, chunk=64)
    dt = time.perf_counter() - t0
    print("%3d decode threads: %d clips (16 x 112 x 112) in %.2f s = %.0f clips/s = %.2f GB/s of fp32 clips; a 4662-clip split: %.1f s"
          % (workers, len(idx), dt, len(idx) / dt, clips.numel() * 4 / dt / 1e9, 4662 / (len(idx) / dt)))
