"""Single-GPU proxy of ONE rank's step at N GPUs (no collectives are issued: torch.distributed is not initialised, the
trainer is simply constructed as rank r of N): real clips 50 x 64/N (batch mode), the rank's class block (class mode) or its block
of 50 // N whole classes + 1/N of the left-over classes' batches (hybrid); synthetic clips of the classes the rank owns.  Prints ms/step -> what 1/T would be at N GPUs if the exchange were free.
usage: python tools/rank_proxy.py [N ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_distillation_amd import distill, plan
from video_distillation_amd.networks import _batch_hint

dev = torch.device("cuda:0")
geo = plan.NetGeometry(16, 112, 112)
C = 50
pool_all = distill.RealPool.synthetic(C, list(range(C)), 70, geo, dev)
base = None
for arg in (sys.argv[1:] or ["1", "2", "4", "8"]):
    N = int(arg)
    for shard in (("class",) if N == 1 else ("batch", "class") + (("hybrid",) if C % N else ())):
        if os.environ.get("VD_PROXY_SHARDS") and shard not in os.environ["VD_PROXY_SHARDS"].split(","):
            continue
        rank = 0
        lo, hi = distill.class_range(C, rank, N)
        nown = (C // N + 1) if shard == "hybrid" else hi - lo
        be = distill.HipBackend(geo, dev, chunk=3200, syn_batch_hint=_batch_hint(nown))
        tr = distill.DMTrainer(be, pool_all, C, 1, 64, lr_img=1.0, rank=rank, world=N, shard=shard)
        for it in range(3):
            tr.step(it, overlap=True)
        tr.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for it in range(3, 3 + n):
            tr.step(it, overlap=True)
        tr.sync(); torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        if N == 1 or base is None:
            base = ms
        print("N=%d %-5s rank 0: %6.2f ms/step  -> %5.1f steps/s, %.2fx of N=1" % (N, shard, ms, 1e3 / ms, base / ms))
