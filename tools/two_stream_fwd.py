"""Does splitting the real-clip forward over two HIP streams (so that one half's first layer overlaps the
other half's second layer) beat one stream?  usage: python tools/two_stream_fwd.py [clips]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_distillation_amd import distill, engine, plan
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3200
geo = plan.NetGeometry(16, 112, 112)
pool = torch.randn(256, 16, 3, 112, 112, device="cuda")
idx = torch.randint(0, 256, (n,), device="cuda")
w = distill.fresh_network_weights(1, "cuda:0")


def make(chunk):
    e = engine.EmbedEngine(geo, prec="f16", chunk=chunk)
    e.set_weights(w)
    return e


def timeit(fn, reps=5):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


e1 = make(n)
rows = e1.pool_rows(pool)
print("one stream, one launch per layer: %.2f ms" % timeit(lambda: e1.forward(pool, index=idx, rows=rows)))
for parts in (2, 4):
    engs = [make(n // parts) for _ in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    per = n // parts

    def run():
        cur = torch.cuda.current_stream()
        for k, (e, s) in enumerate(zip(engs, streams)):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                e.forward(pool, index=idx[k * per:(k + 1) * per], rows=rows)
        for s in streams:
            cur.wait_stream(s)
    print("%d streams x %d clips: %.2f ms" % (parts, per, timeit(run)))
    e2 = make(per)
    print("   (one stream, %d sequential chunks: %.2f ms)" % (parts, timeit(lambda: [e2.forward(pool, index=idx[k * per:(k + 1) * per], rows=rows) for k in range(parts)])))
