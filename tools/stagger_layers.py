"""Two halves of a real batch on two streams, the second half one layer behind the first (its first-layer program runs
under the other half's second-layer program) vs one stream.  usage: python tools/stagger_layers.py [clips per half]"""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from video_distillation_amd import engine, plan, hip

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1600
geo = plan.NetGeometry(16, 112, 112)
params = [torch.randn(s, device="cuda") * 0.02 for s in [(64, 3, 3, 7, 7), (64,), (128, 64, 3, 7, 7), (128,), (128, 128, 3, 7, 7), (128,)]]
L = hip.lib()


class Half:
    def __init__(self):
        self.eng = engine.EmbedEngine(geo, prec="f16", chunk=n)
        self.eng.set_weights(params)
        e = self.eng
        self.x = torch.randn(n, 16, 3, 112, 112, device="cuda")
        self.n0 = n * 16 * 3 * 112 * 15
        self.s0 = torch.empty((1, self.n0, 8), dtype=torch.int16, device="cuda")
        L.vd_pix2rows(hip.ptr(self.x), None, ctypes.c_int64(n), 16, 112, 112, hip.ptr(self.s0[0]), None, e.prec, hip.stream_ptr(e.device))
        self.n1 = n * int(np.prod(e.fwd[0].plan.out_shape[:-1])); self.a1 = torch.empty((1, self.n1, 8), dtype=torch.int16, device="cuda")
        self.n2 = n * int(np.prod(e.fwd[1].plan.out_shape[:-1])); self.a2 = torch.empty((1, self.n2, 8), dtype=torch.int16, device="cuda")
        self.f = torch.empty(n, 2048, device="cuda")

    def layer(self, i):
        e, w = self.eng, self.eng._weights
        if i == 0: e.fwd[0].run(self.s0, self.n0, w[1], self.a1.data_ptr(), self.n1, None, n)
        if i == 1: e.fwd[1].run(self.a1, self.n1, w[3], self.a2.data_ptr(), self.n2, None, n)
        if i == 2: e.fwd[2].run(self.a2, self.n2, w[5], self.f.data_ptr(), 0, None, n)


A, B = Half(), Half()
torch.cuda.synchronize()
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()


def sequential():
    for h in (A, B):
        for i in range(3):
            h.layer(i)


def staggered():
    cur = torch.cuda.current_stream()
    s0.wait_stream(cur); s1.wait_stream(cur)
    with torch.cuda.stream(s0):
        A.layer(0)
        ev = torch.cuda.Event(); ev.record(s0)
        A.layer(1); A.layer(2)
    with torch.cuda.stream(s1):
        s1.wait_event(ev)
        B.layer(0); B.layer(1); B.layer(2)
    cur.wait_stream(s0); cur.wait_stream(s1)


def simultaneous():
    cur = torch.cuda.current_stream()
    s0.wait_stream(cur); s1.wait_stream(cur)
    for h, s in ((A, s0), (B, s1)):
        with torch.cuda.stream(s):
            for i in range(3):
                h.layer(i)
    cur.wait_stream(s0); cur.wait_stream(s1)


def timeit(fn, reps=5):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for rnd in range(2):
    print("queues %s | 2 x %d clips: sequential %.2f ms, second half one layer behind %.2f ms, both at once %.2f ms"
          % (os.environ.get("GPU_MAX_HW_QUEUES"), n, timeit(sequential), timeit(staggered), timeit(simultaneous)))
