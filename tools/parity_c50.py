"""The late-regime parity run of tests/test_gpu_parity_late.py at the BENCHMARK's class count: C = 50 classes x (64 real + 1 syn) clips
112x112x16, shipped mode, one step (the suite runs 2 and 4 classes: the fp64 oracle takes ~8 s per class term).  Every class is one
(step, class) entry: clean entries must be within 1e-3 of the fp64 oracle's pixel gradient, the loss within 1e-3.
   python tools/parity_c50.py [seed] [C] [small]     -> gpurun_out/r06_parity_c50.json + a summary on stdout
("small": config 1's shape, 64x64x8, with the suite's settings for it)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import ref_cpu as R
from tests import test_gpu_parity_late as P

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 12
C = int(sys.argv[2]) if len(sys.argv) > 2 else 50
small = len(sys.argv) > 3 and sys.argv[3] == "small"


class _Converted:
    """The real batches converted to the oracle's dtype one class at a time (3200 clips in fp64 at once are 62 GB)."""
    def __init__(self, reals, dtype):
        self.reals, self.dtype = reals, dtype

    def __iter__(self):
        for r in self.reals:
            yield r.to(self.dtype)


def _oracle(params, reals, syn, dtype):
    old = torch.get_num_threads()
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    try:
        t0 = time.perf_counter()
        loss, grad = R.dm_loss_and_grad([p.to(dtype) for p in params], _Converted(reals, dtype), syn.to(dtype), ipc=1)
        return float(loss), grad, time.perf_counter() - t0
    finally:
        torch.set_num_threads(old)


P._oracle = _oracle
t0 = time.time()
if small:
    rec = P.late_regime_run((8, 64, 64), C=C, NP=80, B=64, steps=2, lr=50.0, seed=seed)
else:
    rec = P.late_regime_run((16, 112, 112), C=C, NP=66, B=64, steps=1, lr=20.0, seed=seed)
P._report("late regime %s, C=%d, seed %d" % ("64x64x8" if small else "112x112x16", C, seed), rec, ("shipped",))
s, clean = rec["shipped"]["summary"], rec["shipped"]["summary_clean"]
per = np.asarray(rec["shipped"]["grad_vs_fp64_per_class"]).reshape(-1)
upper = np.asarray(rec["decisions"]["mismatch_per_class"]).reshape(-1)
far = int(np.asarray(rec["decisions"]["not_near_tie_per_class"]).sum())
ok = (s["loss_vs_fp64_max"] < 1e-3 and s["loss_vs_fp32_max"] < 1e-3 and far == 0 and all(per[c] < P.GRAD_BAR for c in range(per.size) if upper[c] == 0)
      and all(per[c] < 5e-2 for c in range(per.size)))
print("clean entries %d of %d: median %.3e max %.3e; all entries median %.3e; loss vs fp64 %.2e; decisions outside near-ties %d; %s (%.0f s)" % (
    clean["entries"], clean["of"], clean["grad_vs_fp64_median"] or float("nan"), clean["grad_vs_fp64_max"] or float("nan"),
    s["grad_vs_fp64_median"], s["loss_vs_fp64_max"], far, "WITHIN THE BARS" if ok else "OUT OF TOLERANCE", time.time() - t0))
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r06_parity_c50.json")
os.makedirs(os.path.dirname(out), exist_ok=True)
data = json.load(open(out)) if os.path.exists(out) else {}
data["seed%d_C%d%s" % (seed, C, "_64x64x8" if small else "")] = rec
json.dump(data, open(out, "w"), indent=1, sort_keys=True)
sys.exit(0 if ok else 1)
