import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import ref_cpu as R
params = R.init_params(1)
g = torch.Generator().manual_seed(2)
real = torch.randn(64, 16, 3, 112, 112, generator=g); syn = torch.randn(1, 16, 3, 112, 112, generator=g)
print("cpu_count", os.cpu_count())
for th in (16, 32, 64, 128, 256):
    torch.set_num_threads(th)
    R.dm_loss_and_grad(params, [real[:8]], syn, 1)
    t0 = time.perf_counter(); R.dm_loss_and_grad(params, [real], syn, 1); dt = time.perf_counter() - t0
    print("threads %d: %.2f s per class term (112x112x16)" % (th, dt), flush=True)
