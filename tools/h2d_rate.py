"""Host->device copy rate of this box (pinned fp32, 154 MB = one class batch of 64 clips 112x112x16) and what the reference's
per-step get_images upload (distill_baseline.py:84-90: C x 64 clips from host memory every step) would cost at that rate."""
import time
import torch
x = torch.empty(64, 16, 3, 112, 112).pin_memory()
d = torch.empty_like(x, device="cuda")
d.copy_(x, non_blocking=True); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    d.copy_(x, non_blocking=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
gb = x.numel() * 4 / 1e9
print("pinned H2D: %.1f GB/s (%.2f ms per 64-clip class batch); 50 classes per step: %.1f ms of copies -> at most %.1f steps/s if not overlapped"
      % (gb / dt, dt * 1e3, 50 * dt * 1e3, 1.0 / (50 * dt)))
y = torch.empty(64, 16, 3, 112, 112)
t0 = time.perf_counter()
for _ in range(5):
    d.copy_(y)
torch.cuda.synchronize()
dt2 = (time.perf_counter() - t0) / 5
print("pageable H2D (what `.to(device)` of an un-pinned batch does): %.1f GB/s, 50 classes: %.1f ms" % (gb / dt2, 50 * dt2 * 1e3))
