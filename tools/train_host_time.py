"""Host issue time vs device time of the evaluate_synset training step (ConvNet3D.hip_train_step).
usage: python tools/train_host_time.py [B] [T H W] [K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from video_distillation_amd import networks, train

B = int(sys.argv[1]) if len(sys.argv) > 1 else 50
T, H, W = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (16, 112, 112)
K = int(sys.argv[5]) if len(sys.argv) > 5 else 50
torch.manual_seed(0)
net = networks.ConvNet3D(3, K, 128, 3, 'relu', 'none', 'maxpooling', frames=T, im_size=(H, W)).cuda().train()
opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=5e-4)
x = train.standardize(torch.randn(B, T, 3, H, W, device="cuda"))
lab = torch.randint(0, K, (B,), device="cuda")
crit = nn.CrossEntropyLoss().cuda()
assert net.hip_trainable(x, opt, crit)
for _ in range(3):
    net.hip_train_step(x, lab, opt)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    net.hip_train_step(x, lab, opt)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("B=%d (%d,%d,%d) K=%d: host issue %.2f ms/step, total %.2f ms/step" % (B, T, H, W, K, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
