"""20 evaluate_synset training steps (f16x3 / f16x3, B clips) and nothing else: for rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from video_distillation_amd import networks, train
B = int(sys.argv[1]) if len(sys.argv) > 1 else 50
T, H, W, K = 16, 112, 112, 50
torch.manual_seed(0)
net = networks.ConvNet3D(3, K, 128, 3, 'relu', 'none', 'maxpooling', frames=T, im_size=(H, W)).cuda().train()
opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=5e-4)
x = train.standardize(torch.randn(B, T, 3, H, W, device="cuda"))
lab = torch.randint(0, K, (B,), device="cuda")
for _ in range(20):
    net.hip_train_step(x, lab, opt)
torch.cuda.synchronize()
