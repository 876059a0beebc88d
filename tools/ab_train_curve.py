import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from video_distillation_amd import networks, train
torch.manual_seed(0)
B, K = 50, 50
x = train.standardize(torch.randn(B, 16, 3, 112, 112, device="cuda"))
lab = torch.arange(K, device="cuda")
net = networks.ConvNet3D(3, K, 128, 3, 'relu', 'none', 'maxpooling', frames=16, im_size=(112, 112)).cuda().train()
net.dropout.p = 0.0
net2 = copy.deepcopy(net)
crit = nn.CrossEntropyLoss().cuda()
o1 = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=5e-4)
o2 = torch.optim.SGD(net2.parameters(), lr=0.01, momentum=0.9, weight_decay=5e-4)
for ep in range(25):
    out, loss = net.hip_train_step(x, lab, o1)
    out2 = net2(x); loss2 = crit(out2, lab); o2.zero_grad(); loss2.backward(); o2.step()
    if ep % 4 == 0 or ep == 24:
        print(ep, "hip %.5f acc %.2f | torch %.5f acc %.2f" % (float(loss), float((out.argmax(1) == lab).float().mean()), float(loss2), float((out2.argmax(1) == lab).float().mean())), flush=True)
