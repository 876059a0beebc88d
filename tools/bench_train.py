"""Time one evaluate_synset training iteration (utils.py:765-792): HIP train step vs the same
graph on torch-ROCm ops (MIOpen).  usage: python tools/bench_train.py [B] [T H W] [K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from video_distillation_amd import networks, train

B = int(sys.argv[1]) if len(sys.argv) > 1 else 50
T, H, W = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (16, 112, 112)
K = int(sys.argv[5]) if len(sys.argv) > 5 else 50


def make():
    torch.manual_seed(0)
    net = networks.ConvNet3D(3, K, 128, 3, 'relu', 'none', 'maxpooling', frames=T, im_size=(H, W)).cuda().train()
    opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=5e-4)
    return net, opt


x = train.standardize(torch.randn(B, T, 3, H, W, device="cuda"))
lab = torch.randint(0, K, (B,), device="cuda")
crit = nn.CrossEntropyLoss().cuda()


def timeit(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for prec, pb in (("f16x3", "f16x3"), ("f16x3", "f16"), ("f16", "f16")):
    networks.set_precision(train=prec, train_bwd=pb)
    net, opt = make()
    assert net.hip_trainable(x, opt, crit)
    ms = timeit(lambda: net.hip_train_step(x, lab, opt))
    print("HIP train step  %-6s/%-6s B=%d (%d,%d,%d): %.2f ms" % (prec, pb, B, T, H, W, ms))

if os.environ.get('VD_SKIP_TORCH') == '1':
    sys.exit(0)
net, opt = make()


def torch_step():
    out = net(x)
    loss = crit(out, lab)
    opt.zero_grad()
    loss.backward()
    opt.step()


try:
    ms = timeit(torch_step, n=3, warm=1)
    print("torch-ROCm ops step (fp32 MIOpen) B=%d: %.2f ms" % (B, ms))
except Exception as e:   # noqa
    print("torch path failed:", repr(e)[:200])
