import sys, os, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cProfile, pstats
from video_distillation_amd import utils
dev = torch.device("cuda:0")
C = 50
syn = torch.randn(C, 16, 3, 112, 112, device=dev); labels = torch.arange(C, device=dev)
net = utils.get_network("ConvNet3D", 3, C, (112, 112), frames=16, dist=False).to(dev)
args = types.SimpleNamespace(device=dev, lr_net=0.01, epoch_eval_train=30, batch_train=256, model="ConvNet3D", eval_mode="SS")
opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=5e-4)
crit = torch.nn.CrossEntropyLoss().to(dev)
loader = torch.utils.data.DataLoader(utils.TensorDataset(syn, labels), batch_size=256, shuffle=True, num_workers=0)
for _ in range(3): utils.epoch('train', loader, net, opt, crit, args)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): utils.epoch('train', loader, net, opt, crit, args)
torch.cuda.synchronize(); print("epoch %.2f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
t0 = time.perf_counter()
for _ in range(20):
    for d in loader: pass
torch.cuda.synchronize(); print("loader only %.2f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(10): utils.epoch('train', loader, net, opt, crit, args)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
