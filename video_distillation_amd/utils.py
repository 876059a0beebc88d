"""The reference's library surface for the hot path (``utils.py`` of yuz1wan/video_distillation),
re-implemented over the HIP kernels.

Same names, argument meaning and error behaviour as the reference functions cited in each
docstring, so that its driver scripts can ``from video_distillation_amd.utils import ...``
unchanged.  ``args`` is the reference's duck-typed namespace (``device``, ``dis_metric``,
``lr_net``, ``epoch_eval_train``, ``batch_train``, ``model``, ``eval_mode``).

Everything here runs on the HIP kernels -- since round 5 also ``Conv3DNet(mode='add')``, a constructor option no script of
the reference selects (utils.py:1179, default 'concat'): the same fused kernel with a derived 4-channel weight.
"""
from __future__ import annotations

import ctypes
import random
import time
from collections import defaultdict

import numpy as np
import torch
import torch.nn as nn
from torch.utils.data import Dataset

from . import hip
from .networks import ConvNet3D

_REFERENCE_2D_ZOO = ('MLP', 'ConvNet', 'LeNet', 'AlexNet', 'AlexNetBN', 'VGG11', 'VGG11BN', 'ResNet18', 'ResNet18BN_AP',
                     'ResNet18BN', 'VideoConvNetMean', 'VideoConvNetMLP', 'VideoConvNetLSTM', 'VideoConvNetRNN',
                     'VideoConvNetGRU')


def get_time():
    return str(time.strftime("[%Y-%m-%d %H:%M:%S]", time.localtime()))


def get_default_convnet_setting():
    """utils.py:512-514."""
    return 128, 3, 'relu', 'instancenorm', 'avgpooling'


def get_network(model, channel, num_classes, im_size=(32, 32), frames=16, dist=True):
    """``get_network`` (utils.py:518-625).  Only the hot path's architecture is in scope:
    'ConvNet3D' = width 128, depth 3, ReLU, no norm, max pooling (utils.py:608-609).  Like the
    reference it reseeds the global RNG from the wall clock on every call, wraps in
    DataParallel when ``dist`` and more than one device is visible, and ``exit()``s on an
    unknown name."""
    torch.random.manual_seed(int(time.time() * 1000) % 100000)
    net_width, net_depth, net_act, _, _ = get_default_convnet_setting()
    if model == 'ConvNet3D':
        net = ConvNet3D(channel=channel, num_classes=num_classes, net_width=net_width, net_depth=net_depth,
                        net_act=net_act, net_norm='none', net_pooling='maxpooling', im_size=im_size, frames=frames)
    elif model in _REFERENCE_2D_ZOO:
        raise NotImplementedError("%s is outside the accelerated hot path (SURVEY.md section 2); only ConvNet3D is built" % model)
    else:
        net = None
        exit('unknown model: %s' % model)
    if dist:
        gpu_num = torch.cuda.device_count()
        if gpu_num > 0:
            device = 'cuda'
            net = net.to(device)
            if gpu_num > 1:
                net = SingleDeviceParallel(net)
        else:
            device = 'cpu'
            net = net.to(device)
    return net


class SingleDeviceParallel(nn.Module):
    """What ``get_network(dist=True)`` returns when several GPUs are visible.  The reference wraps the net in
    ``nn.DataParallel`` there (utils.py:615-623) and its drivers reach through ``.module`` (distill_baseline.py:339-347).
    This build scales as ONE PROCESS PER GPU (torch.distributed over RCCL) -- threads replicating a
    module over devices inside one process are exactly what it avoids -- so the wrapper keeps the ``.module`` surface and
    runs the net on its own device."""

    def __init__(self, module: nn.Module):
        super().__init__()
        self.module = module

    def forward(self, *inputs, **kwargs):
        return self.module(*inputs, **kwargs)


class TensorDataset(Dataset):
    """utils.py:499-509."""

    def __init__(self, images, labels):
        self.images = images.detach().float()
        self.labels = labels.detach()

    def __getitem__(self, index):
        return self.images[index], self.labels[index]

    def __len__(self):
        return self.images.shape[0]


class MultiStaticSharedDataset(Dataset):
    """utils.py:462-496: every item composes one clip from a randomly picked static image,
    dynamic memory and hallucinator of its class.  Supports spc/C in {2 (vpc 1), 10 (vpc 5)}."""

    def __init__(self, static, dynamic, hallucinator):
        self.static = static.detach().float()
        self.dynamic = dynamic.detach().float()
        self.hallucinator = hallucinator
        self.n_s = static.shape[0]
        self.n_c, self.dpc = dynamic.shape[0], dynamic.shape[1]

    def __getitem__(self, index):
        per_s = self.n_s // self.n_c
        if per_s == 10:
            label, idx = index // 5, index % 5
            static_idx = label * per_s + 2 * idx + random.randint(0, 1)
            dynamic_idx = 2 * idx + random.randint(0, 1)
        elif per_s == 2:
            label = index
            static_idx = random.randint(0, per_s - 1) + label * per_s
            dynamic_idx = random.randint(0, self.dpc - 1)
        else:
            print("error for multi-static-shared-dataset")
            exit()
        hal = self.hallucinator[random.randint(0, len(self.hallucinator) - 1)]
        video = hal(self.static[static_idx].unsqueeze(0), self.dynamic[label, dynamic_idx].unsqueeze(0))
        return video[0], label

    def __len__(self):
        if self.n_s == self.n_c * 10:
            return self.n_c * 5
        if self.n_s == self.n_c * 2:
            return self.n_c
        print("error for multi-static-shared-dataset")
        exit()


# ------------------------------------------------------------------------------------------
# hallucinator
# ------------------------------------------------------------------------------------------
class _HallucinatorFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, static, dynamic, weight, bias):
        n, T, _, H, W = dynamic.shape
        st = static.detach().float().contiguous()
        dy = dynamic.detach().float().contiguous()
        w = weight.detach().float().contiguous()
        b = bias.detach().float().contiguous()
        out = torch.empty((n, T, 3, H, W), dtype=torch.float32, device=dynamic.device)
        hip.check(hip.lib().vd_hallucinator_fwd(hip.ptr(st), hip.ptr(dy), None, None, hip.ptr(w), hip.ptr(b),
                                                n, T, H, W, hip.ptr(out), hip.stream_ptr(out.device)),
                  "vd_hallucinator_fwd")
        ctx.save_for_backward(st, dy, w)
        return out

    @staticmethod
    def backward(ctx, g):
        st, dy, w = ctx.saved_tensors
        n, T, _, H, W = dy.shape
        g = g.float().contiguous()
        g_dyn = torch.zeros_like(dy)
        need_static = ctx.needs_input_grad[0]
        g_stat = torch.zeros_like(st) if need_static else None
        g_w = torch.zeros(324, dtype=torch.float32, device=g.device)
        g_b = torch.zeros(3, dtype=torch.float32, device=g.device)
        hip.check(hip.lib().vd_hallucinator_bwd(hip.ptr(g), hip.ptr(st), hip.ptr(dy), None, None, hip.ptr(w),
                                                n, T, H, W, hip.ptr(g_dyn), hip.ptr(g_stat), hip.ptr(g_w),
                                                hip.ptr(g_b), hip.stream_ptr(g.device)), "vd_hallucinator_bwd")
        return g_stat, g_dyn, g_w.view(3, 4, 3, 3, 3), g_b


class Conv3DNet(nn.Module):
    """The dynamic-memory hallucinator (utils.py:1178-1197): the static image is repeated over
    the T frames, concatenated channel-wise with the 1-channel dynamic memory and passed
    through ``encoder = Conv3d(4, 3, 3, padding=1)``; returns (n, T, 3, H, W).  The repeat /
    permute / cat are never materialised: one fused HIP kernel reads both memories directly."""

    def __init__(self, in_channel=4, mid_channel=3, out_channel=3, img_size=112, kernel_size=3, mode='concat'):
        super().__init__()
        self.mode = mode
        if mode == 'add':
            in_channel = 3
        self.encoder = nn.Conv3d(in_channel, mid_channel, kernel_size, padding=1)

    def forward(self, static, dynamic):
        if self.mode not in ('concat', 'add'):
            raise NotImplementedError
        if not dynamic.is_cuda:
            raise RuntimeError("Conv3DNet has no CPU path: move the memories to a HIP device")
        weight = self.encoder.weight
        if self.mode == 'add':
            # utils.py:1193: x = static + dynamic (the one dynamic channel broadcast over the three static ones), then Conv3d(3 -> 3).
            # The convolution is linear: conv(static + dyn, W) = conv(cat(static, dyn), W') with W'[:, :3] = W and W'[:, 3] =
            # sum_ci W[:, ci] -- the SAME fused kernel with a derived 4-channel weight (a 108-float tensor op; autograd carries
            # d/dW' back: dW[:, ci] = dW'[:, ci] + dW'[:, 3])
            if tuple(weight.shape) != (3, 3, 3, 3, 3):
                raise NotImplementedError("HIP hallucinator (mode='add') is built for Conv3d(3->3, k=3)")
            weight = torch.cat([weight, weight.sum(1, keepdim=True)], 1)
        elif tuple(weight.shape) != (3, 4, 3, 3, 3):
            raise NotImplementedError("HIP hallucinator is built for Conv3d(4->3, k=3)")
        return _HallucinatorFunction.apply(static, dynamic, weight, self.encoder.bias)


# ------------------------------------------------------------------------------------------
# match_loss
# ------------------------------------------------------------------------------------------
def _rows_view(t):
    """How distance_wb (utils.py:634-651) views a gradient tensor as [rows][len]: 4-D / 3-D are
    flattened to (dim0, rest), 2-D kept, 1-D ignored, and EVERYTHING ELSE -- the 5-D Conv3d
    weight gradients -- is left as is, so rows run over the last axis only."""
    if t.dim() in (3, 4):
        return t.shape[0], int(np.prod(t.shape[1:]))
    return int(np.prod(t.shape[:-1])), t.shape[-1]


class _MatchLossFunction(torch.autograd.Function):
    """All three metrics over a list of gradient tensors; differentiable w.r.t. gw_syn.  One launch per direction for
    the whole list (vd_match_rows_*_multi)."""

    @staticmethod
    def forward(ctx, mode, n, *tensors):
        gw_syn, gw_real = tensors[:n], tensors[n:]
        dev = gw_syn[0].device
        L, st = hip.lib(), hip.stream_ptr(dev)
        acc = torch.zeros(5, dtype=torch.float32, device=dev)
        keep, batches = [], [hip.VdMatchBatch()]
        for gs, gr in zip(gw_syn, gw_real):
            if mode == 0 and gs.dim() == 1:
                keep.append(None)
                continue
            gs_c, gr_c = gs.detach().float().contiguous(), gr.detach().float().contiguous()
            rows, ln = _rows_view(gs_c) if mode == 0 else (gs_c.numel(), 1)   # mse / cos: flat sums
            if rows <= 0 or ln <= 0:
                keep.append(None)
                continue
            if batches[-1].nseg == 16:
                batches.append(hip.VdMatchBatch())
            b = batches[-1]
            b.reserved = (1, 2, 28)[mode]               # the sums this metric reads: acc[0] / acc[1] / acc[2..4]
            sg = b.seg[b.nseg]
            sg.gr, sg.gs, sg.g, sg.rows, sg.len, sg.reserved = gr_c.data_ptr(), gs_c.data_ptr(), 0, rows, ln, int(mode != 0)
            b.nseg += 1
            keep.append((gs_c, gr_c, rows, ln))
        for b in batches:
            hip.check(L.vd_match_rows_fwd_multi(ctypes.byref(b), hip.ptr(acc), st), "vd_match_rows_fwd_multi")
        ctx.keep, ctx.mode, ctx.n = keep, mode, n
        ctx.acc = acc
        if mode == 0:
            return acc[0].clone()
        if mode == 1:
            return acc[1].clone()
        return 1 - acc[2] / (torch.sqrt(acc[3]) * torch.sqrt(acc[4]) + 0.000001)

    @staticmethod
    def backward(ctx, gout):
        L = hip.lib()
        gout = gout.detach().float().contiguous().view(1)
        st = hip.stream_ptr(gout.device)
        grads, batches = [], [hip.VdMatchBatch()]
        for item in ctx.keep:
            if item is None:
                grads.append(None)
                continue
            gs_c, gr_c, rows, ln = item
            g = torch.empty_like(gs_c)
            if batches[-1].nseg == 16:
                batches.append(hip.VdMatchBatch())
            b = batches[-1]
            sg = b.seg[b.nseg]
            sg.gr, sg.gs, sg.g, sg.rows, sg.len, sg.reserved = gr_c.data_ptr(), gs_c.data_ptr(), g.data_ptr(), rows, ln, int(ctx.mode != 0)
            b.nseg += 1
            grads.append(g)
        for b in batches:
            hip.check(L.vd_match_rows_bwd_multi(ctypes.byref(b), ctx.mode, hip.ptr(ctx.acc), hip.ptr(gout), st),
                      "vd_match_rows_bwd_multi")
        return (None, None) + tuple(grads) + (None,) * ctx.n


def distance_wb(gwr, gws):
    """utils.py:634-651 for a single layer."""
    if gwr.dim() == 1:
        return torch.tensor(0, dtype=torch.float, device=gwr.device)
    return _MatchLossFunction.apply(0, 1, gws, gwr)


def match_loss(gw_syn, gw_real, args):
    """``match_loss`` (utils.py:655-687): 'ours' = sum over layers of row-wise cosine distances
    (1-D layers contribute 0), 'mse' = squared L2 of the concatenation, 'cos' = cosine distance
    of the concatenation.  Returns a 0-dim tensor on ``args.device``; ``exit()``s on an
    unknown metric like the reference."""
    modes = {'ours': 0, 'mse': 1, 'cos': 2}
    if args.dis_metric not in modes:
        exit('unknown distance function: %s' % args.dis_metric)
    gw_syn, gw_real = list(gw_syn), list(gw_real)
    if len(gw_real) == 0:
        return torch.tensor(0.0).to(args.device)
    if not gw_syn[0].is_cuda:
        raise RuntimeError("match_loss has no CPU path: gradients must live on a HIP device")
    return _MatchLossFunction.apply(modes[args.dis_metric], len(gw_syn), *gw_syn, *gw_real)


def get_loops(ipc, dataset=None):
    """utils.py:691-709."""
    table = {1: (1, 1), 5: (1, 1), 10: (10, 50), 20: (20, 25), 30: (30, 20), 40: (40, 15), 50: (50, 10)}
    if ipc not in table:
        exit('loop hyper-parameters are not defined for %d ipc' % ipc)
    return table[ipc]


def get_eval_pool(eval_mode, model, model_eval):
    """utils.py:973-996 restricted to the modes the video scripts use ('S', 'SS', explicit)."""
    if eval_mode == 'S':
        return [model[:model.index('BN')]] if 'BN' in model else [model]
    if eval_mode == 'SS':
        return [model]
    if eval_mode in ('M', 'B', 'W', 'D', 'A', 'P', 'N'):
        raise NotImplementedError("eval_mode %s sweeps the reference's 2-D model zoo (out of scope)" % eval_mode)
    return [model_eval]


# ------------------------------------------------------------------------------------------
# evaluate_synset / epoch
# ------------------------------------------------------------------------------------------
def _standardize(img):
    """(img - img.mean()) / img.std() (utils.py:770); HIP kernel for device tensors."""
    if img.is_cuda:
        from . import train
        return train.standardize(img)
    return (img - img.mean()) / img.std()


def epoch(mode, dataloader, net, optimizer, criterion, args):
    """``epoch`` (utils.py:752-844): one training pass, or THREE test passes (test clips resample
    their start frame on every read); batch-global standardisation with the unbiased std."""
    loss_avg, acc_avg, num_exp = 0, 0, 0
    top5_acc_avg, top3_acc_avg, top1_acc_avg = 0.0, 0.0, 0.0
    net = net.to(args.device)
    criterion = criterion.to(args.device)
    net.train() if mode == 'train' else net.eval()
    correct_per_class = defaultdict(list)
    stat_loss, stat_match, stat_lab, stat_topk = [], [], [], {3: [], 5: []}
    passes = 1 if mode == 'train' else 3
    for _ in range(passes):
        for datum in dataloader:
            img = datum[0].float().to(args.device)
            if 'Video' in args.model:
                img = img[:, :, :, 24:-24, 24:-24]
            img = _standardize(img)
            lab = datum[1].long().to(args.device)
            n_b = lab.shape[0]
            core = net.module if isinstance(net, SingleDeviceParallel) else net
            hip_step = (mode == 'train' and hasattr(core, 'hip_trainable') and core.hip_trainable(img, optimizer, criterion))
            if hip_step:    # forward + loss + backward + optimizer.step() fused on the HIP path
                output, loss = core.hip_train_step(img, lab, optimizer)
            else:           # any other optimiser / loss: the autograd Functions of ConvNet3D.forward (HIP as well)
                output = net(img)
                loss = criterion(output, lab)
            # statistics stay on the device; one host transfer per epoch instead of one per batch
            with torch.no_grad():
                out_d = output.detach()
                order = torch.argsort(out_d, dim=-1)
                matched = out_d.argmax(dim=-1) == lab          # np.argmax semantics: first maximum
                stat_loss.append(loss.detach().reshape(1) * n_b)
                stat_match.append(matched)
                stat_lab.append(lab)
                for k in (3, 5):
                    stat_topk[k].append((order[:, -k:] == lab[:, None]).any(dim=1).sum().reshape(1))
            num_exp += n_b
            if mode == 'train' and not hip_step:
                optimizer.zero_grad()
                loss.backward()
                optimizer.step()
    matched = torch.cat(stat_match).cpu().numpy()
    labs = torch.cat(stat_lab).cpu().numpy()
    loss_avg = float(torch.cat(stat_loss).sum().cpu()) / num_exp
    acc_avg = float(np.sum(matched)) / num_exp
    top5_acc_avg = float(torch.cat(stat_topk[5]).sum().cpu())
    if mode != 'train':
        top1_acc_avg = float(np.sum(matched))
        top3_acc_avg = float(torch.cat(stat_topk[3]).sum().cpu())
    for y, c in zip(labs.tolist(), matched.tolist()):
        correct_per_class[y].append(c)
    top_acc_avg = [acc_avg, top1_acc_avg / num_exp, top3_acc_avg / num_exp, top5_acc_avg / num_exp]
    per_class = dict(correct_per_class)
    per_class = [np.mean(per_class[i]) if i in per_class else None for i in range(len(per_class))]
    if args.eval_mode == 'top5':
        return loss_avg, top_acc_avg, per_class
    return loss_avg, acc_avg, per_class


def evaluate_synset(it_eval, net, images_train, labels_train, testloader, args, mode='hallucinator',
                    return_loss=False, test_freq=None):
    """``evaluate_synset`` (utils.py:848-886): train ``net`` on the synthetic set for
    ``epoch_eval_train``+1 epochs with SGD(lr_net, m=.9, wd=5e-4); lr*0.1 and a fresh optimiser
    after epoch ``Epoch//2+1``; test at the end (or every ``test_freq``).  Modes 'none'
    (images+labels) and 'multi-static' ((static, dynamic, hallucinators)); anything else raises
    NotImplementedError like the reference.  Returns (net, acc_train, acc_test, acc_per_class)."""
    lr = float(args.lr_net)
    Epoch = int(args.epoch_eval_train)
    lr_schedule = [Epoch // 2 + 1]
    optimizer = torch.optim.SGD(net.parameters(), lr=lr, momentum=0.9, weight_decay=0.0005)
    criterion = nn.CrossEntropyLoss().to(args.device)
    if mode == 'none':
        dst_train = TensorDataset(images_train, labels_train)
    elif mode == 'multi-static':
        dst_train = MultiStaticSharedDataset(images_train[0], images_train[1], images_train[2])
    else:
        raise NotImplementedError
    trainloader = torch.utils.data.DataLoader(dst_train, batch_size=args.batch_train, shuffle=True, num_workers=0)
    start = time.time()
    acc_test, acc_per = None, None
    for ep in range(Epoch + 1):
        loss_train, acc_train, _ = epoch('train', trainloader, net, optimizer, criterion, args)
        if (test_freq is None and ep == Epoch) or (test_freq is not None and ep % test_freq == 0 and ep != 0):
            with torch.no_grad():
                loss_test, acc_test, acc_per = epoch('test', testloader, net, optimizer, criterion, args)
        if ep in lr_schedule:
            lr *= 0.1
            optimizer = torch.optim.SGD(net.parameters(), lr=lr, momentum=0.9, weight_decay=0.0005)
    if args.eval_mode != 'top5':
        print('%s Evaluate_%02d: Ep %d time = %ds loss = %.6f train acc = %.2f, test acc = %.2f' % (
            get_time(), it_eval, Epoch, int(time.time() - start), loss_train, acc_train * 100, acc_test * 100))
    return net, acc_train, acc_test, acc_per
