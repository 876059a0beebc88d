#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference.

Runs only in the build container (needs /root/reference; CPU fp32, torch 2.10).  The
reference's modules are imported unmodified (``torchvision`` is absent here and used only by
``get_dataset``, so empty stub modules are registered for it).  Nothing from the reference
is written into the repo except input/output TENSORS of its functions.

Usage:  python tools/gen_golden.py            # rewrites tests/golden/*.npz
Fixtures follow SURVEY.md section 8(c): G1 layer-wise fwd, G2 DM class term, G3 DM step x2,
G4 hallucinator, G5 s2d DM step, G6 match_loss KATs, G7 evaluate_synset/epoch, G8 is G3
re-used by the sharding tests.
"""
import contextlib
import io
import json
import os
import random
import sys
import types

sys.dont_write_bytecode = True
import numpy as np
import torch

REF = os.environ.get("VD_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def import_reference():
    for name in ("torchvision", "torchvision.datasets", "torchvision.transforms", "torchvision.utils"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["torchvision"].datasets = sys.modules["torchvision.datasets"]
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision"].utils = sys.modules["torchvision.utils"]
    sys.path.insert(0, REF)
    import networks  # noqa
    import utils  # noqa
    return networks, utils


def make_net(networks, seed, num_classes, im, frames):
    # bypass get_network's wall-clock reseed; same settings as utils.py:608-609
    torch.manual_seed(seed)
    return networks.ConvNet3D(channel=3, num_classes=num_classes, net_width=128, net_depth=3,
                              net_act='relu', net_norm='none', net_pooling='maxpooling',
                              im_size=(im, im), frames=frames)


def npz(name, **arrs):
    conv = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = np.asarray(v)
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **conv)
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024))


def param_checksum(net):
    return np.array([float(p.double().sum()) for p in net.parameters()] +
                    [float(p.double().abs().sum()) for p in net.parameters()])


def g1(networks):
    seed = 11
    net = make_net(networks, seed, 50, 64, 8).eval()
    g = torch.Generator().manual_seed(101)
    x = torch.randn(2, 8, 3, 64, 64, generator=g)
    outs = []
    cur = x.permute(0, 2, 1, 3, 4)
    for layer in net.features:
        cur = layer(cur.clone())  # ReLU is in-place in the reference
        outs.append(cur.clone())
    embed = net.embed(x)
    logits = net(x)
    # one 112x112x16 sample: embed + logits only
    net112 = make_net(networks, seed, 50, 112, 16).eval()
    g = torch.Generator().manual_seed(102)
    x112 = torch.randn(1, 16, 3, 112, 112, generator=g)
    npz("g1_layers.npz", seed=seed, x_seed=101, x=x[:, :, :, ::8, ::8],  # strided probe of x only
        conv0=outs[0][:, ::8, :, ::4, ::4], pool0=outs[2][:, ::8, :, ::2, ::2],
        conv1=outs[3][:, ::16], pool1=outs[5][:, ::8], conv2=outs[6][:, ::8], pool2=outs[8],
        embed=embed, logits=logits, checksum=param_checksum(net),
        w0_probe=net.features[0].weight[:4], b2=net.features[6].bias,
        x112_seed=102, embed112=net112.embed(x112), logits112=net112(x112))


def g2_g3(networks):
    seed = 21
    net = make_net(networks, seed, 50, 64, 8).train()
    for p in net.parameters():
        p.requires_grad = False
    g = torch.Generator().manual_seed(201)
    real = torch.randn(4, 8, 3, 64, 64, generator=g)
    syn = torch.randn(1, 8, 3, 64, 64, generator=g).requires_grad_(True)
    out_real = net.embed(real).detach()
    out_syn = net.embed(syn)
    loss = torch.sum((torch.mean(out_real, dim=0) - torch.mean(out_syn, dim=0)) ** 2)
    loss.backward()
    npz("g2_dm_class.npz", seed=seed, data_seed=201, loss=loss, grad_syn=syn.grad,
        feat_real_mean=out_real.mean(0), feat_syn=out_syn)

    # G3: two DM iterations, C=3 classes, ipc=1, batch_real=4, SGD(lr, momentum .5)
    C, ipc, lr = 3, 1, 0.5
    g = torch.Generator().manual_seed(301)
    image_syn = torch.randn(C * ipc, 8, 3, 64, 64, generator=g).requires_grad_(True)
    opt = torch.optim.SGD([image_syn], lr=lr, momentum=0.5)
    losses, syn_after, grads = [], [], []
    for it in range(2):
        net = make_net(networks, 31 + it, 50, 64, 8).train()
        for p in net.parameters():
            p.requires_grad = False
        gr = torch.Generator().manual_seed(310 + it)
        loss = torch.tensor(0.0)
        for c in range(C):
            img_real = torch.randn(4, 8, 3, 64, 64, generator=gr)
            img_syn = image_syn[c * ipc:(c + 1) * ipc].reshape((ipc, 8, 3, 64, 64))
            output_real = net.embed(img_real).detach()
            output_syn = net.embed(img_syn)
            loss += torch.sum((torch.mean(output_real, dim=0) - torch.mean(output_syn, dim=0)) ** 2)
        opt.zero_grad()
        loss.backward()
        grads.append(image_syn.grad.clone())
        opt.step()
        losses.append(float(loss))
        syn_after.append(image_syn.detach().clone())
    npz("g3_dm_steps.npz", net_seeds=[31, 32], syn_seed=301, real_seeds=[310, 311], lr=lr,
        momentum=0.5, losses=losses, grad0=grads[0][:, ::2, :, ::4, ::4], grad1=grads[1][:, ::2, :, ::4, ::4],
        syn1=syn_after[0][:, ::2, :, ::4, ::4], syn2=syn_after[1][:, ::2, :, ::4, ::4],
        syn2_sum=float(syn_after[1].double().sum()), syn2_abs=float(syn_after[1].double().abs().sum()))


def g4_g5(networks, utils):
    torch.manual_seed(41)
    hal = utils.Conv3DNet()
    g = torch.Generator().manual_seed(401)
    static = torch.randn(3, 3, 64, 64, generator=g).requires_grad_(True)
    dynamic = torch.randn(3, 8, 1, 64, 64, generator=g).requires_grad_(True)
    up = torch.randn(3, 8, 3, 64, 64, generator=g)
    out = hal(static, dynamic)
    (out * up).sum().backward()
    npz("g4_hallucinator.npz", weight=hal.encoder.weight, bias=hal.encoder.bias, data_seed=401,
        out=out[:, :, :, ::2, ::2], out_sum=float(out.double().sum()),
        g_dynamic=dynamic.grad[:, :, :, ::2, ::2], g_static=static.grad[:, :, ::2, ::2],
        g_weight=hal.encoder.weight.grad, g_bias=hal.encoder.bias.grad,
        g_dynamic_abs=float(dynamic.grad.double().abs().sum()), g_static_abs=float(static.grad.double().abs().sum()))

    # G5: one s2d DM step, C=3, vpc=1, spc=2, dpc=2, static frozen, SGD(.95) on dynamic + hal
    C, vpc, spc, dpc = 3, 1, 2, 2
    torch.manual_seed(51)
    hal = utils.Conv3DNet()
    w0, b0 = hal.encoder.weight.detach().clone(), hal.encoder.bias.detach().clone()
    g = torch.Generator().manual_seed(501)
    static_syn = torch.randn(C * spc, 3, 64, 64, generator=g)
    dynamic_syn = torch.randn(C, dpc, 8, 1, 64, 64, generator=g).requires_grad_(True)
    draws_dyn = torch.tensor([1, 0, 1])
    draws_sta = torch.tensor([0, 1, 1])
    opt_d = torch.optim.SGD([dynamic_syn], lr=10.0, momentum=0.95)
    opt_h = torch.optim.SGD(hal.parameters(), lr=0.01, momentum=0.95)
    net = make_net(networks, 52, 50, 64, 8).train()
    for p in net.parameters():
        p.requires_grad = False
    label = torch.tensor(np.stack([np.ones(vpc) * i for i in range(0, C)]), dtype=torch.long).view(-1)
    ran = torch.arange(0, C * vpc)
    idx = ran % vpc
    dynamic_idx = 2 * idx + draws_dyn
    static_idx = spc * label + 2 * idx + draws_sta
    image_syn = hal(static_syn[static_idx], dynamic_syn[label, dynamic_idx])
    gr = torch.Generator().manual_seed(510)
    loss = torch.tensor(0.0)
    for c in range(C):
        img_real = torch.randn(4, 8, 3, 64, 64, generator=gr)
        img_syn = image_syn[c * vpc:(c + 1) * vpc].reshape((vpc, 8, 3, 64, 64))
        loss += torch.sum((torch.mean(net.embed(img_real).detach(), dim=0) - torch.mean(net.embed(img_syn), dim=0)) ** 2)
    opt_d.zero_grad(); opt_h.zero_grad()
    loss.backward()
    gd = dynamic_syn.grad.clone()
    gw, gb = hal.encoder.weight.grad.clone(), hal.encoder.bias.grad.clone()
    opt_d.step(); opt_h.step()
    npz("g5_s2d_step.npz", hal_w=w0, hal_b=b0, data_seed=501, net_seed=52, real_seed=510,
        draws_dyn=draws_dyn, draws_sta=draws_sta, dynamic_idx=dynamic_idx, static_idx=static_idx,
        loss=float(loss), g_dynamic=gd[:, :, :, :, ::4, ::4], g_dynamic_abs=float(gd.double().abs().sum()),
        g_dynamic_rowabs=gd.abs().sum(dim=(2, 3, 4, 5)), g_hal_w=gw, g_hal_b=gb,
        hal_w_after=hal.encoder.weight, hal_b_after=hal.encoder.bias,
        dynamic_after_sum=float(dynamic_syn.detach().double().sum()),
        dynamic_after=dynamic_syn.detach()[:, :, :, :, ::4, ::4])


class _Args:
    pass


def g6(networks, utils):
    args = _Args(); args.device = 'cpu'
    g = torch.Generator().manual_seed(601)
    shapes = [(4, 3, 3, 7, 7), (4,), (5, 4, 3, 7, 7), (5,), (6, 5), (6,), (2, 3, 4), (3, 2, 2, 2)]
    gr = [torch.randn(s, generator=g) for s in shapes]
    gs = [torch.randn(s, generator=g).requires_grad_(True) for s in shapes]
    gr[0][1, 2, 0, 3] = 0.0  # zero-norm row in a 5-D member -> epsilon path
    rec = {"n": len(shapes)}
    for i, (a, b) in enumerate(zip(gr, gs)):
        rec["r%d" % i] = a; rec["s%d" % i] = b.detach()
    for metric in ("ours", "mse", "cos"):
        args.dis_metric = metric
        for b in gs:
            b.grad = None
        val = utils.match_loss(gs, gr, args)
        val.backward()
        rec["val_" + metric] = val.detach()
        for i, b in enumerate(gs):
            rec["grad_%s_%d" % (metric, i)] = b.grad if b.grad is not None else torch.zeros_like(b)
    # per-layer 'ours' values (documents the 5-D fall-through and 1-D -> 0)
    rec["ours_per_layer"] = np.array([float(utils.distance_wb(a, b.detach())) for a, b in zip(gr, gs)])
    # a real ConvNet3D gradient pair (dropout made deterministic by seeding)
    net = make_net(networks, 61, 5, 64, 8).train()
    gd = torch.Generator().manual_seed(602)
    xr = torch.randn(2, 8, 3, 64, 64, generator=gd); yr = torch.tensor([1, 3])
    xs = torch.randn(2, 8, 3, 64, 64, generator=gd); ys = torch.tensor([1, 3])
    crit = torch.nn.CrossEntropyLoss()
    torch.manual_seed(611); gw_real = torch.autograd.grad(crit(net(xr), yr), list(net.parameters()))
    torch.manual_seed(611); gw_syn = torch.autograd.grad(crit(net(xs), ys), list(net.parameters()))
    for metric in ("ours", "mse", "cos"):
        args.dis_metric = metric
        rec["net_" + metric] = utils.match_loss(list(gw_syn), list(gw_real), args)
    rec["net_seed"] = 61; rec["net_data_seed"] = 602; rec["net_drop_seed"] = 611
    rec["net_gw_real_l1"] = np.array([float(t.double().abs().sum()) for t in gw_real])
    npz("g6_match_loss.npz", **rec)


def g7(networks, utils):
    # evaluate_synset / epoch on a tiny problem, dropout disabled (p=0) so RNG does not matter.
    C, n_test = 3, 6
    args = _Args()
    args.device = 'cpu'; args.lr_net = 0.01; args.epoch_eval_train = 4; args.batch_train = 256
    args.model = 'ConvNet3D'; args.eval_mode = 'SS'
    g = torch.Generator().manual_seed(701)
    images = torch.randn(C, 8, 3, 64, 64, generator=g)
    labels = torch.arange(C)
    test_x = torch.randn(n_test, 8, 3, 64, 64, generator=g)
    test_y = torch.arange(n_test) % C
    testloader = torch.utils.data.DataLoader(utils.TensorDataset(test_x, test_y), batch_size=4, shuffle=False)
    net = make_net(networks, 71, C, 64, 8)
    net.dropout.p = 0.0
    p0 = [p.detach().clone() for p in net.parameters()]
    # record the shuffle order the DataLoader will draw: seed the global RNG, replay with randperm
    torch.manual_seed(711); random.seed(711); np.random.seed(711)
    rec_losses = []
    orig_epoch = utils.epoch

    def spy(mode, loader, net_, opt, crit, a):
        out = orig_epoch(mode, loader, net_, opt, crit, a)
        rec_losses.append((mode, out[0], out[1]))
        return out
    utils.epoch = spy
    try:
        net_out, acc_train, acc_test, acc_per = utils.evaluate_synset(0, net, images, labels, testloader, args, mode='none')
    finally:
        utils.epoch = orig_epoch
    train = [(l, a) for m, l, a in rec_losses if m == 'train']
    test = [(l, a) for m, l, a in rec_losses if m == 'test']
    # top5 layout
    args.eval_mode = 'top5'
    crit = torch.nn.CrossEntropyLoss()
    with torch.no_grad():
        l5, acc5, per5 = orig_epoch('test', testloader, net_out, None, crit, args)
    npz("g7_evaluate.npz", net_seed=71, data_seed=701, C=C, n_test=n_test, lr_net=0.01, epochs=4,
        train_loss=np.array([t[0] for t in train]), train_acc=np.array([t[1] for t in train]),
        test_loss=np.array([t[0] for t in test]), test_acc=np.array([t[1] for t in test]),
        acc_train=acc_train, acc_test=acc_test,
        acc_per=np.array([np.nan if a is None else a for a in acc_per]),
        top5=np.array(acc5), top5_loss=l5,
        params_after_l1=np.array([float(p.double().abs().sum()) for p in net_out.parameters()]),
        params_before_l1=np.array([float(p.double().abs().sum()) for p in p0]),
        logit_w_after=net_out.logit.weight.detach().reshape(C, -1)[:, :16])


def g9(networks, utils):
    # gradient-matching class term the way the upstream DC loop composes it from the reference's own
    # get_network / match_loss (the reference's DC branch is never executed for video):
    # gw_real detached, gw_syn with create_graph=True, match_loss, backward to the synthetic clips.
    args = _Args(); args.device = 'cpu'
    C = 4
    net = make_net(networks, 91, C, 64, 8).train()
    net.dropout.p = 0.0
    g = torch.Generator().manual_seed(901)
    real = torch.randn(3, 8, 3, 64, 64, generator=g); syn = torch.randn(2, 8, 3, 64, 64, generator=g)
    lab_r = torch.full((3,), 2, dtype=torch.long); lab_s = torch.full((2,), 2, dtype=torch.long)
    crit = torch.nn.CrossEntropyLoss()
    params = list(net.parameters())
    gw_real = [t.detach().clone() for t in torch.autograd.grad(crit(net(real), lab_r), params)]
    rec = {"net_seed": 91, "data_seed": 901, "C": C, "label": 2,
           "gw_real_l1": np.array([float(t.double().abs().sum()) for t in gw_real])}
    for metric in ("ours", "mse", "cos"):
        args.dis_metric = metric
        xs = syn.clone().requires_grad_(True)
        gw_syn = torch.autograd.grad(crit(net(xs), lab_s), params, create_graph=True)
        loss = utils.match_loss(gw_syn, gw_real, args)
        loss.backward()
        rec["loss_" + metric] = loss.detach()
        rec["grad_l1_" + metric] = np.array([float(xs.grad[b].double().abs().sum()) for b in range(2)])
        # one full clip for 'ours' (1.5 MB), a frame of each clip for the others
        rec["grad_" + metric] = xs.grad[0] if metric == "ours" else xs.grad[:, 3]
        if metric == "ours":
            rec["gw_syn_l1"] = np.array([float(t.detach().double().abs().sum()) for t in gw_syn])
    npz("g9_grad_match.npz", **rec)


def g10(networks, utils):
    # one MTT iteration (distill_baseline.py:192-275) re-stated around the reference's ReparamModule /
    # ConvNet3D: unrolled student steps with create_graph, normalised parameter distance, backward to the
    # synthetic clips and to syn_lr.  Expert trajectory = seeded start + a seeded perturbation.
    sys.path.insert(0, REF)
    from reparam_module import ReparamModule
    C, n_syn, syn_steps, batch_syn = 3, 4, 2, 2
    net = make_net(networks, 101, C, 64, 8)
    net.dropout.p = 0.0
    starting = [p.detach().clone() for p in net.parameters()]
    g = torch.Generator().manual_seed(1001)
    target = [p + 0.02 * p.abs().mean() * torch.randn(p.shape, generator=g) for p in starting]
    image_syn = torch.randn(n_syn, 8, 3, 64, 64, generator=g).requires_grad_(True)
    label_syn = torch.tensor([0, 1, 2, 0])
    syn_lr = torch.tensor(0.01).requires_grad_(True)
    student_net = ReparamModule(net)
    student_net.train()
    num_params = sum([np.prod(p.size()) for p in (student_net.parameters())])
    target_params = torch.cat([p.reshape(-1) for p in target], 0)
    student_params = [torch.cat([p.reshape(-1) for p in starting], 0).requires_grad_(True)]
    starting_params = torch.cat([p.reshape(-1) for p in starting], 0)
    criterion = torch.nn.CrossEntropyLoss()
    torch.manual_seed(1011)
    indices_chunks, used = [], []
    for step in range(syn_steps):
        if not indices_chunks:
            indices = torch.randperm(len(image_syn))
            indices_chunks = list(torch.split(indices, batch_syn))
        these = indices_chunks.pop()
        used.append(these.clone())
        out = student_net(image_syn[these], flat_param=student_params[-1])
        ce = criterion(out, label_syn[these])
        grad = torch.autograd.grad(ce, student_params[-1], create_graph=True)[0]
        student_params.append(student_params[-1] - syn_lr * grad)
    param_loss = torch.nn.functional.mse_loss(student_params[-1], target_params, reduction="sum")
    param_dist = torch.nn.functional.mse_loss(starting_params, target_params, reduction="sum")
    param_loss = param_loss / num_params
    param_dist = param_dist / num_params
    grand_loss = param_loss / param_dist
    grand_loss.backward()
    npz("g10_mtt_step.npz", net_seed=101, data_seed=1001, C=C, n_syn=n_syn, syn_steps=syn_steps, batch_syn=batch_syn,
        labels=label_syn, indices=torch.stack(used), syn_lr=0.01, grand_loss=grand_loss.detach(),
        grad_lr=syn_lr.grad, grad_l1=np.array([float(image_syn.grad[b].double().abs().sum()) for b in range(n_syn)]),
        grad_img=image_syn.grad[:, ::2, :, ::2, ::2],
        final_l1=float(student_params[-1].detach().double().abs().sum()),
        target_l1=float(target_params.double().abs().sum()))


def g11(networks, utils):
    # evaluate_synset(mode='multi-static') (utils.py:848-886 with MultiStaticSharedDataset :462-496 and two
    # Conv3DNet hallucinators :1178-1197): python `random` picks (static, dynamic, hallucinator) per item,
    # torch's global RNG shuffles the loader; dropout disabled.
    C, n_test = 3, 6
    args = _Args()
    args.device = 'cpu'; args.lr_net = 0.01; args.epoch_eval_train = 2; args.batch_train = 256
    args.model = 'ConvNet3D'; args.eval_mode = 'SS'
    g = torch.Generator().manual_seed(1101)
    static = torch.randn(C * 2, 3, 64, 64, generator=g)
    dynamic = torch.randn(C, 2, 8, 1, 64, 64, generator=g)
    test_x = torch.randn(n_test, 8, 3, 64, 64, generator=g)
    test_y = torch.arange(n_test) % C
    hals = []
    for k in range(2):
        torch.manual_seed(1110 + k)
        hals.append(utils.Conv3DNet(img_size=64))
    testloader = torch.utils.data.DataLoader(utils.TensorDataset(test_x, test_y), batch_size=4, shuffle=False)
    net = make_net(networks, 111, C, 64, 8)
    net.dropout.p = 0.0
    torch.manual_seed(1121); random.seed(1121); np.random.seed(1121)
    rec = []
    orig_epoch = utils.epoch

    def spy(mode, loader, net_, opt, crit, a):
        out = orig_epoch(mode, loader, net_, opt, crit, a)
        rec.append((mode, out[0], out[1]))
        return out
    utils.epoch = spy
    try:
        with torch.no_grad():
            pass
        net_out, acc_train, acc_test, acc_per = utils.evaluate_synset(0, net, (static, dynamic, hals), None, testloader, args,
                                                                      mode='multi-static')
    finally:
        utils.epoch = orig_epoch
    train = [(l, a) for m, l, a in rec if m == 'train']
    npz("g11_multi_static_eval.npz", net_seed=111, data_seed=1101, hal_seeds=np.array([1110, 1111]), rng_seed=1121, C=C,
        n_test=n_test, lr_net=0.01, epochs=2, train_loss=np.array([t[0] for t in train]),
        train_acc=np.array([t[1] for t in train]), acc_train=acc_train, acc_test=acc_test,
        hal_w=torch.stack([h.encoder.weight.detach() for h in hals]), hal_b=torch.stack([h.encoder.bias.detach() for h in hals]),
        params_after_l1=np.array([float(p.double().abs().sum()) for p in net_out.parameters()]))


def g12(networks):
    # G12: 24 DM iterations (distill_baseline.py:334-355 loop body) at 64x64x8, C=2, ipc=1, batch_real=64 (pool 80 per class),
    # SGD(lr .2, momentum .5).  Real clips of a class share a class pattern (base_c + 0.1 noise) and the
    # synthetic clips start from a real clip (--init real, :96-100), so that the feature difference
    # mean f_real - mean f_syn shrinks over the run: this pins the LATE regime of the distillation, where a
    # fixed absolute error on mean f_real weighs more on the gradient than at a random start.
    C, ipc, B, steps, lr, NP = 2, 1, 64, 24, 50.0, 80
    g = torch.Generator().manual_seed(1201)
    base = torch.randn(C, 8, 3, 64, 64, generator=g)
    pool = torch.stack([base[c] + 0.1 * torch.randn(NP, 8, 3, 64, 64, generator=g) for c in range(C)])   # (C,NP,...)
    image_syn = pool[:, 0].clone().requires_grad_(True)
    opt = torch.optim.SGD([image_syn], lr=lr, momentum=0.5)
    losses, grads, syns, rel_diff, picks = [], [], [], [], []
    for it in range(steps):
        net = make_net(networks, 1210 + it, 50, 64, 8).train()
        for p in net.parameters():
            p.requires_grad = False
        rng = np.random.default_rng([1202, it])
        loss = torch.tensor(0.0)
        pk, rd = [], []
        for c in range(C):
            idx = rng.permutation(NP)[:B]
            pk.append(idx)
            img_real = pool[c, torch.as_tensor(idx)]
            img_syn = image_syn[c * ipc:(c + 1) * ipc].reshape((ipc, 8, 3, 64, 64))
            output_real = net.embed(img_real).detach()
            output_syn = net.embed(img_syn)
            d = torch.mean(output_real, dim=0) - torch.mean(output_syn, dim=0)
            rd.append(float(d.detach().norm() / torch.mean(output_real, dim=0).norm()))
            loss += torch.sum(d ** 2)
        opt.zero_grad()
        loss.backward()
        grads.append(image_syn.grad[:, ::2, :, ::4, ::4].clone())
        opt.step()
        losses.append(float(loss)); rel_diff.append(rd); picks.append(np.stack(pk))
        syns.append(image_syn.detach()[:, ::2, :, ::4, ::4].clone())
    npz("g12_dm_late.npz", data_seed=1201, pool_per_class=NP, net_seed0=1210, C=C, ipc=ipc, batch_real=B, steps=steps, lr=lr, momentum=0.5,
        picks=np.stack(picks), losses=np.array(losses), rel_diff=np.array(rel_diff), grads=torch.stack(grads),
        syns=torch.stack(syns), syn_final_abs=float(image_syn.detach().double().abs().sum()),
        grad_l1=np.array([float(gk.double().abs().sum()) for gk in grads]))


def g13(networks, utils):
    # G13: one "MTT+Ours" iteration (distill_s2d_ms.py:189-300) around the reference's ReparamModule, ConvNet3D and
    # Conv3DNet: per student step the batch is composed by the hallucinator from randomly drawn static / dynamic memories,
    # the grand loss is back-propagated to dynamic memories, static memories, hallucinator and syn_lr, and the four
    # SGD optimisers (momentum .95 / .9 for syn_lr, :107-110) take one step.  C=3, vpc=1, spc=2, dpc=2.
    sys.path.insert(0, REF)
    from reparam_module import ReparamModule
    C, vpc, spc, dpc, syn_steps, batch_syn = 3, 1, 2, 2, 3, 2
    net = make_net(networks, 131, C, 64, 8)
    net.dropout.p = 0.0
    starting = [p.detach().clone() for p in net.parameters()]
    g = torch.Generator().manual_seed(1301)
    target = [p + 0.02 * p.abs().mean() * torch.randn(p.shape, generator=g) for p in starting]
    static_syn = torch.randn(C * spc, 3, 64, 64, generator=g).requires_grad_(True)
    dynamic_syn = torch.randn(C, dpc, 8, 1, 64, 64, generator=g).requires_grad_(True)
    torch.manual_seed(1302)
    hals = torch.nn.ModuleList([utils.Conv3DNet(img_size=64)])
    w0, b0 = hals[0].encoder.weight.detach().clone(), hals[0].encoder.bias.detach().clone()
    syn_lr = torch.tensor(0.01).requires_grad_(True)
    lr_static, lr_dynamic, lr_hal, lr_lr = 10.0, 100.0, 0.01, 1e-5
    optimizer_static = torch.optim.SGD([static_syn], lr=lr_static, momentum=0.95)
    optimizer_dynamic = torch.optim.SGD([dynamic_syn], lr=lr_dynamic, momentum=0.95)
    optimizer_hals = torch.optim.SGD(hals.parameters(), lr=lr_hal, momentum=0.95)
    optimizer_lr = torch.optim.SGD([syn_lr], lr=lr_lr, momentum=0.9)
    student_net = ReparamModule(net)
    student_net.train()
    num_params = sum([np.prod(p.size()) for p in (student_net.parameters())])
    target_params = torch.cat([p.reshape(-1) for p in target], 0)
    student_params = [torch.cat([p.reshape(-1) for p in starting], 0).requires_grad_(True)]
    starting_params = torch.cat([p.reshape(-1) for p in starting], 0)
    criterion = torch.nn.CrossEntropyLoss()
    torch.manual_seed(1311)
    indices_chunks, used, draws_d, draws_s = [], [], [], []
    for step in range(syn_steps):
        if not indices_chunks:
            indices = torch.randperm(C * vpc)
            indices_chunks = list(torch.split(indices, batch_syn))
        these_indices = indices_chunks.pop()
        label = these_indices // vpc
        idx = these_indices % vpc
        rd = torch.randint(2, (these_indices.shape[0],))
        dynamic_idx = 2 * idx + rd
        rs = torch.randint(2, (these_indices.shape[0],))
        static_idx = spc * label + 2 * idx + rs
        used.append(these_indices.clone()); draws_d.append(rd.clone()); draws_s.append(rs.clone())
        x = hals[0](static_syn[static_idx, :, :, :], dynamic_syn[label, dynamic_idx, :, :, :, :])
        out = student_net(x, flat_param=student_params[-1])
        loss = criterion(out, label.long())
        grad = torch.autograd.grad(loss, student_params[-1], create_graph=True)[0]
        student_params.append(student_params[-1] - syn_lr * grad)
    param_loss = torch.nn.functional.mse_loss(student_params[-1], target_params, reduction="sum") / num_params
    param_dist = torch.nn.functional.mse_loss(starting_params, target_params, reduction="sum") / num_params
    grand_loss = param_loss / param_dist
    for o in (optimizer_static, optimizer_dynamic, optimizer_hals, optimizer_lr):
        o.zero_grad()
    grand_loss.backward()
    gd, gs = dynamic_syn.grad.clone(), static_syn.grad.clone()
    gw, gb, glr = hals[0].encoder.weight.grad.clone(), hals[0].encoder.bias.grad.clone(), syn_lr.grad.clone()
    for o in (optimizer_static, optimizer_dynamic, optimizer_hals, optimizer_lr):
        o.step()
    syn_lr.data = syn_lr.data.clip(min=0.001)
    pad = lambda lst: np.stack([np.pad(t.numpy(), (0, batch_syn - len(t)), constant_values=-1) for t in lst])   # noqa: E731
    npz("g13_s2d_mtt_step.npz", net_seed=131, data_seed=1301, hal_seed=1302, C=C, vpc=vpc, spc=spc, dpc=dpc, syn_steps=syn_steps,
        batch_syn=batch_syn, indices=pad(used), draws_dyn=pad(draws_d), draws_sta=pad(draws_s), hal_w=w0, hal_b=b0,
        syn_lr=0.01, lr_static=lr_static, lr_dynamic=lr_dynamic, lr_hal=lr_hal, lr_lr=lr_lr,
        grand_loss=grand_loss.detach(), grad_lr=glr, g_hal_w=gw, g_hal_b=gb,
        g_dynamic=gd[:, :, :, :, ::4, ::4], g_dynamic_rowabs=gd.abs().sum(dim=(2, 3, 4, 5)),
        g_static=gs[:, :, ::4, ::4], g_static_rowabs=gs.abs().sum(dim=(1, 2, 3)),
        hal_w_after=hals[0].encoder.weight.detach(), hal_b_after=hals[0].encoder.bias.detach(), syn_lr_after=syn_lr.detach(),
        dynamic_after=dynamic_syn.detach()[:, :, :, :, ::4, ::4], static_after=static_syn.detach()[:, :, ::4, ::4])


def g14(networks, utils):
    # F4 interchange: the on-disk artefacts exactly as the reference's drivers write them -- hal_{it}.pt =
    # ModuleList[Conv3DNet].state_dict() (distill_s2d_ms.py:373), dynamic_{it}.pt = dynamic_syn.flatten(0, 1).cpu() (:364, 374),
    # images_{it}.pt = static / synthetic clips .cpu() (:372, distill_baseline.py:329), the static-memory file a dict with
    # key "image" (:97), replay_buffer_0.pt = list[expert] of list[epoch] of [p.detach().cpu() for p in net.parameters()]
    # (buffer.py:75-104; a narrow ConvNet3D keeps the file small) -- plus an npz with the same values for the tests.
    # The reverse direction is checked right here: files written by video_distillation_amd.checkpoint load into the
    # reference's own modules.
    import tempfile
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from video_distillation_amd import checkpoint
    d = os.path.join(OUT, "f4")
    os.makedirs(d, exist_ok=True)
    torch.manual_seed(1401)
    hals = torch.nn.ModuleList([utils.Conv3DNet(img_size=16) for _ in range(2)])
    g = torch.Generator().manual_seed(1402)
    C, spc, dpc, T, S = 3, 2, 2, 4, 16
    static_syn = torch.randn(C * spc, 3, S, S, generator=g)
    dynamic_syn = torch.randn(C, dpc, T, 1, S, S, generator=g)
    image_syn = torch.randn(C, T, 3, S, S, generator=g)
    torch.save(hals.state_dict(), os.path.join(d, "hal_7.pt"))
    torch.save(dynamic_syn.flatten(0, 1).detach().cpu(), os.path.join(d, "dynamic_7.pt"))
    torch.save(static_syn.detach().cpu(), os.path.join(d, "images_7.pt"))
    torch.save({"image": static_syn.detach().cpu()}, os.path.join(d, "static_memory.pt"))
    torch.save(image_syn.detach().cpu(), os.path.join(d, "images_baseline_7.pt"))
    trajectories = []
    for e in range(1):
        torch.manual_seed(1410 + e)
        net = networks.ConvNet3D(channel=3, num_classes=3, net_width=8, net_depth=3, net_act='relu', net_norm='none',
                                 net_pooling='maxpooling', im_size=(64, 64), frames=8)
        stamps = [[p.detach().cpu() for p in net.parameters()]]
        with torch.no_grad():
            for p in net.parameters():
                p.mul_(0.99)
        stamps.append([p.detach().cpu() for p in net.parameters()])
        trajectories.append(stamps)
    torch.save(trajectories, os.path.join(d, "replay_buffer_0.pt"))
    npz(os.path.join("f4", "values.npz"), hal_w=torch.stack([h.encoder.weight.detach() for h in hals]),
        hal_b=torch.stack([h.encoder.bias.detach() for h in hals]), static=static_syn, dynamic=dynamic_syn, image_syn=image_syn,
        traj_l1=np.array([[float(p.double().abs().sum()) for p in st] for st in trajectories[0]]),
        traj_shapes=np.array([list(p.shape) + [0] * (5 - p.dim()) for p in trajectories[0][0]]))
    # reverse direction: our writers -> the reference's loaders
    with tempfile.TemporaryDirectory() as tmp:
        checkpoint.save_s2d(tmp, 3, dynamic_syn, [h.encoder.weight for h in hals], [h.encoder.bias for h in hals], best=True)
        checkpoint.save_images(tmp, 3, image_syn, best=True)
        checkpoint.save_expert_buffer(tmp, trajectories)
        again = torch.nn.ModuleList([utils.Conv3DNet(img_size=16) for _ in range(2)])
        again.load_state_dict(torch.load(os.path.join(tmp, "hal_3.pt")))                  # strict: same keys
        assert all(torch.equal(a.encoder.weight, b.encoder.weight) for a, b in zip(again, hals))
        again.load_state_dict(torch.load(os.path.join(tmp, "weights_best.pt")))
        assert torch.equal(torch.load(os.path.join(tmp, "dynamic_3.pt")), dynamic_syn.flatten(0, 1))
        assert torch.equal(torch.load(os.path.join(tmp, "images_best.pt")), image_syn)
        buf = torch.load(os.path.join(tmp, "replay_buffer_0.pt"))                          # distill_baseline.py:128
        start = torch.cat([p.data.reshape(-1) for p in buf[0][0]], 0)                        # :216
        assert start.numel() == sum(p.numel() for p in trajectories[0][0])
    print("f4: reference-written files in %s; checkpoint.py's files load into the reference's modules" % d)


def _install_transform_stubs():
    """torchvision is absent here; the reference's dataset classes need ``transforms.Compose / ToTensor / Normalize`` and
    ``transforms.functional.hflip``.  These four are restated from torchvision's documented behaviour (PIL RGB image ->
    CHW float in [0, 1]; per-channel (x - mean) / std in place; left-right mirror) -- they are the fixture generator's,
    not the reference's; what G15 pins is the reference's DATASET logic around them (index files, label numbering, frame
    picks, generator order, flip-before-transform, stacking)."""
    from PIL import Image
    tv = sys.modules["torchvision.transforms"]

    class Compose:
        def __init__(self, ts): self.ts = ts
        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    class ToTensor:
        def __call__(self, pic):
            arr = np.asarray(pic.convert("RGB") if pic.mode != "RGB" else pic, dtype=np.uint8)
            return torch.from_numpy(arr.copy()).permute(2, 0, 1).contiguous().to(torch.float32).div(255)

    class Normalize:
        def __init__(self, mean, std): self.mean, self.std = mean, std
        def __call__(self, t):
            mean = torch.as_tensor(self.mean, dtype=t.dtype).view(-1, 1, 1)
            std = torch.as_tensor(self.std, dtype=t.dtype).view(-1, 1, 1)
            return t.clone().sub_(mean).div_(std)

    fn = types.ModuleType("torchvision.transforms.functional")
    fn.hflip = lambda img: img.transpose(Image.FLIP_LEFT_RIGHT)
    tv.Compose, tv.ToTensor, tv.Normalize, tv.functional = Compose, ToTensor, Normalize, fn
    sys.modules["torchvision.transforms.functional"] = fn
    return tv


def _write_frame_tree(root):
    """A tiny UCF-style tree (5 videos, 33..64 frames of 112x112 JPEG) and a Kinetics-style one (3 videos x 8 frames 64x64)."""
    from PIL import Image
    rng = np.random.default_rng(2024)
    yy, xx = np.mgrid[0:112, 0:112].astype(np.float32)

    def frame(vid, t, size):
        s = 112 // size
        base = np.stack([(xx * (1 + vid) + 3 * t) % 256, (yy * 2 + 5 * vid) % 256, ((xx + yy) * 0.7 + 11 * t) % 256], -1)
        x0 = (7 * t + 13 * vid) % 80
        base[20 + vid * 5:50 + vid * 5, x0:x0 + 24] = (250 - 40 * vid, 30 + 20 * vid, 120)
        base = base + rng.normal(0, 4, base.shape)
        return Image.fromarray(np.clip(base, 0, 255).astype(np.uint8)[::s, ::s])
    ucf = os.path.join(root, "UCF101")
    vids = [("v_Archery_g01_c01", "Archery", "train", 64), ("v_Biking_g01_c01", "Biking", "train", 40),
            ("v_Archery_g02_c01", "Archery", "train", 36), ("v_Biking_g02_c01", "Biking", "test", 33),
            ("v_Archery_g03_c01", "Archery", "test", 34)]
    for vi, (name, _, _, n) in enumerate(vids):
        d = os.path.join(ucf, "jpegs_112", name)
        os.makedirs(d, exist_ok=True)
        for t in range(1, n + 1):
            frame(vi, t, 112).save(os.path.join(d, "frame%06d.jpg" % t), quality=60)
    for csv_name in ("ucf101_splits1.csv", "ucf50_splits1.csv", "hmdb51_splits.csv", "hmdb25_splits.csv"):
        with open(os.path.join(ucf, csv_name), "w") as fp:
            fp.write("folder_name,label,split\n")
            for name, label, split, _ in vids:
                fp.write("%s,%s,%s\n" % (name, label, split))
    kin = os.path.join(root, "kinetics_64x64x8")
    kvids = [("aaaaaaaaaaa", 3, 13, "zumba", "train"), ("bbbbbbbbbbb", 10, 20, "abseiling", "train"),
             ("ccccccccccc", 0, 10, "zumba", "train"), ("ddddddddddd", 5, 15, "abseiling", "val")]
    for vi, (yt, a, b, label, split) in enumerate(kvids):
        d = os.path.join(kin, split, "%s_%06d_%06d" % (yt, a, b))
        os.makedirs(d, exist_ok=True)
        for t in range(8 if vi != 2 else 5):               # the third clip is short: the reader must skip it
            frame(vi, t, 56).resize((64, 64)).save(os.path.join(d, "img_%05d.jpg" % t), quality=60)
    for split, csv_split in (("train", "train"), ("val", "validate")):
        with open(os.path.join(kin, "%s.csv" % csv_split), "w") as fp:
            fp.write("label,youtube_id,time_start,time_end,split\n")
            for yt, a, b, label, sp in kvids:
                if sp == split:
                    fp.write("%s,%s,%d,%d,%s\n" % (label, yt, a, b, csv_split))


def g15():
    """Frame-folder datasets (distill_utils/dataset.py): a committed JPEG tree + what the reference's classes return for it
    under fixed generator seeds.  Per item: the frame numbers picked, the label, a strided probe of the clip and two
    whole-clip checksums."""
    root = os.path.join(OUT, "frames")
    if not os.path.exists(os.path.join(root, "UCF101")):
        _write_frame_tree(root)
    tv = _install_transform_stubs()
    sys.path.insert(0, REF)
    from distill_utils import dataset as RD
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    tf = tv.Compose([tv.ToTensor(), tv.Normalize(mean=mean, std=std)])
    out = {}

    def record(tag, ds, passes):
        np.random.seed(5); random.seed(7); torch.manual_seed(3)
        k = 0
        for _ in range(passes):
            for i in range(len(ds)):
                x, y = ds[i]
                out["%s_%d_probe" % (tag, k)] = x[:, :, ::16, ::16].numpy()
                out["%s_%d_sums" % (tag, k)] = np.array([float(x.double().sum()), float((x.double() ** 2).sum())])
                out["%s_%d_label" % (tag, k)] = np.int64(y)
                if hasattr(ds, "frames"):
                    out["%s_%d_frames" % (tag, k)] = np.array(ds.frames, dtype=np.int64)
                k += 1
        out["%s_count" % tag] = np.int64(k)
        out["%s_labels" % tag] = np.array(ds.labels, dtype=np.int64)
    ucf = os.path.join(root, "UCF101")
    record("ucf_train", RD.UCF101(ucf, "train", tf), 2)          # second pass: kept start frames, fresh flips
    record("ucf_test", RD.UCF101(ucf, "test", tf), 2)            # test items redraw the start every visit
    record("hmdb_train", RD.HMDB51(ucf, "train", tf), 1)
    record("minihmdb_train", RD.miniHMDB51(ucf, "train", tf), 1)
    record("mini_train", RD.miniUCF101(ucf, "train", tf), 1)
    record("mini_seg", RD.miniUCF101(ucf, "train", tf, sample="split-random"), 1)
    kin = os.path.join(root, "kinetics_64x64x8")
    for split in ("train", "val"):
        ds = RD.Kinetics400(kin, split, tf)
        out["kin_%s_labels" % split] = np.array(ds.labels, dtype=np.int64)
        out["kin_%s_dirs" % split] = np.array([os.path.basename(d) for d in ds.video_dirs])
        for i in range(len(ds)):
            x, y = ds[i]
            names = os.listdir(ds.video_dirs[i])               # the order the reference stacked the frames in
            out["kin_%s_%d_names" % (split, i)] = np.array(names)
            out["kin_%s_%d_probe" % (split, i)] = x[:, :, ::8, ::8].numpy()
            out["kin_%s_%d_sums" % (split, i)] = np.stack([x[t].double().sum().numpy() for t in range(x.shape[0])])
    npz("g15_frame_datasets.npz", **out)


def _write_still_extras(root):
    """What the still-frame families need on top of G15's tree: the UCF50 index with per-video cut points, and an
    SSv2-style frame set (2 train + 1 val videos x 8 frames 64x64, one more that is short)."""
    from PIL import Image
    ucf = os.path.join(root, "UCF101")
    cuts = {"v_Archery_g01_c01": [40, 12, 25], "v_Biking_g01_c01": [9, 20, 31], "v_Archery_g02_c01": [8, 17, 27],
            "v_Biking_g02_c01": [7, 15, 24], "v_Archery_g03_c01": [26, 6, 16]}          # unsorted on purpose: the reader sorts
    with open(os.path.join(ucf, "ucf50_splits1.csv")) as fp, open(os.path.join(ucf, "ucf50_splits1_max.csv"), "w") as out:
        rows = fp.read().strip().split("\n")
        out.write(rows[0] + ",split_index\n")
        for r in rows[1:]:
            out.write('%s,"[%s]"\n' % (r, ", ".join(str(c) for c in cuts[r.split(",")[0]])))
    ss = os.path.join(root, "SSv2_64x8")
    rng = np.random.default_rng(77)
    yy, xx = np.mgrid[0:64, 0:64].astype(np.float32)
    items = [("101", "Pushing something", "train", 8), ("102", "Dropping something", "train", 8), ("103", "Pushing something", "train", 6),
             ("201", "Dropping something", "val", 8)]
    for vi, (vid, _, _, n) in enumerate(items):
        d = os.path.join(ss, "frame", vid)
        os.makedirs(d, exist_ok=True)
        for t in range(n):
            img = np.stack([(xx * (2 + vi) + 9 * t) % 256, (yy * 3 + 17 * vi) % 256, ((xx - yy) * 1.3 + 7 * t) % 256], -1)
            img = img + rng.normal(0, 3, img.shape)
            Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(os.path.join(d, "%04d.jpg" % (t + 1)), quality=60)
    for split in ("train", "val"):
        with open(os.path.join(ss, "annot_%s.json" % split), "w") as fp:
            json.dump([{"id": vid, "class": c} for vid, c, sp, _ in items if sp == split], fp)


def g17():
    """Still-frame datasets (distill_utils/dataset.py staticHMDB51 / staticUCF101 / staticUCF50 / singleKinetics400 /
    singleSSv2, and the SSv2 video class): what the reference's classes return for the committed tree under fixed seeds."""
    root = os.path.join(OUT, "frames")
    if not os.path.exists(os.path.join(root, "UCF101")):
        _write_frame_tree(root)
    if not os.path.exists(os.path.join(root, "SSv2_64x8")):
        _write_still_extras(root)
    tv = _install_transform_stubs()
    sys.path.insert(0, REF)
    from distill_utils import dataset as RD
    tf = tv.Compose([tv.ToTensor(), tv.Normalize(mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225])])
    out = {}
    ucf = os.path.join(root, "UCF101")

    def record(tag, ds, passes=2):
        np.random.seed(5); random.seed(7); torch.manual_seed(3)
        k = 0
        for _ in range(passes):
            for i in range(len(ds)):
                x, y = ds[i]
                out["%s_%d_probe" % (tag, k)] = x[..., ::16, ::16].numpy()
                out["%s_%d_sums" % (tag, k)] = np.array([float(x.double().sum()), float((x.double() ** 2).sum())])
                out["%s_%d_label" % (tag, k)] = np.int64(y)
                out["%s_%d_frame" % (tag, k)] = np.int64(ds.start)
                k += 1
        out["%s_count" % tag] = np.int64(k)
        out["%s_labels" % tag] = np.array(ds.labels, dtype=np.int64)
    with contextlib.redirect_stdout(io.StringIO()):
        record("shmdb_train", RD.staticHMDB51(ucf, "train", tf))
        record("shmdb_test_image", RD.staticHMDB51(ucf, "test", tf, frames=1))
        record("sucf_train", RD.staticUCF101(ucf, "train", tf))
        record("sucf_part1of3", RD.staticUCF101(ucf, "train", tf, frames=4, split_num=3, split_id=1))
        record("sucf_part_wraps", RD.staticUCF101(ucf, "test", tf, frames=1, split_num=2, split_id=2))      # split_id >= split_num -> 0
        record("s50_mean", RD.staticUCF50(ucf, "train", tf, frames=2, split_num=4, split_id=3, split_mode='mean'))
        for sid in range(4):
            record("s50_feature%d" % sid, RD.staticUCF50(ucf, "train", tf, frames=1, split_num=4, split_id=sid, split_mode='feature'))
        for tag, make, path in (("skin", RD.singleKinetics400, os.path.join(root, "kinetics_64x64x8")),
                                ("sssv2", RD.singleSSv2, os.path.join(root, "SSv2_64x8"))):
            for split in ("train", "val"):
                ds = make(path, split, tf)
                out["%s_%s_labels" % (tag, split)] = np.array(ds.labels, dtype=np.int64)
                out["%s_%s_dirs" % (tag, split)] = np.array([os.path.basename(d) for d in ds.video_dirs])
                random.seed(7)
                picks = []
                for rep in range(3):
                    for i in range(len(ds)):
                        state = random.getstate()
                        x, y = ds[i]
                        after = random.getstate()
                        random.setstate(state)
                        idx = random.randint(0, 7)                      # the one draw an item makes
                        assert random.getstate() == after
                        name = os.listdir(ds.video_dirs[i])[idx]
                        want = tf(__import__("PIL.Image").Image.open(os.path.join(ds.video_dirs[i], name)))
                        assert torch.equal(x, want) and y == ds.labels[i]
                        picks.append(idx)
                out["%s_%s_picks" % (tag, split)] = np.array(picks, dtype=np.int64)
        # every frame of the SSv2 tree by name (the single* checks look pixels up here), and the SSv2 video class
        for split in ("train", "val"):
            ds = RD.SSv2(os.path.join(root, "SSv2_64x8"), split, tf)
            out["ssv2_%s_labels" % split] = np.array(ds.labels, dtype=np.int64)
            out["ssv2_%s_dirs" % split] = np.array([os.path.basename(d) for d in ds.video_dirs])
            for i in range(len(ds)):
                x, y = ds[i]
                out["ssv2_%s_%d_names" % (split, i)] = np.array(os.listdir(ds.video_dirs[i]))
                out["ssv2_%s_%d_probe" % (split, i)] = x[:, :, ::8, ::8].numpy()
                out["ssv2_%s_%d_sums" % (split, i)] = np.stack([x[t].double().sum().numpy() for t in range(x.shape[0])])
    npz("g17_still_datasets.npz", **out)


def g16(networks, utils):
    """The metric's accuracy half at the benchmark's scale: the REFERENCE's evaluate_synset (utils.py:848-886, imported) on the
    learnable 50-class problem of tests/synth_problem.py -- C=50, IPC=1, 64x64x8, epoch_eval_train=100, five fixed network
    seeds -- once with dropout off (the HIP run can follow it epoch by epoch: same initial weights, same data, the batch is
    the whole set) and once with the reference's dropout 0.5 (masks come from the CPU generator here and from the device
    generator on the HIP path: comparable only statistically).  Records every epoch's training loss / accuracy and the final
    3-pass test accuracy per seed."""
    sys.path.insert(0, os.path.dirname(OUT.rstrip("/")).rsplit("/tests", 1)[0])
    from tests.synth_problem import template_problem, checksum
    C, T, S, epochs, seeds = 50, 8, 64, 100, [1000, 1001, 1002, 1003, 1004]
    noise_test = float(os.environ.get("VD_G16_NOISE_TEST", "7.0"))
    train_x, train_y, test_x, test_y = template_problem(C, T, S, n_test=8, noise_train=1.0, noise_test=noise_test)
    args = _Args()
    args.device = 'cpu'; args.lr_net = 0.01; args.epoch_eval_train = epochs; args.batch_train = 256
    args.model = 'ConvNet3D'; args.eval_mode = 'SS'
    testloader = torch.utils.data.DataLoader(utils.TensorDataset(test_x, test_y), batch_size=64, shuffle=False)
    out = {}
    orig_epoch = utils.epoch
    for tag, p_drop in (("p0", 0.0), ("p5", 0.5)):
        curves, accs, tests, tlosses, ws = [], [], [], [], []
        for sd in seeds:
            net = make_net(networks, sd, C, S, T)
            net.dropout.p = p_drop
            torch.manual_seed(sd + 7); random.seed(sd + 7); np.random.seed(sd + 7)
            rec = []

            def spy(mode, loader, net_, opt, crit, a):
                o = orig_epoch(mode, loader, net_, opt, crit, a)
                rec.append((mode, o[0], o[1]))
                return o
            utils.epoch = spy
            try:
                net_out, acc_train, acc_test, _ = utils.evaluate_synset(0, net, train_x, train_y, testloader, args, mode='none')
            finally:
                utils.epoch = orig_epoch
            tr = [(l, a) for m, l, a in rec if m == 'train']
            te = [(l, a) for m, l, a in rec if m == 'test']
            curves.append([t[0] for t in tr]); accs.append([t[1] for t in tr]); tests.append(acc_test); tlosses.append(te[-1][0])
            ws.append(net_out.logit.weight.detach().reshape(C, -1)[:4, :8].numpy().copy())
            print("g16 %s seed %d: loss %.4f -> %.4f, train acc %.2f, test top-1 %.3f" % (tag, sd, tr[0][0], tr[-1][0], acc_train, acc_test))
        out.update({tag + "_train_loss": np.array(curves), tag + "_train_acc": np.array(accs), tag + "_test_acc": np.array(tests),
                    tag + "_test_loss": np.array(tlosses), tag + "_logit_w": np.array(ws)})
    npz("g16_eval_c50.npz", C=C, T=T, S=S, epochs=epochs, seeds=np.array(seeds), lr_net=0.01, n_test=8, noise_train=1.0,
        noise_test=noise_test, problem_seed=1606, train_checksum=np.array(checksum(train_x)), test_checksum=np.array(checksum(test_x)), **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    networks, utils = import_reference()
    only = set(sys.argv[1:])
    for name, fn in (("g1", lambda: g1(networks)), ("g2", lambda: g2_g3(networks)), ("g4", lambda: g4_g5(networks, utils)),
                     ("g6", lambda: g6(networks, utils)), ("g7", lambda: g7(networks, utils)),
                     ("g9", lambda: g9(networks, utils)), ("g10", lambda: g10(networks, utils)),
                     ("g11", lambda: g11(networks, utils)), ("g12", lambda: g12(networks)), ("g13", lambda: g13(networks, utils)), ("g14", lambda: g14(networks, utils)), ("g15", g15), ("g17", g17),
                     ("g16", lambda: g16(networks, utils))):
        if not only or name in only:
            fn()


if __name__ == "__main__":
    main()
