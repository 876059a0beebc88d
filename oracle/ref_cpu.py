"""CPU oracle for the distribution/gradient-matching hot path (TEST INFRASTRUCTURE ONLY).

This file restates, in plain fp32 ``torch`` CPU functional ops, what the reference
(yuz1wan/video_distillation) computes on the path named in BASELINE.json:north_star.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it; the product package ``video_distillation_amd`` never does.

Parity pin: every function below is checked in ``tests/test_oracle_golden.py`` against
fixtures in ``tests/golden/*.npz`` that were produced by importing the reference itself
in the build container (``tools/gen_golden.py``; torch 2.10 CPU / MKL-DNN fp32).  The
reference has no tests or golden vectors of its own, so that is the pin.

Reference call sites restated here (paths under /root/reference):
  * ConvNet3D geometry, ``embed`` and ``forward``  -- networks.py:727-814
  * default ConvNet3D settings used by get_network -- utils.py:512-514, 608-609
  * DM class term                                  -- distill_baseline.py:344-351
  * pixel SGD(momentum) update                     -- distill_baseline.py:107, 353-355
  * hallucinator ``Conv3DNet``                     -- utils.py:1178-1197
  * s2d index composition                          -- distill_s2d_ms.py:402-411
  * ``distance_wb`` / ``match_loss``               -- utils.py:634-687
  * ``epoch`` / ``evaluate_synset``                -- utils.py:752-886
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

CONV_KERNEL = (3, 7, 7)
CONV_STRIDE = (1, 2, 2)
CONV_PAD = (1, 3, 3)
# (out_channels, pool kernel) per depth level; width 128, depth 3 (utils.py:512, 608)
LAYER_SPECS = ((64, (1, 2, 2)), (128, (2, 2, 2)), (128, (2, 2, 2)))
PARAM_NAMES = (
    "features.0.weight", "features.0.bias",
    "features.3.weight", "features.3.bias",
    "features.6.weight", "features.6.bias",
    "logit.weight", "logit.bias",
)


def param_shapes(channel: int = 3, num_classes: int = 50) -> List[Tuple[int, ...]]:
    """Shapes of ConvNet3D parameters in ``net.parameters()`` order (networks.py:792-814)."""
    shapes: List[Tuple[int, ...]] = []
    cin = channel
    for cout, _ in LAYER_SPECS:
        shapes.append((cout, cin) + CONV_KERNEL)
        shapes.append((cout,))
        cin = cout
    shapes.append((num_classes, cin, 1, 1, 1))
    shapes.append((num_classes,))
    return shapes


def init_params(seed: int, channel: int = 3, num_classes: int = 50) -> List[torch.Tensor]:
    """PyTorch-default Conv3d init (kaiming-uniform a=sqrt(5); bias U(+-1/sqrt(fan_in))),
    drawn in the same order and with the same RNG calls as ``nn.Conv3d.reset_parameters``
    so that ``torch.manual_seed(seed); ConvNet3D(...)`` in the reference yields identical
    tensors (checked against fixture G1)."""
    gen_state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    out: List[torch.Tensor] = []
    try:
        shapes = param_shapes(channel, num_classes)
        for wi in range(0, len(shapes), 2):
            wshape = shapes[wi]
            fan_in = int(np.prod(wshape[1:]))
            w = torch.empty(wshape)
            # kaiming_uniform_(a=sqrt(5)) == U(-1/sqrt(fan_in), 1/sqrt(fan_in))
            gain = math.sqrt(2.0 / (1 + 5.0))
            bound_w = gain * math.sqrt(3.0 / fan_in)
            w.uniform_(-bound_w, bound_w)
            b = torch.empty(shapes[wi + 1])
            bound_b = 1.0 / math.sqrt(fan_in)
            b.uniform_(-bound_b, bound_b)
            out += [w, b]
    finally:
        torch.random.set_rng_state(gen_state)
    return out


def feature_layers(x_bcthw: torch.Tensor, params: Sequence[torch.Tensor],
                   collect: Optional[list] = None) -> torch.Tensor:
    """``self.features`` of ConvNet3D: 3 x [Conv3d -> ReLU -> MaxPool3d] (networks.py:792-814)."""
    out = x_bcthw
    for li, (_, pool) in enumerate(LAYER_SPECS):
        out = F.conv3d(out, params[2 * li], params[2 * li + 1], stride=CONV_STRIDE, padding=CONV_PAD)
        if collect is not None:
            collect.append(out.clone())
        out = torch.relu(out)
        if collect is not None:
            collect.append(out.clone())
        out = F.max_pool3d(out, kernel_size=pool, stride=pool)
        if collect is not None:
            collect.append(out.clone())
    return out


def convnet3d_embed(x_btchw: torch.Tensor, params: Sequence[torch.Tensor]) -> torch.Tensor:
    """``ConvNet3D.embed`` (networks.py:747-751): (B,T,C,H,W) -> (B, 128*T'*H'*W')."""
    feat = feature_layers(x_btchw.permute(0, 2, 1, 3, 4), params)
    return feat.reshape(feat.shape[0], -1)


def convnet3d_logits(x_btchw: torch.Tensor, params: Sequence[torch.Tensor],
                     drop_mask: Optional[torch.Tensor] = None, training: bool = False,
                     p_drop: float = 0.5) -> torch.Tensor:
    """``ConvNet3D.forward`` (networks.py:738-745): features -> AvgPool3d -> Dropout ->
    1x1x1 conv -> squeeze spatial -> max over T.  AvgPool kernel is (2,2,2) stride 1 when the
    clip is taller than 64 px, otherwise (2,1,1) (networks.py:733)."""
    feat = feature_layers(x_btchw.permute(0, 2, 1, 3, 4), params)
    big = x_btchw.shape[-2] > 64
    feat = F.avg_pool3d(feat, kernel_size=(2, 2, 2) if big else (2, 1, 1), stride=1)
    if drop_mask is not None:
        feat = feat * drop_mask
    elif training:
        feat = F.dropout(feat, p=p_drop, training=True)
    out = F.conv3d(feat, params[6], params[7])
    out = out.squeeze(3).squeeze(3)
    return out.max(dim=2).values


def dm_class_term(feat_real: torch.Tensor, feat_syn: torch.Tensor) -> torch.Tensor:
    """One summand of the DM loss (distill_baseline.py:351): squared L2 distance between
    batch-mean embeddings; the real side carries no gradient."""
    diff = feat_real.detach().mean(dim=0) - feat_syn.mean(dim=0)
    return (diff * diff).sum()


def dm_loss_and_grad(params: Sequence[torch.Tensor], real_per_class: Sequence[torch.Tensor],
                     syn: torch.Tensor, ipc: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Loss of one DM iteration over all classes and d loss / d syn
    (distill_baseline.py:343-354; parameters frozen :336-337)."""
    syn = syn.detach().clone().requires_grad_(True)
    loss = torch.zeros(())
    for c, real in enumerate(real_per_class):
        f_real = convnet3d_embed(real, params).detach()
        f_syn = convnet3d_embed(syn[c * ipc:(c + 1) * ipc], params)
        loss = loss + dm_class_term(f_real, f_syn)
    (grad,) = torch.autograd.grad(loss, syn)
    return loss.detach(), grad


def sgd_momentum_step(x: torch.Tensor, grad: torch.Tensor, buf: Optional[torch.Tensor],
                      lr: float, momentum: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """``torch.optim.SGD`` with momentum, dampening 0, no weight decay
    (distill_baseline.py:107): first step buf=g, afterwards buf=mu*buf+g; x -= lr*buf."""
    buf = grad.clone() if buf is None else buf * momentum + grad
    return x - lr * buf, buf


def hallucinator(static: torch.Tensor, dynamic: torch.Tensor, weight: torch.Tensor,
                 bias: torch.Tensor) -> torch.Tensor:
    """``Conv3DNet.forward`` in 'concat' mode (utils.py:1186-1197).  static (n,3,H,W) is
    broadcast over the T frames of dynamic (n,T,1,H,W); the 4-channel volume goes through
    Conv3d(4->3, k=3, pad=1); result is returned as (n,T,3,H,W)."""
    n, t = dynamic.shape[0], dynamic.shape[1]
    vol_static = static.unsqueeze(2).expand(n, static.shape[1], t, *static.shape[2:])
    vol_dynamic = dynamic.transpose(1, 2)
    vol = torch.cat([vol_static, vol_dynamic], dim=1)
    return F.conv3d(vol, weight, bias, padding=1).transpose(1, 2)


def s2d_indices(num_classes: int, vpc: int, spc: int, draws_dynamic: torch.Tensor,
                draws_static: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Index composition of the s2d DM step (distill_s2d_ms.py:402-406).  ``draws_*`` are
    the two ``randint(2, (C*vpc,))`` tensors (injected so tests can pin them)."""
    label = torch.arange(num_classes).repeat_interleave(vpc)
    idx = torch.arange(num_classes * vpc) % vpc
    dynamic_idx = 2 * idx + draws_dynamic
    static_idx = spc * label + 2 * idx + draws_static
    return label, dynamic_idx, static_idx


def _rowwise_cosine_distance(gwr: torch.Tensor, gws: torch.Tensor) -> torch.Tensor:
    num = (gwr * gws).sum(dim=-1)
    den = gwr.norm(dim=-1) * gws.norm(dim=-1) + 0.000001
    return (1 - num / den).sum()


def distance_wb(gwr: torch.Tensor, gws: torch.Tensor) -> torch.Tensor:
    """Per-layer 'ours' distance (utils.py:634-651).  4-D and 3-D tensors are flattened to
    (dim0, rest); 2-D used as is; 1-D contributes exactly 0.  Anything else -- notably the
    5-D Conv3d weight gradients of ConvNet3D -- falls through un-reshaped, so the cosine is
    taken over the LAST axis only."""
    nd = gwr.dim()
    if nd == 1:
        return torch.zeros((), dtype=torch.float)
    if nd in (3, 4):
        gwr = gwr.reshape(gwr.shape[0], -1)
        gws = gws.reshape(gws.shape[0], -1)
    return _rowwise_cosine_distance(gwr, gws)


def match_loss(gw_syn: Sequence[torch.Tensor], gw_real: Sequence[torch.Tensor],
               dis_metric: str) -> torch.Tensor:
    """``match_loss`` (utils.py:655-687) for the three metrics."""
    if dis_metric == "ours":
        total = torch.zeros(())
        for gr, gs in zip(gw_real, gw_syn):
            total = total + distance_wb(gr, gs)
        return total
    flat_r = torch.cat([g.reshape(-1) for g in gw_real])
    flat_s = torch.cat([g.reshape(-1) for g in gw_syn])
    if dis_metric == "mse":
        return ((flat_s - flat_r) ** 2).sum()
    if dis_metric == "cos":
        return 1 - (flat_r * flat_s).sum() / (flat_r.norm() * flat_s.norm() + 0.000001)
    raise ValueError("unknown distance function: %s" % dis_metric)


def standardise_batch(img: torch.Tensor) -> torch.Tensor:
    """Batch-global scalar standardisation used by ``epoch`` (utils.py:770): unbiased std."""
    return (img - img.mean()) / img.std()


def train_epochs(params: List[torch.Tensor], images: torch.Tensor, labels: torch.Tensor,
                 lr: float, epochs: int, batch_order: Sequence[Sequence[int]],
                 momentum: float = 0.9, weight_decay: float = 0.0005) -> Dict[str, list]:
    """Training half of ``evaluate_synset`` with dropout disabled (fixture G7): SGD(m=.9,
    wd=5e-4), CrossEntropy, ``Epoch+1`` passes, lr *= 0.1 and a FRESH optimiser (momentum
    reset) after epoch ``Epoch//2+1`` (utils.py:848-877).  ``batch_order[ep]`` lists the
    sample indices of each epoch's single shuffled pass (batch = whole set here)."""
    params = [p.detach().clone().requires_grad_(True) for p in params]
    bufs: List[Optional[torch.Tensor]] = [None] * len(params)
    losses, accs, lrs = [], [], []
    for ep in range(epochs + 1):
        order = torch.as_tensor(batch_order[ep])
        img = standardise_batch(images[order].float())
        lab = labels[order]
        logits = convnet3d_logits(img, params, training=False)
        loss = F.cross_entropy(logits, lab)
        grads = torch.autograd.grad(loss, params)
        with torch.no_grad():
            for i, (p, g) in enumerate(zip(params, grads)):
                g = g + weight_decay * p
                bufs[i] = g.clone() if bufs[i] is None else bufs[i] * momentum + g
                p -= lr * bufs[i]
        losses.append(float(loss))
        accs.append(float((logits.argmax(dim=1) == lab).float().mean()))
        lrs.append(lr)
        if ep == epochs // 2 + 1:
            lr *= 0.1
            bufs = [None] * len(params)
    return {"loss": losses, "acc": accs, "lr": lrs, "params": [p.detach() for p in params]}


def flatten_params(params: Sequence[torch.Tensor]) -> torch.Tensor:
    """``torch.cat([p.reshape(-1) for p in params])`` -- ReparamModule's flat parameter
    (reparam_module.py:51), parameters() order."""
    return torch.cat([p.reshape(-1) for p in params], 0)


def unflatten_params(flat: torch.Tensor, channel: int = 3, num_classes: int = 50) -> List[torch.Tensor]:
    out, o = [], 0
    for shp in param_shapes(channel, num_classes):
        n = int(np.prod(shp))
        out.append(flat[o:o + n].view(shp))
        o += n
    return out


def mtt_step(start: Sequence[torch.Tensor], target: Sequence[torch.Tensor], image_syn: torch.Tensor,
             label_syn: torch.Tensor, syn_lr: float, index_chunks: Sequence[torch.Tensor],
             drop_masks: Optional[Sequence[torch.Tensor]] = None, dtype=torch.float32):
    """One MTT iteration (distill_baseline.py:213-262): ``len(index_chunks)`` unrolled student steps
    theta <- theta - syn_lr * dCE/dtheta on the given synthetic batches (create_graph), then
    grand_loss = |theta_N - target|^2 / |theta_0 - target|^2; returns (grand_loss, d/d image_syn,
    d/d syn_lr)."""
    num_classes = start[6].shape[0]
    x = image_syn.detach().to(dtype).clone().requires_grad_(True)
    lr = torch.tensor(float(syn_lr), dtype=dtype, requires_grad=True)
    theta0 = flatten_params([p.detach().to(dtype) for p in start])
    tgt = flatten_params([p.detach().to(dtype) for p in target])
    theta = theta0.clone().requires_grad_(True)
    for s, idx in enumerate(index_chunks):
        m = None if drop_masks is None else drop_masks[s].to(dtype)[:, :, :, None, None]
        logits = convnet3d_logits(x[idx], unflatten_params(theta, 3, num_classes), drop_mask=m)
        ce = F.cross_entropy(logits, label_syn[idx])
        (g,) = torch.autograd.grad(ce, theta, create_graph=True)
        theta = theta - lr * g
    grand = ((theta - tgt) ** 2).sum() / ((theta0 - tgt) ** 2).sum()
    gx, glr = torch.autograd.grad(grand, [x, lr])
    return grand.detach(), gx, glr


def dm_step_flops(num_classes: int, batch_real: int, ipc: int, frames: int, h: int, w: int) -> float:
    """Algorithmic FLOPs of one DM step as BASELINE.md section 3 counts them: every tap incl.
    zero padding, forward for every clip plus one input-gradient pass per synthetic clip."""
    macs = 0
    cin, t, hh, ww = 3, frames, h, w
    for cout, pool in LAYER_SPECS:
        ho, wo = (hh + 6 - 7) // 2 + 1, (ww + 6 - 7) // 2 + 1
        macs += cout * t * ho * wo * cin * 3 * 7 * 7
        t, hh, ww, cin = t // pool[0], ho // pool[1], wo // pool[2], cout
    passes = num_classes * (batch_real + ipc) + num_classes * ipc
    return 2.0 * macs * passes
