"""ConvNet3D with the reference's module surface, executing ``embed`` on the HIP path.

Mirrors ``networks.ConvNet3D`` of the reference (networks.py:727-814): same constructor
arguments, the same sub-module names (``features.{0,3,6}``, ``avg_pool``, ``dropout``,
``logit``) and therefore the same ``state_dict`` keys and ``parameters()`` order that MTT
expert buffers rely on (buffer.py:75, 89).

Execution
  * ``embed(x)`` with frozen parameters on a HIP device -- the DM hot path
    (distill_baseline.py:336-349) -- runs the hand-written MFMA kernels through
    ``EmbedEngine``; the gradient w.r.t. ``x`` comes from the HIP input-gradient passes.
    Clips that carry no gradient (the real batches) use the fast operand precision, clips
    that do (the synthetic ones) the split precision (see ``set_precision``).
  * ``hip_train_step`` -- one ``epoch('train')`` iteration of ``evaluate_synset`` (forward, CE
    loss, all parameter gradients, SGD with momentum / weight decay) on the HIP path
    (train.TrainEngine: weight gradients are tile programs of the same MFMA kernel).
  * ``net(x)`` / ``embed(x)`` with parameters that require a gradient are autograd Functions over the same HIP
    passes, differentiable TWICE: ``torch.autograd.grad(criterion(net(x), y), params, create_graph=True)`` -- the
    reference's DC / MTT call (distill_baseline.py:250) -- and a ``backward()`` through a loss on those gradients run
    train.GradMatchEngine's first- and second-order passes (``_FeatFunction`` / ``_HeadFunction``).
  * ``ReparamModule`` -- the flat-parameter wrapper of reparam_module.py (``forward(x, flat_param=...)``).
There is no CPU and no torch-op (eager) compute path: CPU tensors and non-default architectures raise.
"""
from __future__ import annotations

import os
from typing import Dict, Tuple

import torch
import torch.nn as nn

from . import plan as P

_PRECISION = {"real": os.environ.get("VD_PREC_REAL", "f16"), "syn": os.environ.get("VD_PREC_SYN", "f16x3"),
              "bwd": os.environ.get("VD_PREC_BWD", "f16x3"), "real_last": {"c8": "x3"}.get(os.environ.get("VD_REAL_LAST", "x3"), os.environ.get("VD_REAL_LAST", "x3")),     # (module path: c8 -> x3)
              "train": os.environ.get("VD_PREC_TRAIN", "f16x3"), "train_bwd": os.environ.get("VD_PREC_TRAIN_BWD", "f16x3"),
              "match": os.environ.get("VD_PREC_MATCH", "f16x3"), "match_real_bwd": os.environ.get("VD_PREC_MATCH_REAL_BWD", "f16")}
_ENGINES: Dict[Tuple, object] = {}


def set_precision(real: str = None, syn: str = None, bwd: str = None, train: str = None, train_bwd: str = None,
                  match: str = None, match_real_bwd: str = None, real_last: str = None) -> None:
    """Operand precision of the MFMA contraction: ``real`` for the forward of inputs without
    gradient, ``syn`` for the forward of inputs that need d/dx (its arg-max decisions steer the
    gradient), ``bwd`` for the input-gradient passes (no discrete decisions: single-pass fp16 with
    per-layer power-of-two scaling is the default); ``train`` / ``train_bwd`` for the forward and
    the gradient passes of ``hip_train_step``; ``match`` for the twice-differentiable passes of gradient / trajectory
    matching and ``match_real_bwd`` for the gradient passes of ``param_grads(create_graph=False)`` -- the DETACHED real-batch
    side of gradient matching, whose forward stays in ``train`` precision (its arg-max decisions route the gradient) while
    the backward, which decides nothing, runs single-pass with the power-of-two scaling.  One of 'bf16', 'f16', 'bf16x3',
    'f16x3'."""
    from . import hip
    if real_last is not None:       # "x3": the last conv level of single-pass ``real`` forwards of the DM loop runs in hi+lo pairs
        if real_last not in ("x1", "x3"):
            raise ValueError("real_last is 'x1' or 'x3'")
        _PRECISION["real_last"] = real_last
    for k, v in (("real", real), ("syn", syn), ("bwd", bwd), ("train", train), ("train_bwd", train_bwd), ("match", match),
                 ("match_real_bwd", match_real_bwd)):
        if v is not None:
            if v not in hip.PREC:
                raise ValueError("unknown precision %r" % v)
            _PRECISION[k] = v


def _batch_hint(b: int):
    """Clips-per-launch bucket an engine's programs are planned for (plan.latency_variant): the next power of two
    up to 512, None (throughput-oriented programs) beyond."""
    b = int(b)
    if b > 512:
        return None
    h = 1
    while h < b:
        h *= 2
    return h


def _weight_format():
    """Mixed mode (real single-pass f16/bf16, synthetic clips the hi+lo pairs of the same format): clips that carry a
    gradient get a second forward with the weights rounded to the real side's operand format, and THOSE features are
    returned (the pooling decisions that route the gradient stay those of the exact weights): both sides of a DM class
    term then carry the same weight-rounding perturbation.  See distill.HipBackend.weight_format."""
    r, s = _PRECISION["real"], _PRECISION["syn"]
    if r in ("f16", "bf16") and s == r + "x3" and os.environ.get("VD_VALUE_PASS", "1") == "1":
        return r
    return None


def get_precision() -> Dict[str, str]:
    return dict(_PRECISION)


def get_engine(geo: P.NetGeometry, prec: str, device, prec_bwd: str = None, last_hilo: bool = False) -> "object":
    from . import engine
    device = torch.device(device)
    key = (geo.frames, geo.height, geo.width, prec, prec_bwd, bool(last_hilo),
           device.index if device.index is not None else torch.cuda.current_device())
    eng = _ENGINES.get(key)
    if eng is None:
        eng = engine.EmbedEngine(geo, prec=prec, device=device, prec_bwd=prec_bwd, last_hilo=last_hilo)
        _ENGINES[key] = eng
    return eng


class _EmbedFunction(torch.autograd.Function):
    """features = embed(x) on the HIP path; backward = HIP input gradient.

    Mixed mode and the weight-rounding bias of the single-pass real side (distill.HipBackend has the same logic for the
    fused trainers): a batch of >= 4 clips WITHOUT gradient is dealt to dithered weight sets (``engine.dither_groups``:
    its mean feature, which is what DM consumes, then carries no first-order rounding bias); clips WITH gradient get the
    exact-weight f16x3 forward, plus the value pass (``_weight_format``) when the real side they are compared with is NOT
    dithered.  Which of the two holds is ``net.real_dither``:

      * ``"auto"`` (default, the reference's call order ``embed(real).detach(); embed(syn)``, distill_baseline.py:344-349):
        the net remembers whether its latest no-gradient batch WITH THE CURRENT WEIGHTS was dithered; inference passes
        (``_infer_hip``) never touch that memory, and a weight update forgets it;
      * ``True`` / ``False``: the caller states it -- dither every no-gradient batch of >= 4 clips and never run the value pass /
        never dither and always run it -- for loops that embed the synthetic clips BEFORE the real ones or interleave
        other no-gradient batches."""

    @staticmethod
    def forward(ctx, x, net, dither_ok=True):
        need_grad = ctx.needs_input_grad[0]
        prec = _PRECISION["syn"] if need_grad else _PRECISION["real"]
        geo = P.NetGeometry(x.shape[1], x.shape[3], x.shape[4])
        # (the DM loop's real batches: single-pass with the last level in hi+lo pairs, like distill.HipBackend; inference
        #  passes -- dither_ok False -- stay single-pass throughout)
        hilo = (not need_grad) and dither_ok and prec in ("f16", "bf16") and _PRECISION["real_last"] == "x3" \
            and _PRECISION["syn"] == prec + "x3"
        eng = get_engine(geo, prec, x.device, _PRECISION["bwd"] if need_grad else None, last_hilo=hilo)
        if need_grad:
            net._sync_engine(eng)
            feats, saved = eng.forward(x, keep=True)
            q = _weight_format()
            mode = getattr(net, "real_dither", "auto")
            dithered = (getattr(net, "_real_dithered_key", None) == net._weights_key()) if mode == "auto" else bool(mode)
            if q is not None and not dithered:   # value pass (see distill.HipBackend.weight_format)
                # (the undithered real side still runs its last level on the exact hi+lo weights when real_last = x3: the value
                #  pass rounds only the levels that side multiplies by rn16(W))
                real_hilo = _PRECISION["real_last"] == "x3" and _PRECISION["syn"] == _PRECISION["real"] + "x3"
                net._sync_engine(eng, quantize=q, quantize_levels=(0, 1) if real_hilo else (0, 1, 2))
                feats = eng.forward(x)
            ctx.saved = saved
            ctx.eng = eng
            ctx.wkey = net._weights_key()
            ctx.net = net
            return feats
        from .engine import dither_groups
        mode = getattr(net, "real_dither", "auto")
        G = dither_groups(int(x.shape[0]), prec) if (dither_ok and mode is not False and _PRECISION["syn"] == prec + "x3") else 0
        if dither_ok:       # (remembered per weight state: stale after an update, untouched by inference passes)
            net._real_dithered_key = net._weights_key() if G else None
        net._sync_engine(eng, dither=G)
        if not G:
            return eng.forward(x)
        B = int(x.shape[0])
        feats = torch.empty((B, eng.num_feat), dtype=torch.float32, device=x.device)
        for g in range(G):     # clip j runs with dithered weight set j mod G
            feats[g::G] = eng.forward(x, index=torch.arange(g, B, G, device=x.device), group=g)
        return feats

    @staticmethod
    def backward(ctx, g):
        eng, net = ctx.eng, ctx.net
        if net._weights_key() != ctx.wkey:
            raise RuntimeError("ConvNet3D parameters changed between embed() and backward()")
        net._sync_engine(eng)
        return eng.backward(ctx.saved, g), None, None


class _ParamGradFunction(torch.autograd.Function):
    """(loss, logits, dCE/dparams...) = f(x); backward = HIP second-order pass (train.GradMatchEngine.vjp)."""

    @staticmethod
    def forward(ctx, x, labels, mask, net, slot=0):
        te = net._gm_engine(x, slot)
        loss, logits, g, state = te.param_grads(x, labels, list(net.parameters()), mask)
        ctx.te, ctx.state, ctx.net = te, state, net
        ctx.wkey = tuple((p.data_ptr(), p._version) for p in net.parameters())
        loss = loss.detach()
        ctx.mark_non_differentiable(loss, logits)
        return (loss, logits) + tuple(g)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, _gl, _glog, *v):
        net = ctx.net
        if tuple((p.data_ptr(), p._version) for p in net.parameters()) != ctx.wkey:
            raise RuntimeError("ConvNet3D parameters changed between param_grads() and backward()")
        return ctx.te.vjp(ctx.state, v, list(net.parameters())), None, None, None, None


def _zeros_like_or(t, ref):
    return torch.zeros_like(ref) if t is None else t


class _FeatFunction(torch.autograd.Function):
    """features = embed(x; w0,b0,w1,b1,w2,b2), differentiable w.r.t. the clips AND the parameters, twice."""

    @staticmethod
    def forward(ctx, x, net, *params6):
        te = net._gm_engine(x)
        feats, fs = te.ag_feat_forward(x, params6)
        ctx.te, ctx.fs = te, fs
        ctx.save_for_backward(x, *params6)
        return feats

    @staticmethod
    def backward(ctx, g):
        x, *params6 = ctx.saved_tensors
        need_dx, need_p = ctx.needs_input_grad[0], any(ctx.needs_input_grad[2:])
        outs = _FeatBackward.apply(g, x, ctx.te, ctx.fs, need_dx, need_p, torch.is_grad_enabled(), *params6)
        return (outs[0], None) + tuple(outs[1:])


class _FeatBackward(torch.autograd.Function):
    """(dx, dw0, db0, dw1, db1, dw2, db2) = backward of the conv levels for g_feat; its own backward is the second-order
    pass (train.GradMatchEngine.ag_feat_second_order): adjoints of g_feat, x and the parameters for adjoints of the six
    parameter gradients."""

    @staticmethod
    def forward(ctx, g_feat, x, te, fs, need_dx, need_p, keep, *params6):
        dx, g, bs = te.ag_feat_backward(fs, g_feat, params6, need_dx, need_p, keep)
        ctx.te, ctx.fs, ctx.bs = te, fs, bs
        ctx.save_for_backward(x, *params6)
        ctx.set_materialize_grads(False)
        return (dx,) + (tuple(g) if g is not None else (None,) * 6)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, vx, *v6):
        if vx is not None:
            raise NotImplementedError("ConvNet3D: second derivative through the INPUT gradient is not built "
                                      "(only through the parameter gradients, which is what DC / MTT need)")
        if ctx.bs is None:
            raise RuntimeError("ConvNet3D: double backward needs the first backward to run with create_graph=True")
        x, *params6 = ctx.saved_tensors
        if all(t is None for t in v6):
            return (None,) * (7 + 6)
        gbar3, xbar, hv = ctx.te.ag_feat_second_order(ctx.fs, ctx.bs, v6, params6, any(ctx.needs_input_grad[7:]))
        return (gbar3, xbar, None, None, None, None, None) + (tuple(hv) if hv is not None else (None,) * 6)


class _HeadFunction(torch.autograd.Function):
    """logits = max_t conv1x1x1(dropout(avgpool(features)))  (networks.py:741-745), twice differentiable."""

    @staticmethod
    def forward(ctx, feats, mask, te, w, b):
        hs = te.head_forward(feats.detach().to(torch.float32).contiguous(), mask, w, b)
        ctx.te, ctx.hs = te, hs
        ctx.save_for_backward(feats, w)
        return hs["logits"]

    @staticmethod
    def backward(ctx, dlog):
        feats, w = ctx.saved_tensors
        g_feat, g_w, g_b = _HeadBackward.apply(dlog, feats, w, ctx.te, ctx.hs)
        return g_feat, None, None, g_w, g_b


class _HeadBackward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dlog, feats, w, te, hs):
        dlog = dlog.detach().to(torch.float32).contiguous()
        g_w = torch.zeros((te.K, te.C), dtype=torch.float32, device=dlog.device)
        g_b = torch.zeros(te.K, dtype=torch.float32, device=dlog.device)
        g_feat = te.head_backward(hs, dlog, g_w, g_b)
        ctx.te, ctx.hs = te, hs
        ctx.save_for_backward(dlog, feats, w)
        ctx.set_materialize_grads(False)
        return g_feat, g_w.view(w.shape), g_b

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gbar_feat, v_w, v_b):
        dlog, feats, w = ctx.saved_tensors
        te = ctx.te
        if gbar_feat is None and v_w is None and v_b is None:
            return None, None, None, None, None
        gbar_feat = torch.zeros_like(feats) if gbar_feat is None else gbar_feat.detach().to(torch.float32).contiguous()
        v_w = (torch.zeros((te.K, te.C), dtype=torch.float32, device=dlog.device) if v_w is None
               else v_w.detach().to(torch.float32).reshape(te.K, te.C).contiguous())
        v_b = torch.zeros(te.K, dtype=torch.float32, device=dlog.device) if v_b is None else v_b.detach().to(torch.float32).contiguous()
        abar, wbar, _, dlogbar = te.head_second_order(ctx.hs, dlog, gbar_feat, v_w, v_b, want_params=True, hessian=False)
        return dlogbar, abar, wbar.view(w.shape), None, None



class ConvNet3D(nn.Module):
    def __init__(self, channel, num_classes, net_width, net_depth, net_act, net_norm, net_pooling, frames,
                 im_size=(32, 32), dropout_keep_prob=0.5):
        super().__init__()
        self.features, shape_feat = self._make_layers(channel, net_width, net_depth, net_norm, net_act,
                                                      net_pooling, im_size, frames)
        big = im_size[0] > 64
        self.avg_pool = nn.AvgPool3d(kernel_size=(2, 2, 2) if big else (2, 1, 1), stride=(1, 1, 1))
        self.dropout = nn.Dropout(dropout_keep_prob)
        self.logit = nn.Conv3d(net_width, num_classes, kernel_size=(1, 1, 1), stride=(1, 1, 1), bias=True)
        self._hip_ok = (channel == 3 and net_width == 128 and net_depth == 3 and net_act == 'relu'
                        and net_norm == 'none' and net_pooling == 'maxpooling')
        self._engine_keys: Dict[int, Tuple] = {}
        # mixed-precision DM: how embed() reconciles the single-pass real side with the hi+lo synthetic side ("auto" /
        # True / False; see _EmbedFunction).  Not part of the reference's surface; the default needs no caller change.
        self.real_dither = "auto"

    # -- construction (same layer sequence / naming as networks.py:792-814) ------------------
    @staticmethod
    def _activation(net_act):
        if net_act == 'relu':
            return nn.ReLU(inplace=True)
        if net_act == 'sigmoid':
            return nn.Sigmoid()
        if net_act == 'leakyrelu':
            return nn.LeakyReLU(negative_slope=0.01)
        exit('unknown activation function: %s' % net_act)

    @staticmethod
    def _pooling(net_pooling, first):
        if net_pooling == 'maxpooling':
            return nn.MaxPool3d(kernel_size=(1, 2, 2), stride=(1, 2, 2)) if first else nn.MaxPool3d(kernel_size=2, stride=2)
        if net_pooling == 'avgpooling':
            return nn.AvgPool3d(kernel_size=2, stride=2)
        if net_pooling == 'none':
            return None
        exit('unknown net_pooling: %s' % net_pooling)

    @staticmethod
    def _norm(net_norm, shape_feat):
        if net_norm == 'none':
            return None
        if net_norm == 'batchnorm':
            return nn.BatchNorm3d(shape_feat[0], affine=True)
        if net_norm == 'layernorm':
            return nn.LayerNorm(shape_feat, elementwise_affine=True)
        if net_norm == 'instancenorm':
            return nn.GroupNorm(shape_feat[0], shape_feat[0], affine=True)
        if net_norm == 'groupnorm':
            return nn.GroupNorm(4, shape_feat[0], affine=True)
        exit('unknown net_norm: %s' % net_norm)

    def _make_layers(self, channel, net_width, net_depth, net_norm, net_act, net_pooling, im_size, frames):
        layers = []
        cin = channel
        if im_size[0] == 28:
            im_size = (32, 32)
        shape_feat = [cin, frames, im_size[0], im_size[1]]
        for d in range(net_depth):
            cout = 64 if d == 0 else net_width
            layers.append(nn.Conv3d(cin, cout, kernel_size=(3, 7, 7), padding=(1, 3, 3), stride=(1, 2, 2)))
            shape_feat = [cout, shape_feat[1], shape_feat[2] // 2, shape_feat[3] // 2]
            norm = self._norm(net_norm, shape_feat)
            if norm is not None:
                layers.append(norm)
            layers.append(self._activation(net_act))
            cin = cout
            pool = self._pooling(net_pooling, d == 0)
            if pool is not None:
                layers.append(pool)
                shape_feat = [cout, shape_feat[1] // (1 if d == 0 else 2), shape_feat[2] // 2, shape_feat[3] // 2]
        return nn.Sequential(*layers), shape_feat

    # -- HIP dispatch ---------------------------------------------------------------------------
    def _feature_params(self):
        return [self.features[0].weight, self.features[0].bias, self.features[3].weight, self.features[3].bias,
                self.features[6].weight, self.features[6].bias]

    def _weights_key(self):
        return tuple((p.data_ptr(), p._version) for p in self._feature_params()) + (getattr(self, "_pack_epoch", 0),)

    def invalidate(self) -> None:
        """Force a re-pack of the MFMA operands at the next ``embed`` / ``forward``.  The packed copy is keyed on every
        parameter's storage address and autograd version counter, which optimisers and ``copy_`` / ``load_state_dict``
        bump; an in-place edit through ``.data`` (``p.data.mul_(2)``) does NOT bump it -- call this after such an edit."""
        self._pack_epoch = getattr(self, "_pack_epoch", 0) + 1

    def _sync_engine(self, eng, quantize=None, dither: int = 0, quantize_levels=(0, 1, 2)) -> None:
        # the cached engines outlive nets: remember the owner by a weak reference, not by id() (a new net may be built
        # at a freed net's address, on storage the caching allocator hands out again)
        import weakref
        key = (self._weights_key(), quantize, dither, tuple(quantize_levels) if quantize else None)
        owner = getattr(eng, "_owner_ref", None)
        if owner is None or owner() is not self or getattr(eng, "_owner_key", None) != key:
            eng.set_weights(self._feature_params(), quantize=quantize, dither=dither, quantize_levels=quantize_levels)
            eng._owner_key = key
            eng._owner_ref = weakref.ref(self)

    def _all_params(self):
        """The 8 tensors in parameters() order, read by ATTRIBUTE: under ReparamModule they are views of the flat
        parameter set as plain attributes (reparam_module.py:110-115), not registered nn.Parameters."""
        return self._feature_params() + [self.logit.weight, self.logit.bias]

    def _check_hip(self, x, what: str) -> None:
        if not x.is_cuda:
            raise RuntimeError("video_distillation_amd.ConvNet3D.%s has no CPU path: move the clips to a HIP "
                               "device (the CPU restatement lives in oracle/, for tests only)" % what)
        if not self._hip_ok:
            raise NotImplementedError("ConvNet3D.%s: only get_network('ConvNet3D')'s architecture (3 input channels, width "
                                      "128, depth 3, ReLU, no norm, max pooling; utils.py:608) has a HIP path, and there "
                                      "is no torch-op fallback" % what)

    def embed(self, x):
        """networks.py:747-751.  Frozen parameters (the DM loop, distill_baseline.py:336-349): mixed-precision fast path,
        gradient to the clips only.  Parameters that require a gradient: the twice-differentiable path."""
        self._check_hip(x, "embed")
        params = self._feature_params()
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            return _FeatFunction.apply(x, self, *params)
        return _EmbedFunction.apply(x, self)

    def _infer_hip(self, x):
        """Inference logits on the HIP path: MFMA features + fused head kernel (avg-pool, 1x1x1
        conv, max over T).  Used whenever no gradient is being recorded (evaluate_synset's three
        test passes per evaluation, utils.py:793-824)."""
        import ctypes
        from . import hip
        feats = _EmbedFunction.apply(x.detach(), self, False)      # inference: plain rn16 weights, one launch per layer
        g = P.NetGeometry(x.shape[1], x.shape[3], x.shape[4])
        d = g.layer_dims()[-1]
        C, To, Ho, Wo = d[1], d[8], d[9], d[10]
        kt, kh, kw = self.avg_pool.kernel_size
        K = self.logit.weight.shape[0]
        w = self.logit.weight.detach().reshape(K, C).float().contiguous()
        b = self.logit.bias.detach().float().contiguous()
        out = torch.empty((x.shape[0], K), dtype=torch.float32, device=x.device)
        hip.check(hip.lib().vd_head_fwd(hip.ptr(feats), hip.ptr(w), hip.ptr(b), ctypes.c_int64(x.shape[0]), C, To, Ho, Wo,
                                        kt, kh, kw, K, hip.ptr(out), hip.stream_ptr(x.device)), "vd_head_fwd")
        return out

    # -- training step on the HIP path (utils.epoch 'train' dispatches here) ----------------------
    def hip_trainable(self, x, optimizer, criterion) -> bool:
        """True when (net, optimiser, loss) is the combination ``evaluate_synset`` builds
        (utils.py:852-853): plain SGD with momentum / weight decay over exactly this module's
        parameters and mean cross-entropy."""
        if not (self._hip_ok and x.is_cuda and x.dim() == 5 and x.shape[2] == 3):
            return False
        if type(optimizer) is not torch.optim.SGD or len(optimizer.param_groups) != 1:
            return False
        g = optimizer.param_groups[0]
        if g.get("nesterov") or g.get("dampening", 0) != 0 or g.get("maximize"):
            return False
        mine = list(self.parameters())
        if len(g["params"]) != len(mine) or any(a is not b for a, b in zip(g["params"], mine)):
            return False
        if any((not p.is_cuda) or p.dtype != torch.float32 or not p.is_contiguous() for p in mine):
            return False
        return (type(criterion) is nn.CrossEntropyLoss and criterion.weight is None and criterion.reduction == "mean"
                and getattr(criterion, "label_smoothing", 0.0) == 0.0)

    def _train_engine(self, x, slot: int = 0, prec_bwd: str = None):
        """``slot``: engines own workspaces and packed operands, so calls that run concurrently on different streams (the
        class lanes of distill.GMTrainer) each use their own instance.  ``prec_bwd`` overrides the ``train_bwd`` precision."""
        from . import train
        hint = _batch_hint(x.shape[0])
        prec_bwd = prec_bwd or _PRECISION["train_bwd"]
        key = ("train", x.shape[1], x.shape[3], x.shape[4], self.logit.weight.shape[0], hint, _PRECISION["train"], prec_bwd,
               x.device.index if x.device.index is not None else torch.cuda.current_device(), slot)
        te = _ENGINES.get(key)
        if te is None:
            te = train.TrainEngine(P.NetGeometry(x.shape[1], x.shape[3], x.shape[4]), self.logit.weight.shape[0],
                                   self.avg_pool.kernel_size, x.device, prec=_PRECISION["train"],
                                   prec_bwd=prec_bwd, batch_hint=hint)
            _ENGINES[key] = te
        return te

    def _gm_engine(self, x, slot: int = 0):
        from . import train
        hint = _batch_hint(x.shape[0])
        key = ("gm", x.shape[1], x.shape[3], x.shape[4], self.logit.weight.shape[0], hint, _PRECISION["match"],
               x.device.index if x.device.index is not None else torch.cuda.current_device(), slot)
        te = _ENGINES.get(key)
        if te is None:
            te = train.GradMatchEngine(P.NetGeometry(x.shape[1], x.shape[3], x.shape[4]), self.logit.weight.shape[0],
                                       self.avg_pool.kernel_size, x.device, prec=_PRECISION["match"], batch_hint=hint)
            _ENGINES[key] = te
        return te

    def _dropout_mask(self, x, te):
        if self.training and self.dropout.p > 0:
            keep = 1.0 - self.dropout.p
            return torch.bernoulli(torch.full((x.shape[0], te.C, te.Tp), keep, device=x.device)) / keep
        return None

    def param_grads(self, x, labels, create_graph: bool = False, mask=None, slot: int = 0):
        """``torch.autograd.grad(CrossEntropyLoss()(net(x), labels), net.parameters(), create_graph=...)``
        on the HIP path (gradient matching: upstream DC loop).  Returns
        (loss, [8 gradients in parameters() order]).  With ``create_graph`` the gradients are
        differentiable w.r.t. ``x`` (needed for ``match_loss(gw_syn, gw_real).backward()``): forward
        mode of that second derivative is the HIP second-order pass of train.GradMatchEngine.
        ``mask`` (B,128,T') overrides the dropout draw of train mode."""
        if not (self._hip_ok and x.is_cuda):
            raise RuntimeError("ConvNet3D.param_grads: HIP tensors and the ConvNet3D of get_network only (no CPU path)")
        if create_graph:
            te = self._gm_engine(x, slot)
            if mask is None:
                mask = self._dropout_mask(x, te)
            out = _ParamGradFunction.apply(x, labels, mask, self, slot)
            return out[0], list(out[2:])
        te = self._train_engine(x, slot, _PRECISION["match_real_bwd"])
        if mask is None:
            mask = self._dropout_mask(x, te)
        loss, _, g = te.loss_and_grads(x, labels, list(self.parameters()), mask)
        return loss, [t.clone() for t in g]

    def param_grads_grouped(self, x, labels, groups: int, mask=None, slot: int = 0):
        """``param_grads(create_graph=False)`` of ``groups`` equal consecutive sub-batches of x (the real batches of several classes
        of one gradient-matching step) with one forward over all of them (train.TrainEngine.loss_and_grads_grouped).
        Returns [(loss, [8 gradients])] per sub-batch."""
        if not (self._hip_ok and x.is_cuda):
            raise RuntimeError("ConvNet3D.param_grads_grouped: HIP tensors and the ConvNet3D of get_network only (no CPU path)")
        te = self._train_engine(x, slot, _PRECISION["match_real_bwd"])
        if mask is None:
            mask = self._dropout_mask(x, te)
        losses, _, gs = te.loss_and_grads_grouped(x, labels, list(self.parameters()), int(groups), mask)
        return [(losses[k], gs[k]) for k in range(int(groups))]

    def hip_train_step(self, x, labels, optimizer):
        """forward + CrossEntropyLoss + backward + ``optimizer.step()`` for one batch, on the HIP
        path.  ``x`` is the (already standardised) batch (B,T,3,H,W).  Momentum buffers live in
        ``optimizer.state`` exactly where torch keeps them.  Returns (logits, loss)."""
        te = self._train_engine(x)
        params = list(self.parameters())
        # the weight gradients of a level run on a side stream under the input-gradient passes (5.30 -> 5.08 ms per step of 50
        # clips); not under the class lanes of gradient matching, where the extra streams cost 4 % (VD_WGRAD_SIDE=0: off)
        te.side_wgrad = os.environ.get("VD_WGRAD_SIDE", "1") == "1"
        try:
            loss, logits, grads = te.loss_and_grads(x, labels, params, self._dropout_mask(x, te))
        finally:
            te.side_wgrad = False
        grp = optimizer.param_groups[0]
        bufs = [optimizer.state[p].get("momentum_buffer") for p in params]
        new = te.sgd_step([p.data for p in params], grads, bufs, float(grp["lr"]), float(grp["momentum"]),
                          float(grp["weight_decay"]))
        for p, b in zip(params, new):
            optimizer.state[p]["momentum_buffer"] = b
            torch.autograd.graph.increment_version(p)
        return logits, loss

    def forward(self, x):
        """networks.py:738-745 on the HIP path.  Evaluation without gradient: MFMA features + fused head kernel.
        Otherwise (training mode, or anything requires a gradient): ``_FeatFunction`` + ``_HeadFunction`` -- logits that
        are differentiable w.r.t. clips and parameters, twice (``create_graph=True``, distill_baseline.py:250)."""
        self._check_hip(x, "forward")
        params = self._all_params()
        need = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))
        if not need and not self.training:
            return self._infer_hip(x)
        te = self._gm_engine(x)
        feats = _FeatFunction.apply(x, self, *params[:6])
        return _HeadFunction.apply(feats, self._dropout_mask(x, te), te, params[6], params[7])
