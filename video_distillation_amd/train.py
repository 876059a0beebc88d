"""Training step of ConvNet3D on the HIP path: the evaluation half of the distillation loop.

What the reference does with ``loss.backward(); optimizer.step()`` inside ``epoch('train')``
(utils.py:765-792) for the network ``evaluate_synset`` trains on the synthetic set
(utils.py:848-886): forward keeping activations, classifier head with dropout and max over
frames, cross-entropy, head backward, then per layer (last to first) un-pool + ReLU backward,
bias gradient, weight gradient (tile program of the MFMA kernel, plan.plan_wgrad) and the
input-gradient passes down to layer 1, finally SGD with momentum and weight decay.

``TrainEngine`` owns the per-geometry device state; ``ConvNet3D.hip_train_step`` (networks.py)
is the module-level entry ``utils.epoch`` dispatches to.  No CPU path.
"""
from __future__ import annotations

import ctypes
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import hip
from . import plan as P
from .engine import EmbedEngine, WgradOp


def standardize(x: torch.Tensor) -> torch.Tensor:
    """(x - x.mean()) / x.std() with batch-global scalars (utils.py:770), on the device."""
    if not x.is_cuda:
        raise RuntimeError("standardize: HIP tensors only (no CPU path)")
    x = x.detach().to(torch.float32).contiguous()
    out = torch.empty_like(x)
    scratch = torch.empty(2, dtype=torch.float64, device=x.device)
    hip.check(hip.lib().vd_standardize(hip.ptr(x), ctypes.c_int64(x.numel()), hip.ptr(scratch), hip.ptr(out),
                                       hip.stream_ptr(x.device)), "vd_standardize")
    return out


class TrainEngine:
    def __init__(self, geo: P.NetGeometry, num_classes: int, pool_kernel: Tuple[int, int, int], device,
                 prec: str = "f16x3", prec_bwd: str = "f16x3"):
        self.geo = geo
        self.device = torch.device(device)
        self.K = int(num_classes)
        self.pool_kernel = tuple(int(k) for k in pool_kernel)
        self.eng = EmbedEngine(geo, prec=prec, device=device, chunk=1 << 30, prec_bwd=prec_bwd)
        if self.eng.planes_bwd > self.eng.planes or (prec[:2] != prec_bwd[:2]):
            raise ValueError("backward operands are read from the forward's activations: %s / %s do not combine"
                             % (prec, prec_bwd))
        self.prec_bwd_name = prec_bwd
        self._wg: Dict[Tuple[int, int], WgradOp] = {}
        d = self.eng.dims[-1]
        self.C, self.To, self.Ho, self.Wo = d[1], d[8], d[9], d[10]
        self.Tp = self.To - self.pool_kernel[0] + 1
        self.shapes = [(64, 3, 3, 7, 7), (64,), (128, 64, 3, 7, 7), (128,), (128, 128, 3, 7, 7), (128,),
                       (self.K, self.C, 1, 1, 1), (self.K,)]
        self.sizes = [int(np.prod(s)) for s in self.shapes]
        self.gflat = torch.zeros(sum(self.sizes), dtype=torch.float32, device=self.device)

    def _wgrad(self, li: int, nb: int) -> WgradOp:
        op = self._wg.get((li, nb))
        if op is None:
            cin, cout, t, h, w = self.eng.dims[li][:5]
            op = WgradOp(cin, cout, t, h, w, nb, self.prec_bwd_name, self.device)
            self._wg[(li, nb)] = op
        return op

    def grads(self) -> List[torch.Tensor]:
        out, o = [], 0
        for s, n in zip(self.shapes, self.sizes):
            out.append(self.gflat[o:o + n].view(*s))
            o += n
        return out

    # ------------------------------------------------------------------------------------
    def loss_and_grads(self, x: torch.Tensor, labels: torch.Tensor, params: Sequence[torch.Tensor],
                       mask: Optional[torch.Tensor] = None):
        """x (B,T,3,H,W) fp32 (already standardised), labels (B,) int64, params = the 8 network
        tensors in ``parameters()`` order, mask (B,C,Tp) dropout multipliers or None.
        Returns (mean CE loss [device scalar], logits (B,K), [8 gradient tensors])."""
        eng, L, st = self.eng, hip.lib(), hip.stream_ptr(self.device)
        B = int(x.shape[0])
        x = x.detach().to(torch.float32).contiguous()
        labels = labels.to(self.device, torch.int64).contiguous()
        eng.set_weights(params[:6])
        for li in (1, 2):
            for dp in eng.bwd[li]:
                dp.pack(eng._weights[2 * li])
        feats, saved = eng.forward(x, keep=True)
        (_, nb, am0, am1, am2), = saved
        per1 = int(np.prod(eng.fwd[0].plan.out_shape[:-1]))
        per2 = int(np.prod(eng.fwd[1].plan.out_shape[:-1]))
        acts = [None, eng._buf("act1", (eng.planes, nb * per1, 8), torch.int16),
                eng._buf("act2", (eng.planes, nb * per2, 8), torch.int16)]
        act_plane = [0, nb * per1, nb * per2]

        wl = params[6].detach().reshape(self.K, self.C).to(torch.float32).contiguous()
        bl = params[7].detach().to(torch.float32).contiguous()
        kt, kh, kw = self.pool_kernel
        dropped = torch.empty((B, self.Tp, self.C), dtype=torch.float32, device=self.device)
        logits = torch.empty((B, self.K), dtype=torch.float32, device=self.device)
        amt = torch.empty((B, self.K), dtype=torch.int32, device=self.device)
        if mask is not None:
            mask = mask.to(self.device, torch.float32).contiguous()
            assert tuple(mask.shape) == (B, self.C, self.Tp), mask.shape
        hip.check(L.vd_head_train_fwd(hip.ptr(feats), hip.ptr(mask), hip.ptr(wl), hip.ptr(bl), ctypes.c_int64(B), self.C,
                                      self.To, self.Ho, self.Wo, kt, kh, kw, self.K, hip.ptr(dropped), hip.ptr(logits),
                                      hip.ptr(amt), st), "vd_head_train_fwd")
        loss_c = torch.empty(B, dtype=torch.float32, device=self.device)
        dlog = torch.empty((B, self.K), dtype=torch.float32, device=self.device)
        hip.check(L.vd_ce_loss(hip.ptr(logits), hip.ptr(labels), B, self.K, hip.ptr(loss_c), hip.ptr(dlog), st), "vd_ce_loss")
        self.gflat.zero_()
        g = self.grads()
        g_feat = torch.empty((B, eng.num_feat), dtype=torch.float32, device=self.device)
        hip.check(L.vd_head_train_bwd(hip.ptr(dlog), hip.ptr(amt), hip.ptr(dropped), hip.ptr(mask), hip.ptr(wl),
                                      ctypes.c_int64(B), self.C, self.To, self.Ho, self.Wo, kt, kh, kw, self.K,
                                      hip.ptr(g[6]), hip.ptr(g[7]), hip.ptr(g_feat), st), "vd_head_train_bwd")

        grad, layout = g_feat, 0
        scaled = eng.prec_bwd in (hip.PREC["f16"], hip.PREC["f16x3"])
        for li, am in ((2, am2), (1, am1), (0, am0)):
            cin, cout, t, h, w, T, OH, OW, To, Ho, Wo, pt = eng.dims[li]
            nslots = nb * (cout // 8) * T * OH * OW
            dy = eng._buf("dy%d" % li, (eng.planes_bwd, nslots, 8), torch.int16)
            lo = dy[1] if eng.planes_bwd == 2 else None
            sc = inv = None
            if scaled:
                scb = eng._buf("gscale%d" % li, (4,), torch.float32)
                hip.check(L.vd_absmax_scale(hip.ptr(grad), ctypes.c_int64(grad.numel()), ctypes.c_float(1024.0),
                                            hip.ptr(scb), st), "vd_absmax_scale")
                sc, inv = scb, scb[1:]
            hip.check(L.vd_unpool_relu_bwd(hip.ptr(grad), hip.ptr(am), ctypes.c_int64(nb), cout, To, Ho, Wo, pt, T, OH, OW,
                                           layout, hip.ptr(dy[0]), hip.ptr(lo), eng.prec_bwd, hip.ptr(sc), st),
                      "vd_unpool_relu_bwd")
            hip.check(L.vd_bias_grad(hip.ptr(dy), ctypes.c_int64(nslots), eng.planes_bwd, ctypes.c_int64(nb), cout,
                                     ctypes.c_int64(T * OH * OW), eng.prec_bwd, hip.ptr(inv), hip.ptr(g[2 * li + 1]), st),
                      "vd_bias_grad")
            op = self._wgrad(li, nb)
            if li == 0:
                op.run(x, True, 0, dy, nslots, g[0], out_scale=inv)
            else:
                op.run(acts[li], False, act_plane[li], dy, nslots, g[2 * li], out_scale=inv)
            if li > 0:
                out = eng._buf("dx%d" % li, (nb, t, h, w, cin), torch.float32)
                for dp in eng.bwd[li]:
                    dp.run(dy, nslots, None, out.data_ptr(), 0, None, nb, out_scale=inv)
                grad, layout = out, 1
        return loss_c.mean(), logits, g

    # ------------------------------------------------------------------------------------
    def sgd_step(self, params: Sequence[torch.Tensor], grads: Sequence[torch.Tensor], bufs: Sequence[Optional[torch.Tensor]],
                 lr: float, momentum: float, weight_decay: float) -> List[torch.Tensor]:
        """In-place torch.optim.SGD(momentum, weight_decay) update; returns the momentum buffers
        (created on first use, as torch does)."""
        L, st = hip.lib(), hip.stream_ptr(self.device)
        out = []
        for p, gr, b in zip(params, grads, bufs):
            first = b is None
            if first:
                b = torch.empty_like(p, memory_format=torch.contiguous_format)
            assert p.is_contiguous() and p.dtype == torch.float32
            hip.check(L.vd_sgd_momentum_wd(hip.ptr(p), hip.ptr(b), hip.ptr(gr), ctypes.c_int64(p.numel()), ctypes.c_float(lr),
                                           ctypes.c_float(momentum), ctypes.c_float(weight_decay), int(first), st),
                      "vd_sgd_momentum_wd")
            out.append(b)
        return out
