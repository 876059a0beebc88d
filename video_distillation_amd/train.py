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

import contextlib
import ctypes
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import hip
from . import plan as P
from . import engine
from .engine import EmbedEngine, WgradOp, _DevPlan, run_together, GRAD_TARGET


def standardize(x: torch.Tensor) -> torch.Tensor:
    """(x - x.mean()) / x.std() with batch-global scalars (utils.py:770), on the device."""
    if not x.is_cuda:
        raise RuntimeError("standardize: HIP tensors only (no CPU path)")
    x = x.detach().to(torch.float32).contiguous()
    out = torch.empty_like(x)
    if hip.deterministic():         # fixed summation order: per-block partial sums, folded identically by every block
        scratch = torch.empty(4096, dtype=torch.float64, device=x.device)
        hip.check(hip.lib().vd_standardize_ordered(hip.ptr(x), ctypes.c_int64(x.numel()), hip.ptr(scratch), hip.ptr(out),
                                                   hip.stream_ptr(x.device)), "vd_standardize_ordered")
        return out
    scratch = torch.empty(2, dtype=torch.float64, device=x.device)
    hip.check(hip.lib().vd_standardize(hip.ptr(x), ctypes.c_int64(x.numel()), hip.ptr(scratch), hip.ptr(out),
                                       hip.stream_ptr(x.device)), "vd_standardize")
    return out


class TrainEngine:
    def __init__(self, geo: P.NetGeometry, num_classes: int, pool_kernel: Tuple[int, int, int], device,
                 prec: str = "f16x3", prec_bwd: str = "f16x3", batch_hint: Optional[int] = None):
        self.geo = geo
        self.device = torch.device(device)
        self.K = int(num_classes)
        self.pool_kernel = tuple(int(k) for k in pool_kernel)
        self.batch_hint = batch_hint
        self.eng = EmbedEngine(geo, prec=prec, device=device, chunk=1 << 30, prec_bwd=prec_bwd, batch_hint=batch_hint,
                               bwd0_small=os.environ.get("VD_BWD0_SMALL", "1") == "1")
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
        self.side_wgrad = False        # set by ConvNet3D.hip_train_step (see feat_backward)

    def _wgrad(self, li: int, nb: int) -> WgradOp:
        det = hip.deterministic()
        op = self._wg.get((li, nb, det))
        if op is None:
            cin, cout, t, h, w = self.eng.dims[li][:5]
            op = WgradOp(cin, cout, t, h, w, nb, self.prec_bwd_name, self.device, ordered=det)
            self._wg[(li, nb, det)] = op
        return op

    def grads(self) -> List[torch.Tensor]:
        out, o = [], 0
        for s, n in zip(self.shapes, self.sizes):
            out.append(self.gflat[o:o + n].view(*s))
            o += n
        return out

    def _forward(self, x, params):
        """Forward with kept activations ("act1"/"act2" in the engine workspace) and arg-max bytes."""
        feats, saved = self.eng.forward(x, keep=True)
        (_, nb, am0, am1, am2), = saved
        return feats, nb, (am0, am1, am2)

    # ------------------------------------------------------------------------------------
    def _acts(self, nb: int):
        eng = self.eng
        per1 = int(np.prod(eng.fwd[0].plan.out_shape[:-1]))
        per2 = int(np.prod(eng.fwd[1].plan.out_shape[:-1]))
        acts = [None, eng._buf("act1", (eng.planes, nb * per1, 8), torch.int16),
                eng._buf("act2", (eng.planes, nb * per2, 8), torch.int16)]
        return acts, [0, nb * per1, nb * per2]

    def head_forward(self, feats: torch.Tensor, mask: Optional[torch.Tensor], w: torch.Tensor, b: torch.Tensor) -> dict:
        """AvgPool3d -> dropout mask -> 1x1x1 conv -> max over frames (networks.py:741-745) of (B, num_feat) features.
        Returns the head state {logits, dropped, amt, mask, wl, bl}."""
        L, st = hip.lib(), hip.stream_ptr(self.device)
        B = int(feats.shape[0])
        wl = w.detach().reshape(self.K, self.C).to(torch.float32).contiguous()
        bl = b.detach().to(torch.float32).contiguous()
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
        return dict(logits=logits, dropped=dropped, amt=amt, mask=mask, wl=wl, bl=bl)

    def head_backward(self, hs: dict, dlog: torch.Tensor, g_w: torch.Tensor, g_b: torch.Tensor) -> torch.Tensor:
        """Backward of ``head_forward`` for the logit gradient ``dlog`` (B,K): accumulates into g_w / g_b (zeroed by the
        caller), returns the feature gradient (B, num_feat)."""
        L, st = hip.lib(), hip.stream_ptr(self.device)
        B = int(dlog.shape[0])
        kt, kh, kw = self.pool_kernel
        g_feat = torch.empty((B, self.eng.num_feat), dtype=torch.float32, device=self.device)
        fn = L.vd_head_train_bwd_ordered if hip.deterministic() else L.vd_head_train_bwd
        hip.check(fn(hip.ptr(dlog), hip.ptr(hs["amt"]), hip.ptr(hs["dropped"]), hip.ptr(hs["mask"]),
                     hip.ptr(hs["wl"]), ctypes.c_int64(B), self.C, self.To, self.Ho, self.Wo, kt, kh, kw,
                     self.K, hip.ptr(g_w), hip.ptr(g_b), hip.ptr(g_feat), st), "vd_head_train_bwd")
        return g_feat

    def feat_backward(self, x: torch.Tensor, nb: int, am, g_feat: torch.Tensor, g: Optional[Sequence[torch.Tensor]],
                      dx: Optional[torch.Tensor] = None, keep_dense: bool = False, acts_override=None) -> None:
        """Backward of the three conv levels for the feature gradient ``g_feat`` (nb, num_feat): per layer, last to
        first, un-pool + ReLU backward, bias gradient, weight gradient (accumulated into g[0..5], zeroed by the caller;
        skipped when ``g`` is None) and the input-gradient passes (down to the pixels, into ``dx``, when given).  The
        engine's weights / packed dgrad operands must be current; activations are read from the engine workspace.
        ``keep_dense``: a second-order sweep will read the first layer's dense gradient slots (workspace ``dy0``); without
        it and without ``dx`` they are never materialised (``WgradOp.run_pooled``)."""
        eng, L, st = self.eng, hip.lib(), hip.stream_ptr(self.device)
        # (``acts_override`` = (activation views, plane strides): the clips are a slice of a larger forward, loss_and_grads_grouped)
        acts, act_plane = acts_override if acts_override is not None else self._acts(nb)
        grad, layout = g_feat, 0
        scaled = eng.prec_bwd in (hip.PREC["f16"], hip.PREC["f16x3"])
        # ``side_wgrad``: the parameter side of a level (bias + weight gradient: staging, packing, one tile program, replica sum)
        # depends only on the level's incoming gradient, and nothing downstream depends on it -- it runs on a side stream under
        # the input-gradient passes of the same and the following levels; joined before returning
        main = torch.cuda.current_stream(self.device)
        side = self._side_stream() if (self.side_wgrad and g is not None) else None
        for li in (2, 1, 0):
            cin, cout, t, h, w, T, OH, OW, To, Ho, Wo, pt = eng.dims[li]
            nslots = nb * (cout // 8) * T * OH * OW
            dense = li > 0 or dx is not None or keep_dense   # the first layer's dense dy: pixel-gradient and second-order passes only
            if not dense and g is None:
                break
            sc = inv = None
            if scaled:
                scb = eng._buf("gscale%d" % li, (4,), torch.float32)
                hip.check(L.vd_absmax_scale(hip.ptr(grad), ctypes.c_int64(grad.numel()), ctypes.c_float(GRAD_TARGET()),
                                            hip.ptr(scb), st), "vd_absmax_scale")
                sc, inv = scb, scb[1:]
            if dense:
                dy = eng._buf("dy%d" % li, (eng.planes_bwd, nslots, 8), torch.int16)
                lo = dy[1] if eng.planes_bwd == 2 else None
                hip.check(L.vd_unpool_relu_bwd(hip.ptr(grad), hip.ptr(am[li]), ctypes.c_int64(nb), cout, To, Ho, Wo, pt, T, OH, OW,
                                               layout, hip.ptr(dy[0]), hip.ptr(lo), eng.prec_bwd, hip.ptr(sc), st),
                          "vd_unpool_relu_bwd")
            if g is not None:
                if side is not None:
                    side.wait_stream(main)          # the level's gradient and its scale are queued on the main stream
                with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
                    # bias gradient from the pooled gradient (the dense dy has one non-zero per live pool window)
                    if hip.deterministic():
                        nsc = int(L.vd_bias_grad_pooled_scratch_floats(ctypes.c_int64(nb), cout, ctypes.c_int64(To * Ho * Wo)))
                        bsc = eng._buf("bias_part%d" % li, (nsc,), torch.float32)
                        hip.check(L.vd_bias_grad_pooled_ordered(hip.ptr(grad), hip.ptr(am[li]), ctypes.c_int64(nb), cout,
                                                                ctypes.c_int64(To * Ho * Wo), layout, hip.ptr(bsc), hip.ptr(g[2 * li + 1]),
                                                                hip.stream_ptr(self.device)), "vd_bias_grad_pooled_ordered")
                    else:
                        hip.check(L.vd_bias_grad_pooled(hip.ptr(grad), hip.ptr(am[li]), ctypes.c_int64(nb), cout,
                                                        ctypes.c_int64(To * Ho * Wo), layout, hip.ptr(g[2 * li + 1]),
                                                        hip.stream_ptr(self.device)), "vd_bias_grad_pooled")
                    op = self._wgrad(li, nb)
                    # pooled gradient -> packed B operand of the weight-gradient program in one pass (4-8x fewer bytes than
                    # re-reading the dense slots, which for the first layer are not even written when nobody else needs them)
                    if li == 0:
                        op.run_pooled(x, True, 0, grad, am[0], layout, (To, Ho, Wo, pt), sc, g[0], out_scale=inv)
                    else:
                        op.run_pooled(acts[li], False, act_plane[li], grad, am[li], layout, (To, Ho, Wo, pt), sc, g[2 * li], out_scale=inv)
            if li > 0 or dx is not None:
                out = dx if li == 0 else eng._buf("dx%d" % li, (nb, t, h, w, cin), torch.float32)
                run_together(eng.bwd[li], dy, nslots, None, out.data_ptr(), 0, None, nb, out_scale=inv)
                grad, layout = out, 1
        if side is not None:
            main.wait_stream(side)

    def _side_stream(self):
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.device)
        return self._side

    def loss_and_grads(self, x: torch.Tensor, labels: torch.Tensor, params: Sequence[torch.Tensor],
                       mask: Optional[torch.Tensor] = None, state: Optional[dict] = None):
        """x (B,T,3,H,W) fp32 (already standardised), labels (B,) int64, params = the 8 network
        tensors in ``parameters()`` order, mask (B,C,Tp) dropout multipliers or None.
        Returns (mean CE loss [device scalar], logits (B,K), [8 gradient tensors]).  ``state`` (a dict)
        receives what a second-order pass needs (GradMatchEngine)."""
        eng, L, st = self.eng, hip.lib(), hip.stream_ptr(self.device)
        B = int(x.shape[0])
        x = x.detach().to(torch.float32).contiguous()
        labels = labels.to(self.device, torch.int64).contiguous()
        with engine.batched_packs():          # (forward + input-gradient operands of the three levels: one launch)
            eng.set_weights(params[:6])
            for li in (1, 2):
                for dp in eng.bwd[li]:
                    dp.pack(eng._weights[2 * li])
        feats, nb, am = self._forward(x, params)
        hs = self.head_forward(feats, mask, params[6], params[7])
        logits = hs["logits"]
        loss_c = torch.empty(B, dtype=torch.float32, device=self.device)
        dlog = torch.empty((B, self.K), dtype=torch.float32, device=self.device)
        hip.check(L.vd_ce_loss(hip.ptr(logits), hip.ptr(labels), B, self.K, hip.ptr(loss_c), hip.ptr(dlog), st), "vd_ce_loss")
        self.gflat.zero_()
        g = self.grads()
        g_feat = self.head_backward(hs, dlog, g[6], g[7])
        self.feat_backward(x, nb, am, g_feat, g, keep_dense=state is not None)
        if state is not None:
            _, act_plane = self._acts(nb)
            state.update(nb=nb, am=am, dropped=hs["dropped"], logits=logits, dlog=dlog, amt=hs["amt"], mask=hs["mask"],
                         wl=hs["wl"], act_plane=act_plane, x=x)
        return loss_c.mean(), logits, g

    def loss_and_grads_grouped(self, x: torch.Tensor, labels: torch.Tensor, params: Sequence[torch.Tensor], groups: int,
                               mask: Optional[torch.Tensor] = None):
        """``loss_and_grads`` of ``groups`` equal consecutive sub-batches of x (the real batches of several classes of a
        gradient-matching step: same network, independent mean-CE losses) with ONE forward over all clips -- a 64-clip launch
        leaves a quarter of the chip idle in the last level and a tail in the others -- and one backward per sub-batch on its slice
        of the kept activations / arg-max bytes.  Returns (losses (groups,), logits, [8 gradients] per sub-batch); equal to
        per-sub-batch calls (tests/test_gpu_train.py::test_grouped_real_batches_equal_separate_calls)."""
        eng, L, st = self.eng, hip.lib(), hip.stream_ptr(self.device)
        B = int(x.shape[0])
        if groups < 1 or B % groups != 0:
            raise ValueError("loss_and_grads_grouped: %d clips do not split into %d equal sub-batches" % (B, groups))
        per = B // groups
        x = x.detach().to(torch.float32).contiguous()
        labels = labels.to(self.device, torch.int64).contiguous()
        with engine.batched_packs():          # (forward + input-gradient operands of the three levels: one launch)
            eng.set_weights(params[:6])
            for li in (1, 2):
                for dp in eng.bwd[li]:
                    dp.pack(eng._weights[2 * li])
        feats, nb, am = self._forward(x, params)
        hs = self.head_forward(feats, mask, params[6], params[7])
        logits = hs["logits"]
        loss_c = torch.empty(B, dtype=torch.float32, device=self.device)
        dlog = torch.empty((B, self.K), dtype=torch.float32, device=self.device)
        for k in range(groups):                 # (per sub-batch: vd_ce_loss scales the logit gradient by 1 / its batch size -- exactly what a
            sl = slice(k * per, (k + 1) * per)  #  separate call does; rescaling a 1 / B gradient would differ in the last bit, which the
            hip.check(L.vd_ce_loss(hip.ptr(logits[sl]), hip.ptr(labels[sl]), per, self.K, hip.ptr(loss_c[sl]),   # f16 passes' power-of-two
                                   hip.ptr(dlog[sl]), st), "vd_ce_loss")                                         #  scaling can amplify)
        acts, act_plane = self._acts(nb)
        per1, per2, nf = act_plane[1] // nb, act_plane[2] // nb, eng.num_feat
        outs = []
        for k in range(groups):
            sl = slice(k * per, (k + 1) * per)
            self.gflat.zero_()
            g = self.grads()
            hs_k = dict(hs, amt=hs["amt"][sl], dropped=hs["dropped"][sl], mask=None if hs["mask"] is None else hs["mask"][sl])
            g_feat = self.head_backward(hs_k, dlog[sl], g[6], g[7])
            am_k = (am[0][k * per * per1 * 8:(k + 1) * per * per1 * 8], am[1][k * per * per2 * 8:(k + 1) * per * per2 * 8],
                    am[2][k * per * nf:(k + 1) * per * nf])
            acts_k = [None, acts[1][:, k * per * per1:(k + 1) * per * per1], acts[2][:, k * per * per2:(k + 1) * per * per2]]
            self.feat_backward(x[sl], per, am_k, g_feat, g, acts_override=(acts_k, act_plane))
            outs.append([t.clone() for t in g])
        return loss_c.view(groups, per).mean(1), logits, outs

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


class GradMatchEngine(TrainEngine):
    """Parameter gradients of CE(net(x)) that stay differentiable w.r.t. the clips x -- the
    ``torch.autograd.grad(loss, params, create_graph=True)`` of gradient matching (upstream DC
    loop; distill_baseline.py:250 for the reference's DC/MTT callers) -- and the vector-Jacobian
    product that ``match_loss(gw_syn, gw_real).backward()`` needs: d <v, gw(x)> / dx.

    With a_l the layer inputs, dz_l the gradients at the conv outputs, P_l the (fixed) ReLU +
    arg-max selection of layer l and V_l = v for the weights of layer l, the adjoint sweep is
      up:    dzbar_l = conv(a_l, V_l) + vb_l + conv(gbar_l, W_l),  gbar_{l+1} = P_l dzbar_l
      head:  second-order pass through avg-pool / dropout / logit conv / max over T / CE Hessian
      down:  abar_l = convT(P_l^T abar_{l+1}, W_l) + convT(dz_l, V_l)
    Every conv is a tile program of the MFMA kernel: the two convs of the upward sweep are one
    program with K-concatenated operands ([a_l | gbar_l] x [V_l | W_l], ``src_split_cc``) and the
    ``select`` epilogue; the second convT of the downward sweep accumulates with fp32 atomics.

    Operand formats.  ``f16x3`` (round 6, the default): fp16 hi+lo pairs, 22 bits, with every gradient-like operand brought into
    fp16's range by an exact power-of-two scale that the fp32 epilogue of the consuming program undoes -- the first-order
    gradients at the conv outputs (per level, TrainEngine.feat_backward), the adjoints V_l of the weights, the incoming adjoint
    of every level of the downward sweep, and the tangents gbar_l of the upward sweep, which a program writes as fp32
    (``select = 2``) so that their range can be MEASURED before the split (vd_absmax_scale + vd_split_scaled); V_l and gbar_l
    share an accumulator (K-concatenated operands) and therefore a scale, the smaller of the two.  Weights are packed
    x 2^8 (hip.F16X3_WSHIFT).  The forward of this engine is then its own fp16 hi+lo forward: activations are shared, nothing is
    re-split.  ``bf16x3`` (rounds 1 - 5): bf16 hi+lo pairs, 16 bits, unscaled, on top of an f16x3 forward for the pooling
    decisions -- 3 - 11x further from the exact gradient than fp32 arithmetic over the ten unrolled steps of configuration 5
    (profiles/r05_parity_mtt_unroll.json); kept as the A/B reference (``VD_PREC_MATCH=bf16x3``)."""

    V_TARGET = 128.0          # max |V_l| * scale in [64, 128): x 2^8 in the packed operand stays below fp16's 65504

    def __init__(self, geo: P.NetGeometry, num_classes: int, pool_kernel, device, prec: str = "f16x3",
                 batch_hint: Optional[int] = None):
        if prec not in ("bf16x3", "f16x3"):
            raise ValueError("GradMatchEngine: hi+lo operand pairs only (f16x3 with power-of-two scales, or bf16x3), not %r" % (prec,))
        super().__init__(geo, num_classes, pool_kernel, device, prec=prec, prec_bwd=prec, batch_hint=batch_hint)
        self.scaled = (prec == "f16x3")
        eng = self.eng
        self.sel = [_DevPlan(eng.fwd[0].plan, self.device, eng.prec)]
        for li in (1, 2):
            cin, cout, t, h, w = eng.dims[li][:5]
            pl = P.plan_forward_cl("sel%d" % li, 2 * cin, cout, t, h, w, eng.dims[li][11], feat_out=(li == 2))
            pl = P.latency_variant(pl, batch_hint, lambda opts, li=li, cin=cin, cout=cout, t=t, h=h, w=w: P.plan_forward_cl(
                "sel%d" % li, 2 * cin, cout, t, h, w, eng.dims[li][11], feat_out=(li == 2), mtw_options=opts))
            dp = _DevPlan(pl, self.device, eng.prec)
            dp.params.src_split_cc = cin // 8
            dp.params.src_clip_stride4 = pl.clip_stride4 // 2      # each of the two source tensors holds cin channels per clip
            self.sel.append(dp)
        for dp in self.sel:
            dp.params.select = 1
        if self.scaled:
            for dp in self.sel[:2]:
                dp.params.select = 2          # fp32 tangents in slot order: measured, then split at their own scale
            self.eng_fwd = None
        else:
            # the pooling decisions come from an f16x3 forward (operand error ~4e-7, like fp32's own rounding;
            # bf16 pairs are exact to ~1e-5 only and flip near-tied windows far more often than the reference)
            self.eng_fwd = EmbedEngine(geo, prec="f16x3", device=device, chunk=1 << 30, batch_hint=batch_hint)
        self.bwdV = [[_DevPlan(dp.plan, self.device, eng.prec_bwd) for dp in layer] for layer in eng.bwd]
        for layer in self.bwdV:
            for dp in layer:
                dp.params.atomic = 1

    def _forward(self, x, params):
        if self.scaled:      # the engine's own fp16 hi+lo forward: its kept activations and pixel rows ARE the sweeps' operands
            return TrainEngine._forward(self, x, params)
        eng, ef, L, st = self.eng, self.eng_fwd, hip.lib(), hip.stream_ptr(self.device)
        ef.set_weights(params[:6])
        keep_ws, ef._ws = ef._ws, {}
        try:
            feats, saved = ef.forward(x, keep=True)
            (_, nb, am0, am1, am2), = saved
            g = self.geo
            per1 = int(np.prod(eng.fwd[0].plan.out_shape[:-1]))
            per2 = int(np.prod(eng.fwd[1].plan.out_shape[:-1]))
            for name, n in (("act1", nb * per1), ("act2", nb * per2)):
                src = ef._buf(name, (2, n, 8), torch.int16)
                dst = eng._buf(name, (eng.planes, n, 8), torch.int16)
                hip.check(L.vd_resplit_slots(hip.ptr(src[0]), hip.ptr(src[1]), ctypes.c_int64(n * 8), ef.prec, hip.ptr(dst[0]),
                                             hip.ptr(dst[1] if eng.planes == 2 else None), eng.prec, st), "vd_resplit_slots")
            rowp = P.pix_row_pitch(g.width)
            n_slots0 = nb * g.frames * 3 * g.height * (rowp // 8)
            slots0 = eng._buf("slots0", (eng.planes, n_slots0, 8), torch.int16)
            hip.check(L.vd_pix2rows(hip.ptr(x), hip.ptr(None), ctypes.c_int64(nb), g.frames, g.height, g.width, hip.ptr(slots0[0]),
                                     hip.ptr(slots0[1] if eng.planes == 2 else None), eng.prec, st), "vd_pix2rows")
        finally:
            ef._ws = keep_ws
        return feats, nb, (am0, am1, am2)

    def param_grads(self, x, labels, params, mask=None):
        """-> (loss, logits, [8 gradient tensors], state); state feeds ``vjp``."""
        eng = self.eng
        keep_ws, eng._ws = eng._ws, {}
        state = {}
        try:
            loss, logits, g = self.loss_and_grads(x, labels, params, mask, state=state)
            g = [t.clone() for t in g]
        finally:
            state["ws"], eng._ws = eng._ws, keep_ws
        return loss, logits, g, state

    # -- pieces of the adjoint sweep (shared by the fused ``vjp`` and the autograd path of ConvNet3D.forward) ----------
    def _scale_buf(self, name: str) -> torch.Tensor:
        return self.eng._buf(name, (4,), torch.float32)

    def _absmax_scale(self, t: torch.Tensor, name: str, target: float) -> torch.Tensor:
        """[2^k, 2^-k, scratch, -] on the device with max|t| * 2^k in [target / 2, target) (vd_absmax_scale)."""
        scb = self._scale_buf(name)
        hip.check(hip.lib().vd_absmax_scale(hip.ptr(t), ctypes.c_int64(t.numel()), ctypes.c_float(target), hip.ptr(scb),
                                            hip.stream_ptr(self.device)), "vd_absmax_scale")
        return scb

    def _combine(self, a: torch.Tensor, b: Optional[torch.Tensor], mode: int, name: str) -> torch.Tensor:
        """mode 0: [a0 * b0, 1 / (a0 * b0)]; mode 1: [min(a0, b0), 1 / min] (device scalars, vd_scale_combine)."""
        out = self._scale_buf(name)
        hip.check(hip.lib().vd_scale_combine(hip.ptr(a), hip.ptr(b), mode, hip.ptr(out), hip.stream_ptr(self.device)), "vd_scale_combine")
        return out

    def _pack_adjoint(self, W: Sequence[torch.Tensor], V: Sequence[torch.Tensor]) -> None:
        eng = self.eng
        if self.scaled:
            # every V_l at its own power-of-two scale (the accumulating input-gradient programs and the first level's select
            # program); the K-concatenated operands [V_l | W_l] of levels 1 / 2 are packed in the upward sweep, once the range of
            # gbar_l -- which must share V_l's scale -- is known
            self._sv = [self._absmax_scale(V[2 * li], "vscale%d" % li, self.V_TARGET) for li in range(3)]
            self._Vs = [V[2 * li] * self._sv[li][0] for li in range(3)]
            with engine.batched_packs():
                for li in range(3):
                    for dp in eng.bwd[li]:
                        dp.pack(W[2 * li])
                    for dp in self.bwdV[li]:
                        dp.pack(self._Vs[li])
                self.sel[0].pack(self._Vs[0])
            return
        with engine.batched_packs():          # (some twenty programs' operands: one or two launches)
            for li in range(3):
                for dp in eng.bwd[li]:
                    dp.pack(W[2 * li])
                for dp in self.bwdV[li]:
                    dp.pack(V[2 * li])
            self.sel[0].pack(V[0])
            for li in (1, 2):
                self.sel[li].pack(torch.cat([V[2 * li], W[2 * li]], dim=1).contiguous())

    def _sweep_bufs(self, nb: int):
        eng, geo = self.eng, self.geo
        rowp = P.pix_row_pitch(geo.width)
        n_slots0 = nb * geo.frames * 3 * geo.height * (rowp // 8)
        per1 = int(np.prod(eng.fwd[0].plan.out_shape[:-1]))
        per2 = int(np.prod(eng.fwd[1].plan.out_shape[:-1]))
        n1, n2 = nb * per1, nb * per2
        slots0 = eng._buf("slots0", (eng.planes, n_slots0, 8), torch.int16)
        act1 = eng._buf("act1", (eng.planes, n1, 8), torch.int16)
        act2 = eng._buf("act2", (eng.planes, n2, 8), torch.int16)
        gbar1 = eng._buf("gbar1", (eng.planes, n1, 8), torch.int16)
        gbar2 = eng._buf("gbar2", (eng.planes, n2, 8), torch.int16)
        return n_slots0, n1, n2, slots0, act1, act2, gbar1, gbar2

    def _up_sweep(self, nb: int, am, V: Sequence[torch.Tensor], W: Optional[Sequence[torch.Tensor]] = None) -> torch.Tensor:
        """gbar_{l+1} = P_l (conv(a_l, V_l) + vb_l + conv(gbar_l, W_l)), gbar_0 = 0  ->  gbar_3 (nb, num_feat): the tangent
        of the features in the parameter direction V, equally the adjoint of the feature gradient."""
        eng = self.eng
        n_slots0, n1, n2, slots0, act1, act2, gbar1, gbar2 = self._sweep_bufs(nb)
        gbar3 = torch.empty((nb, eng.num_feat), dtype=torch.float32, device=self.device)
        for dp, a, gb in ((self.sel[1], act1, gbar1), (self.sel[2], act2, gbar2)):
            off = gb.data_ptr() - a.data_ptr()
            assert off % 4 == 0
            dp.params.src_split_off4 = off // 4
        if self.scaled:
            L, st = hip.lib(), hip.stream_ptr(self.device)
            g32 = [eng._buf("gbar32_1", (n1 * 8,), torch.float32), eng._buf("gbar32_2", (n2 * 8,), torch.float32)]
            self.sel[0].run(slots0, n_slots0, V[1], g32[0].data_ptr(), 0, am[0], nb, out_scale=self._sv[0][1:])
            self._sg = [None, None, None]
            for li, a, n_a, gb, n_out in ((1, act1, n1, gbar1, n2), (2, act2, n2, gbar2, 0)):
                # the tangent the level below wrote as fp32: its range, the common scale with V_l (the smaller of the two:
                # both stay below fp16's maximum), the split, and only then [V_l * s | W_l] for this level's program
                sg = self._absmax_scale(g32[li - 1], "gscale_up%d" % li, self.V_TARGET)
                s_l = self._combine(sg, self._sv[li], 1, "sscale%d" % li)
                self._sg[li] = s_l
                hip.check(L.vd_split_scaled(hip.ptr(g32[li - 1]), ctypes.c_int64(g32[li - 1].numel()), hip.ptr(s_l), hip.ptr(gb[0]),
                                            hip.ptr(gb[1]), eng.prec, st), "vd_split_scaled")
                self.sel[li].pack(torch.cat([V[2 * li] * s_l[0], W[2 * li]], dim=1).contiguous())
                dst = g32[1].data_ptr() if li == 1 else gbar3.data_ptr()
                self.sel[li].run(a, n_a, V[2 * li + 1], dst, 0, am[li], nb, out_scale=s_l[1:])
            return gbar3
        self.sel[0].run(slots0, n_slots0, V[1], gbar1.data_ptr(), n1, am[0], nb)
        self.sel[1].run(act1, n1, V[3], gbar2.data_ptr(), n2, am[1], nb)
        self.sel[2].run(act2, n2, V[5], gbar3.data_ptr(), 0, am[2], nb)
        return gbar3

    def _down_sweep(self, nb: int, am, abar: torch.Tensor, x: torch.Tensor, hv: Optional[Sequence[torch.Tensor]]) -> torch.Tensor:
        """abar_l = convT(P_l^T abar_{l+1}, W_l) + convT(dz_l, V_l) from abar_3 = ``abar`` down to the pixels; with ``hv`` the
        parameter side (weight / bias adjoints of the three conv levels) is accumulated into hv[0..5]."""
        eng, geo, L, st = self.eng, self.geo, hip.lib(), hip.stream_ptr(self.device)
        _, n1, n2, _, act1, act2, gbar1, gbar2 = self._sweep_bufs(nb)
        dx = torch.empty((nb, geo.frames, geo.channel, geo.height, geo.width), dtype=torch.float32, device=self.device)
        grad, layout = abar, 0
        for li in (2, 1, 0):
            cin, cout, t, h, w, T, OH, OW, To, Ho, Wo, pt = eng.dims[li]
            nslots = nb * (cout // 8) * T * OH * OW
            dy = eng._buf("dy%d" % li, (eng.planes_bwd, nslots, 8), torch.int16)       # dz_l of the first-order pass
            zb = eng._buf("zb%d" % li, (eng.planes_bwd, nslots, 8), torch.int16)
            lo = zb[1] if eng.planes_bwd == 2 else None
            sz = inv_z = inv_dv = inv_gd = None
            if self.scaled:
                # zb_l = P_l^T abar_{l+1} at its own scale; dy_l carries the first-order pass's scale of this level (gscale%d of
                # TrainEngine.feat_backward, kept in the step's workspace), V_l and gbar_l the scales of the upward sweep
                sz = self._absmax_scale(grad, "zscale%d" % li, GRAD_TARGET())
                inv_z = sz[1:]
                gs = self._scale_buf("gscale%d" % li)
                inv_dv = self._combine(gs, self._sv[li], 0, "dvscale%d" % li)[1:]
                if hv and li > 0:
                    inv_gd = self._combine(gs, self._sg[li], 0, "gdscale%d" % li)[1:]
            hip.check(L.vd_unpool_relu_bwd(hip.ptr(grad), hip.ptr(am[li]), ctypes.c_int64(nb), cout, To, Ho, Wo, pt, T, OH,
                                           OW, layout, hip.ptr(zb[0]), hip.ptr(lo), eng.prec_bwd, hip.ptr(sz), st),
                      "vd_unpool_relu_bwd")
            out = dx if li == 0 else eng._buf("ax%d" % li, (nb, t, h, w, cin), torch.float32)
            run_together(eng.bwd[li], zb, nslots, None, out.data_ptr(), 0, None, nb, out_scale=inv_z)
            run_together(self.bwdV[li], dy, nslots, None, out.data_ptr(), 0, None, nb, out_scale=inv_dv)   # (accumulates on top of the stores above)
            if hv:
                # parameter side: z_l = conv(a_l, W_l) + b_l carries zbar_l, and g_{a_l} = convT(dz_l, W_l) carries gbar_l
                op = self._wgrad(li, nb)
                hip.check(L.vd_bias_grad_pooled(hip.ptr(grad), hip.ptr(am[li]), ctypes.c_int64(nb), cout,
                                                ctypes.c_int64(To * Ho * Wo), layout, hip.ptr(hv[2 * li + 1]), st),
                          "vd_bias_grad_pooled")
                if li == 0:
                    op.run(x, True, 0, zb, nslots, hv[0], out_scale=inv_z)
                else:
                    a_l, g_l, n_l = (act1, gbar1, n1) if li == 1 else (act2, gbar2, n2)
                    op.run(a_l, False, n_l, zb, nslots, hv[2 * li], out_scale=inv_z)
                    op.run(g_l, False, n_l, dy, nslots, hv[2 * li], out_scale=inv_gd)
            grad, layout = out, 1
        return dx

    def _views(self, n_tensors: int):
        flat = torch.zeros(sum(self.sizes[:n_tensors]), dtype=torch.float32, device=self.device)
        out, o = [], 0
        for shp, n in zip(self.shapes[:n_tensors], self.sizes[:n_tensors]):
            out.append(flat[o:o + n].view(*shp))
            o += n
        return out

    def head_second_order(self, hs: dict, dlog: torch.Tensor, gbar_feat: torch.Tensor, v_w: torch.Tensor, v_b: torch.Tensor,
                          want_params: bool, hessian: bool):
        """Second-order pass through the head.  ``hessian``: fused with the Hessian of the mean cross-entropy (dlog must
        then be that loss's own logit gradient) -- the trainers' path; otherwise the adjoint of ``dlog`` is returned
        for the caller's loss to chain through.  -> (abar_feats, wbar, bbar, dlogbar)."""
        L, st = hip.lib(), hip.stream_ptr(self.device)
        nb = int(dlog.shape[0])
        kt, kh, kw = self.pool_kernel
        abar = torch.empty((nb, self.eng.num_feat), dtype=torch.float32, device=self.device)
        wbar = torch.zeros((self.K, self.C), dtype=torch.float32, device=self.device) if want_params else None
        bbar = torch.zeros(self.K, dtype=torch.float32, device=self.device) if want_params else None
        dlogbar = None if hessian else torch.empty((nb, self.K), dtype=torch.float32, device=self.device)
        hip.check(L.vd_head_second_order(hip.ptr(hs["logits"] if hessian else None), hip.ptr(dlog), hip.ptr(hs["amt"]),
                                         hip.ptr(hs["dropped"]), hip.ptr(hs["mask"]), hip.ptr(hs["wl"]), hip.ptr(v_w),
                                         hip.ptr(v_b), hip.ptr(gbar_feat), ctypes.c_int64(nb), self.C, self.To, self.Ho, self.Wo,
                                         kt, kh, kw, self.K, hip.ptr(abar), hip.ptr(wbar), hip.ptr(bbar), hip.ptr(dlogbar), st),
                  "vd_head_second_order")
        return abar, wbar, bbar, dlogbar

    def vjp(self, state: dict, v: Sequence[Optional[torch.Tensor]], params: Sequence[torch.Tensor],
            param_adjoint: bool = False):
        """d (sum_i <v_i, g_i(x)>) / dx for the ``param_grads`` call that produced ``state``.  With
        ``param_adjoint`` also d (sum_i <v_i, g_i>) / d params -- the Hessian-vector product H v of the
        CE loss w.r.t. the parameters (MTT's unrolled inner loop) -- returned as (dx, [8 tensors])."""
        eng = self.eng
        nb, am = state["nb"], state["am"]
        W = [p.detach().to(self.device, torch.float32).contiguous() for p in params]
        V = [torch.zeros_like(w) if t is None else t.detach().to(self.device, torch.float32).contiguous().view_as(w)
             for t, w in zip(v, W)]
        keep_ws, eng._ws = eng._ws, state["ws"]
        try:
            self._pack_adjoint(W, V)
            hv = self._views(8) if param_adjoint else None
            gbar3 = self._up_sweep(nb, am, V, W)
            abar, wbar, bbar, _ = self.head_second_order(state, state["dlog"], gbar3, V[6].reshape(self.K, self.C).contiguous(),
                                                         V[7], param_adjoint, hessian=True)
            if hv:
                hv[6].view(self.K, self.C).copy_(wbar)
                hv[7].copy_(bbar)
            dx = self._down_sweep(nb, am, abar, state["x"], hv)
        finally:
            eng._ws = keep_ws
        return (dx, hv) if param_adjoint else dx

    # -- autograd path: every piece as its own call with caller-held state (networks._FeatFunction / _HeadFunction) --------
    _KEEP = ("act1", "act2", "slots0")

    def ag_feat_forward(self, x: torch.Tensor, params6: Sequence[torch.Tensor]):
        """features (B, num_feat) of the clips with the activations / arg-max of all levels kept -> (feats, fstate)."""
        eng = self.eng
        x = x.detach().to(self.device, torch.float32).contiguous()
        keep_ws, eng._ws = eng._ws, {}
        try:
            if self.scaled:      # (the fused entry, loss_and_grads, packs the engine's forward operands itself)
                eng.set_weights([p.detach() for p in params6])
            feats, nb, am = self._forward(x, [p.detach() for p in params6])
            ws = eng._ws
        finally:
            eng._ws = keep_ws
        return feats, dict(ws=ws, nb=nb, am=am, x=x)

    def ag_feat_backward(self, fs: dict, g_feat: torch.Tensor, params6: Sequence[torch.Tensor], need_dx: bool,
                         need_params: bool, keep: bool):
        """First-order backward of the conv levels: -> (dx or None, [6 parameter gradients] or None, bstate).  ``keep``:
        the first-order gradients at the conv outputs stay in bstate for ``ag_feat_second_order`` (create_graph)."""
        eng = self.eng
        ws = {k: v for k, v in fs["ws"].items() if k in self._KEEP}
        if not keep:
            ws.update(getattr(self, "_scratch", {}))
        W = [p.detach().to(self.device, torch.float32).contiguous() for p in params6]
        keep_ws, eng._ws = eng._ws, ws
        try:
            with engine.batched_packs():
                for li in ((0, 1, 2) if need_dx else (1, 2)):
                    for dp in eng.bwd[li]:
                        dp.pack(W[2 * li])
            g = self._views(6) if need_params else None
            geo = self.geo
            dx = torch.empty((fs["nb"], geo.frames, geo.channel, geo.height, geo.width), dtype=torch.float32,
                             device=self.device) if need_dx else None
            self.feat_backward(fs["x"], fs["nb"], fs["am"], g_feat.detach().to(torch.float32).contiguous(), g, dx, keep_dense=keep)
        finally:
            eng._ws = keep_ws
        if not keep:
            self._scratch = {k: v for k, v in ws.items() if k not in self._KEEP}
            return dx, g, None
        return dx, g, dict(ws=ws)

    def ag_feat_second_order(self, fs: dict, bs: dict, v6: Sequence[Optional[torch.Tensor]], params6: Sequence[torch.Tensor],
                             need_params: bool):
        """Adjoints of an ``ag_feat_backward`` call for the adjoints ``v6`` of its six parameter gradients:
        -> (adjoint of g_feat (nb, num_feat), adjoint of x, [6 parameter adjoints] or None)."""
        eng = self.eng
        nb, am = fs["nb"], fs["am"]
        W = [p.detach().to(self.device, torch.float32).contiguous() for p in params6]
        V = [torch.zeros_like(w) if t is None else t.detach().to(self.device, torch.float32).contiguous().view_as(w)
             for t, w in zip(v6, W)]
        keep_ws, eng._ws = eng._ws, bs["ws"]
        try:
            self._pack_adjoint(W, V)
            hv = self._views(6) if need_params else None
            gbar3 = self._up_sweep(nb, am, V, W)
            abar = torch.zeros((nb, eng.num_feat), dtype=torch.float32, device=self.device)
            xbar = self._down_sweep(nb, am, abar, fs["x"], hv)
        finally:
            eng._ws = keep_ws
        return gbar3, xbar, hv
