"""``ReparamModule``: a module whose parameters are views of ONE flat parameter vector.

Same surface as the reference's ``reparam_module.ReparamModule`` (reparam_module.py:9-177), which MTT's unrolled
student loop drives as ``student_net(x, flat_param=student_params[-1])`` (distill_baseline.py:243-250,
distill_s2d_ms.py:256-262): constructor flattens ``module``'s parameters in ``named_parameters`` order into the
registered parameter ``flat_param`` (the order expert buffers are stored in, buffer.py:75-104), the sub-modules keep
plain-attribute views, and ``forward`` / ``embed`` take an optional replacement ``flat_param`` (squeezed, so the
``unsqueeze(0).expand(n_gpu, -1)`` rows DataParallel scatters arrive as a vector) and optional ``buffers``.

Nothing here computes: the wrapped ``ConvNet3D`` reads its tensors by attribute, so a forward under a replacement
vector runs the twice-differentiable HIP path (networks._FeatFunction / _HeadFunction) on slices of that vector and
autograd accumulates the eight slice gradients back into it.  ``trace`` (TorchScript tracing) is not offered: the HIP
path is already a fixed sequence of launches.
"""
from __future__ import annotations

from contextlib import contextmanager

import torch
import torch.nn as nn


def _resolve(root: nn.Module, path: str) -> nn.Module:
    mod = root
    for part in (path.split('.') if path else ()):
        mod = getattr(mod, part)
    return mod


class ReparamModule(nn.Module):
    def __init__(self, module: nn.Module):
        super().__init__()
        self.module = module
        owners, seen, shared, flat_parts, numels, shapes = [], {}, [], [], [], []
        for mod_name, mod in self.named_modules():
            for pname, p in mod.named_parameters(recurse=False):
                if p is None:
                    continue
                if p in seen:                       # a parameter registered twice keeps ONE slot
                    shared.append((mod_name, pname) + seen[p])
                    continue
                seen[p] = (mod_name, pname)
                owners.append((mod_name, pname))
                flat_parts.append(p.detach().reshape(-1))
                numels.append(p.numel())
                shapes.append(p.size())
        assert len({t.dtype for t in flat_parts}) <= 1, "expects all parameters in module to have same dtype"
        self._param_infos = tuple(owners)
        self._shared_param_infos = tuple(shared)
        self._param_numels = tuple(numels)
        self._param_shapes = tuple(shapes)
        self.register_parameter('flat_param', nn.Parameter(torch.cat(flat_parts, 0)))
        self.param_numel = self.flat_param.numel()
        for mod_name, pname in self._param_infos + tuple(s[:2] for s in self._shared_param_infos):
            delattr(_resolve(self, mod_name), pname)          # no longer nn.Parameters of the sub-modules
        self._unflatten_param(self.flat_param)
        self._buffer_infos = tuple((mn, bn, b) for mn, m in self.named_modules()
                                   for bn, b in m.named_buffers(recurse=False) if b is not None)

    # -- views -------------------------------------------------------------------------------------------------------
    def _views_of(self, flat_param):
        return [t.view(shp) for t, shp in zip(flat_param.split(self._param_numels), self._param_shapes)]

    def _install(self, views) -> None:
        for (mod_name, pname), v in zip(self._param_infos, views):
            setattr(_resolve(self, mod_name), pname, v)
        for mod_name, pname, src_mod, src_name in self._shared_param_infos:
            setattr(_resolve(self, mod_name), pname, getattr(_resolve(self, src_mod), src_name))

    def _unflatten_param(self, flat_param) -> None:
        self._install(self._views_of(flat_param))

    def clear_views(self) -> None:
        for mod_name, pname in self._param_infos:
            setattr(_resolve(self, mod_name), pname, None)

    @contextmanager
    def unflattened_param(self, flat_param):
        previous = [getattr(_resolve(self, mn), pn) for mn, pn in self._param_infos]
        self._unflatten_param(flat_param)
        try:
            yield
        finally:
            self._install(previous)

    @contextmanager
    def replaced_buffers(self, buffers):
        for (mod_name, bname, _), new in zip(self._buffer_infos, buffers):
            setattr(_resolve(self, mod_name), bname, new)
        try:
            yield
        finally:
            for mod_name, bname, old in self._buffer_infos:
                setattr(_resolve(self, mod_name), bname, old)

    def trace(self, *a, **kw):
        raise NotImplementedError("ReparamModule.trace: TorchScript tracing is not part of the HIP path")

    # -- calls ---------------------------------------------------------------------------------------------------------
    def _call(self, fn_name: str, inputs, kwinputs, flat_param, buffers):
        if flat_param is None:
            flat_param = self.flat_param
        else:
            flat_param = torch.squeeze(flat_param)
        fn = self.module if fn_name == 'forward' else getattr(self.module, fn_name)
        with self.unflattened_param(flat_param):
            if buffers is None:
                return fn(*inputs, **kwinputs)
            with self.replaced_buffers(tuple(buffers)):
                return fn(*inputs, **kwinputs)

    def _forward_with_param(self, flat_param, *inputs, **kwinputs):
        return self._call('forward', inputs, kwinputs, flat_param, None)

    def _forward_with_param_and_buffers(self, flat_param, buffers, *inputs, **kwinputs):
        return self._call('forward', inputs, kwinputs, flat_param, buffers)

    def forward(self, *inputs, flat_param=None, buffers=None, **kwinputs):
        return self._call('forward', inputs, kwinputs, flat_param, buffers)

    def embed(self, *inputs, flat_param=None, buffers=None, **kwinputs):
        return self._call('embed', inputs, kwinputs, flat_param, buffers)
