"""Measurement tool (not a test): fixture G3's pixel-update error for every combination of operand precisions.
    python tests/parity_modes_tool.py            # on a GPU box
Lives under tests/ because it uses the oracle's reference-identical weight initialisation."""
import itertools
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_gpu_api import _FixedNetBackend, _rel, randn  # noqa: E402


def g3(mode, value_pass=True):
    from video_distillation_amd import distill, plan
    import video_distillation_amd.distill as D
    z = np.load(os.path.join(ROOT, "tests", "golden", "g3_dm_steps.npz"))
    geo = plan.NetGeometry(8, 64, 64)
    inner = distill.HipBackend(geo, "cuda:0", **mode)
    if not value_pass:
        inner.weight_format = None
    be = _FixedNetBackend(inner, z["net_seeds"])
    (syn,) = randn(z["syn_seed"], (3, 8, 3, 64, 64))
    reals = [randn(z["real_seeds"][it], *[(4, 8, 3, 64, 64)] * 3) for it in range(2)]
    pool = types.SimpleNamespace(clips=torch.cat([torch.cat(r) for r in reals]).cuda(), counts=[4, 4, 4], offsets=[0, 4, 8])
    tr = distill.DMTrainer(be, pool, 3, 1, 4, lr_img=float(z["lr"]), momentum=float(z["momentum"]), image_syn=syn.cuda())
    orig = D.sample_real_indices
    try:
        D.sample_real_indices = lambda it_, counts, offsets, b, classes: np.concatenate(
            [offsets[c] + np.arange(4) for c in classes]).astype(np.int64)
        loss = float(tr.step(0))
    finally:
        D.sample_real_indices = orig
    upd = _rel((tr.image_syn.cpu() - syn)[:, ::2, :, ::4, ::4], torch.tensor(z["syn1"]) - syn[:, ::2, :, ::4, ::4])
    return abs(loss / float(z["losses"][0]) - 1), upd


if __name__ == "__main__":
    for pr, ps, pb in itertools.product(("f16", "f16x3"), ("f16x3",), ("f16", "f16x3", "bf16x3")):
        for vp in ((True, False) if pr == "f16" else (True,)):
            l, u = g3(dict(prec_real=pr, prec_syn=ps, prec_bwd=pb), vp)
            print("real %-6s syn %-6s bwd %-6s value_pass %-5s: loss rel %.2e  first update rel-l2 %.2e" % (pr, ps, pb, vp, l, u))
