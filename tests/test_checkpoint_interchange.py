"""files written by the REFERENCE's own save code (tools/gen_golden.py g14 -> tests/golden/f4/) load
through video_distillation_amd.checkpoint / utils, value for value.  (The reverse direction -- files written by
checkpoint.py loading into the reference's ModuleList[Conv3DNet] / MTT buffer reader -- is asserted by the generator,
which runs where the reference is importable.)"""
import os
import shutil

import numpy as np
import pytest
import torch

from video_distillation_amd import checkpoint, distill, utils

F4 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "f4")


def test_hallucinator_state_dict_from_reference():
    z = np.load(os.path.join(F4, "values.npz"))
    pairs = checkpoint.load_hallucinators(os.path.join(F4, "hal_7.pt"))
    assert len(pairs) == 2
    for i, (w, b) in enumerate(pairs):
        np.testing.assert_array_equal(w.numpy(), z["hal_w"][i])
        np.testing.assert_array_equal(b.numpy(), z["hal_b"][i])
    # and straight into this package's modules, strict keys (what the reference's evaluation scripts do)
    hals = torch.nn.ModuleList([utils.Conv3DNet(img_size=16) for _ in range(2)])
    hals.load_state_dict(torch.load(os.path.join(F4, "hal_7.pt")))
    np.testing.assert_array_equal(hals[1].encoder.weight.detach().numpy(), z["hal_w"][1])


def test_memories_and_images_from_reference():
    z = np.load(os.path.join(F4, "values.npz"))
    np.testing.assert_array_equal(checkpoint.load_static(os.path.join(F4, "static_memory.pt")).numpy(), z["static"])
    dyn = torch.load(os.path.join(F4, "dynamic_7.pt"))
    assert tuple(dyn.shape) == (6, 4, 1, 16, 16)                    # (C*dpc, T, 1, H, W): dynamic_syn.flatten(0, 1)
    np.testing.assert_array_equal(dyn.reshape(z["dynamic"].shape).numpy(), z["dynamic"])
    np.testing.assert_array_equal(torch.load(os.path.join(F4, "images_baseline_7.pt")).numpy(), z["image_syn"])
    with pytest.raises(KeyError):
        checkpoint.load_static(os.path.join(F4, "images_7.pt"))     # a bare tensor is not a static-memory file


def test_expert_buffer_from_reference(tmp_path):
    z = np.load(os.path.join(F4, "values.npz"))
    shutil.copy(os.path.join(F4, "replay_buffer_0.pt"), tmp_path / "replay_buffer_0.pt")
    buf = checkpoint.load_expert_buffers(str(tmp_path))
    assert len(buf) == 1 and len(buf[0]) == 2 and len(buf[0][0]) == 8
    l1 = np.array([[float(p.double().abs().sum()) for p in st] for st in buf[0]])
    np.testing.assert_allclose(l1, z["traj_l1"], rtol=1e-12)
    shapes = [list(p.shape) + [0] * (5 - p.dim()) for p in buf[0][0]]
    assert shapes == z["traj_shapes"].tolist()
    flat = distill.flatten_params(buf[0][1])                        # what MTTTrainer.step builds its target from
    assert flat.numel() == sum(int(np.prod([d for d in s if d])) for s in shapes)
    with pytest.raises(AssertionError):
        checkpoint.load_expert_buffers(str(tmp_path / "nothing_here"))


def test_writer_reader_round_trip(tmp_path):
    z = np.load(os.path.join(F4, "values.npz"))
    dyn = torch.tensor(z["dynamic"])
    checkpoint.save_s2d(str(tmp_path), 11, dyn, list(torch.tensor(z["hal_w"])), list(torch.tensor(z["hal_b"])), best=True)
    ref = torch.load(os.path.join(F4, "hal_7.pt"))
    mine = torch.load(tmp_path / "hal_11.pt")
    assert list(mine.keys()) == list(ref.keys()) and all(torch.equal(mine[k], ref[k]) for k in ref)
    assert torch.equal(torch.load(tmp_path / "dynamic_best.pt"), torch.load(os.path.join(F4, "dynamic_7.pt")))
