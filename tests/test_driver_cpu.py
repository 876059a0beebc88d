"""Host logic of the DM driver and of the on-disk formats, on the CPU with the oracle backend."""
import os

import numpy as np
import torch

from tests.cpu_backend import OracleBackend
from video_distillation_amd import checkpoint, run_dm


def test_checkpoint_formats_roundtrip(tmp_path):
    C, ipc, dpc = 3, 2, 2
    syn = torch.randn(C * ipc, 4, 3, 8, 8)
    checkpoint.save_images(str(tmp_path), 7, syn, best=True)
    for name in ("images_7.pt", "images_best.pt"):
        got = torch.load(os.path.join(tmp_path, name))
        assert got.shape == (C * ipc, 4, 3, 8, 8) and torch.equal(got, syn)       # reference layout (C*ipc,T,3,H,W)
    dyn = torch.randn(C, dpc, 4, 1, 8, 8)
    w, b = torch.randn(3, 4, 3, 3, 3), torch.randn(3)
    checkpoint.save_s2d(str(tmp_path), 7, dyn, [w], [b], best=True)
    flat = torch.load(os.path.join(tmp_path, "dynamic_7.pt"))
    assert flat.shape == (C * dpc, 4, 1, 8, 8) and torch.equal(flat.view_as(dyn), dyn)
    state = torch.load(os.path.join(tmp_path, "hal_7.pt"))
    assert sorted(state.keys()) == ["0.encoder.bias", "0.encoder.weight"]        # ModuleList[Conv3DNet].state_dict()
    (w2, b2), = checkpoint.load_hallucinators(os.path.join(tmp_path, "weights_best.pt"))
    assert torch.equal(w2, w) and torch.equal(b2, b)
    torch.save({"image": torch.ones(C * 2, 3, 8, 8)}, os.path.join(tmp_path, "static.pt"))
    assert checkpoint.load_static(os.path.join(tmp_path, "static.pt")).shape == (C * 2, 3, 8, 8)


def test_dm_driver_runs_and_logs_reference_keys(tmp_path):
    g = torch.Generator().manual_seed(0)
    C, per = 3, 5
    clips = torch.randn(C * per, 8, 3, 64, 64, generator=g)
    labels = torch.arange(C).repeat_interleave(per)[torch.randperm(C * per, generator=g)]
    data = os.path.join(tmp_path, "toy.pt")
    torch.save({"clips": clips, "labels": labels, "test_clips": clips[:4], "test_labels": labels[:4]}, data)
    args = run_dm.build_parser().parse_args([
        "--dataset", "toy", "--data_file", data, "--ipc", "1", "--Iteration", "2", "--eval_it", "2", "--num_eval", "1",
        "--epoch_eval_train", "1", "--batch_real", "3", "--frames", "8", "--im_size", "64", "--lr_img", "0.1",
        "--save_path", str(tmp_path), "--no_eval"])
    log = []
    tr = run_dm.run(args, backend=OracleBackend(), log=log)
    losses = [r["Loss"] for r in log if "Loss" in r]
    assert len(losses) == 2 and all(np.isfinite(losses)) and losses[0] > 0          # it 0 and the final iteration
    assert tr.image_syn.shape == (C, 8, 3, 64, 64) and tr.steps_done == 3
    # init 'real': synthetic clips start as the first clip of each class (sorted pool), then move
    order = torch.argsort(labels, stable=True)
    first = clips[order][[0, per, 2 * per]]
    assert not torch.equal(tr.image_syn, first) and float((tr.image_syn - first).abs().max()) < 1.0


def test_expert_buffer_format_roundtrip(tmp_path):
    """replay_buffer_N.pt: list[expert] of list[epoch] of the 8 parameter tensors; first free N;
    loader concatenates all files and asserts when none exists (distill_baseline.py:116-133)."""
    import pytest
    from oracle import ref_cpu as R
    traj = [[R.init_params(s + e, 3, 4) for e in range(3)] for s in (10, 20)]
    p0 = checkpoint.save_expert_buffer(str(tmp_path), traj[:1])
    p1 = checkpoint.save_expert_buffer(str(tmp_path), traj[1:])
    assert os.path.basename(p0) == "replay_buffer_0.pt" and os.path.basename(p1) == "replay_buffer_1.pt"
    raw = torch.load(p0)
    assert isinstance(raw, list) and len(raw) == 1 and len(raw[0]) == 3 and len(raw[0][0]) == 8
    assert [tuple(t.shape) for t in raw[0][0]] == [tuple(s) for s in R.param_shapes(3, 4)]
    both = checkpoint.load_expert_buffers(str(tmp_path))
    assert len(both) == 2 and torch.equal(both[1][2][6], traj[1][2][6])
    with pytest.raises(AssertionError):
        checkpoint.load_expert_buffers(os.path.join(tmp_path, "nope"))
