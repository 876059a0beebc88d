"""Device half of the dataset preload: vd_frames_normalize and dataset.preload against the host transform (bit-equal),
and RealPool.from_dataset's class bookkeeping."""
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
UCF = os.path.join(GOLD, "frames", "UCF101")


def _seed():
    np.random.seed(5); random.seed(7); torch.manual_seed(3)


@pytest.mark.parametrize("n,h,w", [(5, 112, 112), (3, 64, 64), (2, 7, 5), (1, 1, 1), (0, 8, 8)])
def test_frames_normalize_is_bit_equal_to_the_host_transform(n, h, w):
    from video_distillation_amd import dataset as D
    g = torch.Generator().manual_seed(n * 100 + h)
    u8 = torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8, generator=g)
    if n:
        u8.view(-1)[:256 if u8.numel() >= 256 else u8.numel()] = torch.arange(min(256, u8.numel()), dtype=torch.uint8)   # every byte value
    tf = D.FrameTransform((h, w), stored=(h, w))
    want = tf.normalise(u8)
    out = torch.full((n, 3, h, w), float("nan"), device="cuda:0")
    D.frames_normalize(u8.to("cuda:0"), out, tf.mean.tolist(), tf.std.tolist())
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), want)


def test_frames_normalize_unaligned_views_take_the_scalar_path():
    from video_distillation_amd import dataset as D
    tf = D.FrameTransform((8, 8), stored=(8, 8))
    base = torch.randint(0, 256, (2 * 8 * 8 * 3 + 1,), dtype=torch.uint8)
    u8 = base[1:].view(2, 8, 8, 3)                        # 1-byte offset: not 4-byte aligned on the device either
    dev = base.to("cuda:0")[1:].view(2, 8, 8, 3)
    out = torch.empty((2, 3, 8, 8), device="cuda:0")
    D.frames_normalize(dev, out, tf.mean.tolist(), tf.std.tolist())
    assert torch.equal(out.cpu(), tf.normalise(u8))


def test_preload_equals_stacked_items_and_feeds_the_pool():
    from video_distillation_amd import dataset as D, distill
    ds = D.UCF101(UCF, "train")
    _seed()
    want = torch.stack([ds[i][0] for i in range(len(ds))])
    ds2 = D.UCF101(UCF, "train")
    _seed()
    clips, labels = D.preload(ds2, "cuda:0", workers=3, chunk=2)          # 3 items, chunks of 2: both staging buffers + a tail
    assert labels.tolist() == ds.labels and clips.shape == (3, 16, 3, 112, 112)
    assert torch.equal(clips.cpu(), want)
    # pool: class-major order, counts / offsets per global class, only the owned classes resident
    ds3 = D.UCF101(UCF, "train")
    _seed()
    pool = distill.RealPool.from_dataset(ds3, 2, [1], "cuda:0")
    assert pool.counts == [2, 1] and pool.offsets[1] == 0 and pool.clips.shape[0] == 1
    idx = distill.sample_real_indices(0, pool.counts, pool.offsets, 1, [1])
    assert idx.tolist() == [0]


@pytest.mark.parametrize("frames,shape", [(16, (3, 16, 3, 112, 112)), (1, (3, 3, 112, 112))])
def test_preload_of_still_frame_items(frames, shape):
    from video_distillation_amd import dataset as D
    ds = D.staticUCF50(UCF, "train", frames=frames, split_num=4, split_id=1, split_mode='feature')
    _seed()
    want = torch.stack([ds[i][0] for i in range(len(ds))])
    ds2 = D.staticUCF50(UCF, "train", frames=frames, split_num=4, split_id=1, split_mode='feature')
    _seed()
    clips, labels = D.preload(ds2, "cuda:0", workers=2, chunk=2)
    assert clips.shape == shape and labels.tolist() == ds.labels
    assert torch.equal(clips.cpu(), want)


def test_dm_trainer_runs_on_a_preloaded_pool():
    """The decoded frames go through the path: one DM step on the pool built from the JPEG tree, finite and decreasing."""
    from video_distillation_amd import dataset as D, distill, plan
    ds = D.UCF101(UCF, "train")
    _seed()
    pool = distill.RealPool.from_dataset(ds, 2, [0, 1], "cuda:0")
    geo = plan.NetGeometry(16, 112, 112)
    be = distill.HipBackend(geo, "cuda:0")
    tr = distill.DMTrainer(be, pool, 2, 1, 1, lr_img=1.0, image_syn=torch.randn(2, 16, 3, 112, 112, device="cuda:0"))
    l0 = float(tr.step(0))
    assert np.isfinite(l0) and l0 > 0
