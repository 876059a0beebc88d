"""Frame-folder datasets (video_distillation_amd/dataset.py) against fixture G15: the committed JPEG tree under
tests/golden/frames and what the reference's dataset classes returned for it under fixed generator seeds
(tools/gen_golden.py::g15).  The preload's device half is tested in the gpu tier."""
import os
import random

import numpy as np
import pytest
import torch

from video_distillation_amd import dataset as D

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
UCF = os.path.join(GOLD, "frames", "UCF101")
KIN = os.path.join(GOLD, "frames", "kinetics_64x64x8")


@pytest.fixture(scope="module")
def g15():
    return np.load(os.path.join(GOLD, "g15_frame_datasets.npz"))


def _seed():
    np.random.seed(5); random.seed(7); torch.manual_seed(3)


def _check_item(g, tag, k, x, y):
    assert int(g["%s_%d_label" % (tag, k)]) == y
    np.testing.assert_array_equal(g["%s_%d_probe" % (tag, k)], x[:, :, ::16, ::16].numpy())      # bit-equal fp32
    sums = g["%s_%d_sums" % (tag, k)]
    assert float(x.double().sum()) == pytest.approx(sums[0], rel=1e-12, abs=1e-9)
    assert float((x.double() ** 2).sum()) == pytest.approx(sums[1], rel=1e-12)


@pytest.mark.parametrize("tag,make,passes", [
    ("ucf_train", lambda: D.UCF101(UCF, "train"), 2),
    ("ucf_test", lambda: D.UCF101(UCF, "test"), 2),
    ("hmdb_train", lambda: D.HMDB51(UCF, "train"), 1),
    ("minihmdb_train", lambda: D.miniHMDB51(UCF, "train"), 1),
    ("mini_train", lambda: D.miniUCF101(UCF, "train"), 1),
    ("mini_seg", lambda: D.miniUCF101(UCF, "train", sample="split-random"), 1),
])
def test_window_datasets_match_reference_items(g15, tag, make, passes):
    ds = make()
    np.testing.assert_array_equal(g15["%s_labels" % tag], np.array(ds.labels))
    assert ds.targets is ds.labels
    _seed()
    k = 0
    for _ in range(passes):
        for i in range(len(ds)):
            d = ds.draw(i)
            want = g15["%s_%d_frames" % (tag, k)]
            assert [int(os.path.basename(f)[5:11]) for f in d.files] == want.tolist()
            x = ds.transform.normalise(torch.from_numpy(ds.read_u8(d)))
            assert x.shape == (16, 3, 112, 112) and x.dtype == torch.float32
            _check_item(g15, tag, k, x, ds.labels[i])
            k += 1
    assert k == int(g15["%s_count" % tag])


def test_getitem_is_draw_plus_read(g15):
    ds = D.UCF101(UCF, "train")
    _seed()
    x, y = ds[0]
    _check_item(g15, "ucf_train", 0, x, y)


def test_kinetics_listing_and_items(g15):
    for split in ("train", "val"):
        ds = D.Kinetics400(KIN, split)
        assert [os.path.basename(d) for d in ds.video_dirs] == g15["kin_%s_dirs" % split].tolist()
        np.testing.assert_array_equal(g15["kin_%s_labels" % split], np.array(ds.labels))
        assert ds.skipped == (1 if split == "train" else 0)            # the 5-frame clip is not a sample
        for i in range(len(ds)):
            x, y = ds[i]
            assert x.shape == (8, 3, 64, 64)
            # frames are stacked in directory order on both sides; compare by file name
            mine = os.listdir(ds.video_dirs[i])
            ref_names = g15["kin_%s_%d_names" % (split, i)].tolist()
            ref_probe, ref_sums = g15["kin_%s_%d_probe" % (split, i)], g15["kin_%s_%d_sums" % (split, i)]
            for t, name in enumerate(mine):
                r = ref_names.index(name)
                np.testing.assert_array_equal(ref_probe[r], x[t, :, ::8, ::8].numpy())
                assert float(x[t].double().sum()) == pytest.approx(float(ref_sums[r]), rel=1e-12, abs=1e-9)


def test_resize_crop_branch_and_generator_order():
    """64x64 targets: Resize((100, 80)) + RandomCrop per frame (utils.py:164-169).  torchvision is not in this image, so
    this branch has no reference-generated pin; what is checked is the draw order (start, flip, then row/column per frame
    from torch's generator) and that the pixels are PIL's bilinear resize cut at the drawn origin."""
    from PIL import Image
    tf = D.FrameTransform((64, 64))
    ds = D.UCF101(UCF, "train", tf)
    _seed()
    d = ds.draw(1)
    _seed()
    length = len(os.listdir(ds.video_dirs[1]))
    skip = length // 16
    start = int(np.random.randint(1, length - 15 * skip))
    flip = random.random() > 0.5
    crops = [(int(torch.randint(0, 37, (1,))), int(torch.randint(0, 17, (1,)))) for _ in range(16)]
    assert d.flip == flip and d.crops == crops
    assert d.files[0].endswith("frame%06d.jpg" % start) and d.files[1].endswith("frame%06d.jpg" % (start + skip))
    u8 = ds.read_u8(d)
    assert u8.shape == (16, 64, 64, 3)
    im = Image.open(d.files[3])
    if flip:
        im = im.transpose(Image.FLIP_LEFT_RIGHT)
    i, j = crops[3]
    want = np.asarray(im.resize((80, 100), Image.BILINEAR))[i:i + 64, j:j + 64]
    np.testing.assert_array_equal(u8[3], want)


def test_get_dataset_tuple_and_unknown_name(tmp_path):
    root = os.path.join(GOLD, "frames")
    channel, im_size, num_classes, class_names, mean, std, dst_train, dst_test, testloader = D.get_dataset("miniUCF101", root)
    assert (channel, im_size, num_classes, class_names) == (3, (112, 112), 50, None)
    assert mean == [0.485, 0.456, 0.406] and std == [0.229, 0.224, 0.225]
    assert len(dst_train) == 3 and len(dst_test) == 2 and testloader.batch_size == 64
    xb, yb = next(iter(testloader))
    assert xb.shape == (2, 16, 3, 112, 112) and yb.tolist() == dst_test.labels
    with pytest.raises(SystemExit):
        D.get_dataset("CIFAR10", root)
    with pytest.raises(AssertionError):
        D.get_dataset("HMDB51", root)             # no HMDB51 directory under this root
    assert D.indices_class(dst_train.labels, 2) == [[0, 2], [1]]


def test_preload_needs_the_device():
    with pytest.raises(RuntimeError):
        D.preload(D.UCF101(UCF, "train"), "cpu")


@pytest.mark.parametrize("own_generator", [False, True])
def test_device_batches_follow_the_dataloader_shuffle(own_generator):
    x = torch.arange(11, dtype=torch.float32).view(11, 1)
    y = torch.arange(11)
    for shuffle in (True, False):
        g1 = torch.Generator().manual_seed(31) if own_generator else None
        g2 = torch.Generator().manual_seed(31) if own_generator else None
        torch.manual_seed(17)
        ref = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(x, y), batch_size=4, shuffle=shuffle, generator=g1)
        want = [[b[1].tolist() for b in ref] for _ in range(3)]           # three epochs
        after_ref = torch.rand(1)
        torch.manual_seed(17)
        mine = D.DeviceBatches(x, y, 4, shuffle=shuffle, generator=g2)
        got = [[b[1].tolist() for b in mine] for _ in range(3)]
        assert got == want and len(mine) == 3
        assert torch.equal(torch.rand(1), after_ref)                      # the global generator advanced identically
        assert all(torch.equal(b[0].view(-1), b[1].float()) for b in mine)


# ---------------------------------------------------------------------------------------------------------------------
# still-frame families (fixture G17, tools/gen_golden.py::g17)
# ---------------------------------------------------------------------------------------------------------------------
SSV2 = os.path.join(GOLD, "frames", "SSv2_64x8")


@pytest.fixture(scope="module")
def g17():
    return np.load(os.path.join(GOLD, "g17_still_datasets.npz"))


@pytest.mark.parametrize("tag,make,shape", [
    ("shmdb_train", lambda: D.staticHMDB51(UCF, "train"), (16, 3, 112, 112)),
    ("shmdb_test_image", lambda: D.staticHMDB51(UCF, "test", frames=1), (3, 112, 112)),
    ("sucf_train", lambda: D.staticUCF101(UCF, "train"), (16, 3, 112, 112)),
    ("sucf_part1of3", lambda: D.staticUCF101(UCF, "train", frames=4, split_num=3, split_id=1), (4, 3, 112, 112)),
    ("sucf_part_wraps", lambda: D.staticUCF101(UCF, "test", frames=1, split_num=2, split_id=2), (3, 112, 112)),
    ("s50_mean", lambda: D.staticUCF50(UCF, "train", frames=2, split_num=4, split_id=3, split_mode='mean'), (2, 3, 112, 112)),
] + [("s50_feature%d" % k, (lambda k=k: D.staticUCF50(UCF, "train", frames=1, split_num=4, split_id=k, split_mode='feature')), (3, 112, 112))
     for k in range(4)])
def test_static_datasets_match_reference_items(g17, tag, make, shape):
    ds = make()
    np.testing.assert_array_equal(g17["%s_labels" % tag], np.array(ds.labels))
    _seed()
    k = 0
    for _ in range(2):
        for i in range(len(ds)):
            x, y = ds[i]
            assert x.shape == shape and x.dtype == torch.float32
            assert ds.start[i] == int(g17["%s_%d_frame" % (tag, k)])
            assert int(g17["%s_%d_label" % (tag, k)]) == y
            np.testing.assert_array_equal(g17["%s_%d_probe" % (tag, k)], x[..., ::16, ::16].numpy())
            sums = g17["%s_%d_sums" % (tag, k)]
            assert float(x.double().sum()) == pytest.approx(sums[0], rel=1e-12, abs=1e-9)
            assert float((x.double() ** 2).sum()) == pytest.approx(sums[1], rel=1e-12)
            if x.dim() == 4:
                assert all(torch.equal(x[0], x[t]) for t in range(1, x.shape[0]))       # one frame, repeated
            k += 1
    assert k == int(g17["%s_count" % tag])


def test_static_part_ranges_and_bad_mode():
    ds = D.staticUCF50(UCF, "train", frames=1, split_num=4, split_id=2, split_mode='feature')
    assert ds.split_lists[0] == [12, 25, 40]                             # the index file lists them unsorted
    assert ds._frame_range(0, 64) == (26, 41)
    assert D.staticUCF50(UCF, "train", split_num=4, split_id=0, split_mode='feature')._frame_range(0, 64) == (1, 13)
    assert D.staticUCF50(UCF, "train", split_num=4, split_id=3, split_mode='feature')._frame_range(0, 64) == (41, 64)
    assert D.staticUCF101(UCF, "train", split_num=3, split_id=1)._frame_range(1, 40) == (14, 26)
    assert D.staticHMDB51(UCF, "train")._frame_range(2, 36) == (1, 36)
    with pytest.raises(SystemExit):
        D.staticUCF50(UCF, "train", split_mode='other')
    with pytest.raises(ValueError):
        D.StillFrameVideos('UCF101', UCF, "train")


@pytest.mark.parametrize("tag,make,path,frames_tag,fixture", [
    ("skin", D.singleKinetics400, KIN, "kin", "g15"), ("sssv2", D.singleSSv2, SSV2, "ssv2", "g17")])
def test_single_frame_sets_pick_a_listed_file(g15, g17, tag, make, path, frames_tag, fixture):
    """One ``random.randint`` per item and nothing else; the image is the listed file at that position (directory order is
    the box's, so pixels are looked up by file name in the per-frame records of the video-class fixtures)."""
    frames = g15 if fixture == "g15" else g17
    for split in ("train", "val"):
        ds = make(path, split)
        assert [os.path.basename(d) for d in ds.video_dirs] == g17["%s_%s_dirs" % (tag, split)].tolist()
        np.testing.assert_array_equal(g17["%s_%s_labels" % (tag, split)], np.array(ds.labels))
        picks = g17["%s_%s_picks" % (tag, split)].tolist()
        random.seed(7)
        np_state, torch_state = np.random.get_state()[1].copy(), torch.get_rng_state()
        k = 0
        for _ in range(3):
            for i in range(len(ds)):
                x, y = ds[i]
                assert x.shape == (3, 64, 64) and y == ds.labels[i]
                name = os.listdir(ds.video_dirs[i])[picks[k]]
                r = frames["%s_%s_%d_names" % (frames_tag, split, i)].tolist().index(name)
                np.testing.assert_array_equal(frames["%s_%s_%d_probe" % (frames_tag, split, i)][r], x[:, ::8, ::8].numpy())
                assert float(x.double().sum()) == pytest.approx(float(frames["%s_%s_%d_sums" % (frames_tag, split, i)][r]), rel=1e-12, abs=1e-9)
                k += 1
        assert k == len(picks)
        assert (np.random.get_state()[1] == np_state).all() and torch.equal(torch.get_rng_state(), torch_state)


def test_ssv2_video_class_listing_and_items(g17):
    for split in ("train", "val"):
        ds = D.SSv2(SSV2, split)
        assert [os.path.basename(d) for d in ds.video_dirs] == g17["ssv2_%s_dirs" % split].tolist()
        np.testing.assert_array_equal(g17["ssv2_%s_labels" % split], np.array(ds.labels))
        assert ds.skipped == (1 if split == "train" else 0)              # the 6-frame video is not a sample
        for i in range(len(ds)):
            x, _ = ds[i]
            assert x.shape == (8, 3, 64, 64)
            ref_names = g17["ssv2_%s_%d_names" % (split, i)].tolist()
            for t, name in enumerate(os.listdir(ds.video_dirs[i])):
                r = ref_names.index(name)
                np.testing.assert_array_equal(g17["ssv2_%s_%d_probe" % (split, i)][r], x[t, :, ::8, ::8].numpy())


def test_get_dataset_still_branches():
    root = os.path.join(GOLD, "frames")
    channel, im_size, num_classes, _, _, _, dst_train, dst_test, loader = D.get_dataset("staticUCF50", root, split_num=4, split_id=2)
    assert (channel, im_size, num_classes) == (3, (112, 112), 50) and dst_train.frames == 16
    assert (dst_train.split_num, dst_train.split_id) == (1, 0)           # static* branches do not forward the split arguments
    assert next(iter(loader))[0].shape == (2, 16, 3, 112, 112)
    _, _, nc, _, _, _, tr, te, loader = D.get_dataset("singleUCF50", root, split_num=4, split_id=2, split_mode='feature')
    assert nc == 50 and (tr.frames, tr.split_num, tr.split_id, tr.split_mode) == (1, 4, 2, 'feature')
    assert next(iter(loader))[0].shape == (2, 3, 112, 112)
    _, _, nc, _, _, _, tr, te, _ = D.get_dataset("singleUCF101", root, split_num=2, split_id=1)
    assert nc == 101 and tr.family == 'staticUCF101' and (tr.split_num, tr.split_id) == (2, 1)
    _, im_size, nc, _, _, _, tr, _, _ = D.get_dataset("singleUCF101", root, img_size=(64, 64))
    assert im_size == (64, 64) and tr.transform.resize == (100, 80) and tr[0][0].shape == (3, 64, 64)
    for name in ("staticHMDB51", "singleHMDB51", "singleKinetics400", "singleSSv2"):
        with pytest.raises(AssertionError):
            D.get_dataset(name, root)                                    # directories this tree does not have


def test_single_hmdb_resizes_only_for_64(tmp_path):
    os.symlink(UCF, tmp_path / "HMDB51")
    _, _, nc, _, _, _, tr, _, _ = D.get_dataset("singleHMDB51", str(tmp_path), img_size=(64, 64))
    assert nc == 51 and tr.frames == 1 and tr.transform.resize == (100, 80) and tr[0][0].shape == (3, 64, 64)
    _, _, _, _, _, _, tr, _, _ = D.get_dataset("singleHMDB51", str(tmp_path), img_size=(96, 96))
    assert tr.transform.resize is None and tr[0][0].shape == (3, 112, 112)          # utils.py:345: no Resize / RandomCrop unless 64x64
    _, _, _, _, _, _, tr, _, _ = D.get_dataset("staticHMDB51", str(tmp_path), img_size=(96, 80))
    assert tr.transform.resize == (100, 80) and tr[0][0].shape == (16, 3, 96, 80)
