"""Frame-folder video datasets and their HBM-resident preload.

The reference reads its real clips from folders of JPEG frames through six near-identical ``Dataset`` classes
(distill_utils/dataset.py: ``UCF101`` :146-249, ``HMDB51`` :251-351, ``miniUCF101`` :353-467, ``miniHMDB51`` :469-568,
``Kinetics400`` :79-144, ``SSv2`` :841-895) and, under ``--preload``, stacks every item into one host tensor that ``get_images`` then slices and
copies to the GPU per class and step (distill_baseline.py:36-45, 84-90; 7.7 GB of PCIe traffic per step at config 2).

Here one class, ``FrameFolderVideos``, covers the six (a ``spec`` says where the index file is and how frames are
picked), with the reference's item semantics — including the ORDER in which the global ``numpy.random`` / ``random`` /
``torch`` generators are consumed, so a seeded run picks the same frames, flips and crops — and ``preload`` puts the
whole split into HBM once: JPEG decode on host threads (PIL releases the GIL), uint8 frames through pinned staging,
``vd_frames_normalize`` (HWC uint8 -> CHW fp32, ``(v/255 - mean)/std``, the arithmetic of torchvision's
``ToTensor`` + ``Normalize``) on the device.  A 112x112x16 split of miniUCF101 (4662 clips) is 11.2 GB of fp32 clips:
resident, and the per-step gather is an index_select on the device (``distill.RealPool``).

``StillFrameVideos`` covers the reference's still-image variants (``staticHMDB51`` :570-650, ``staticUCF101`` :652-736,
``staticUCF50`` :738-839, ``singleKinetics400`` :18-77, ``singleSSv2`` :897-947): one frame of a video, repeated into a
"boring" clip or handed out as an image.  They feed the reference's static-memory learning stage, which is off the hot
path, so they are host-side only (no preload kernel of their own: an item is a clip or an image like any other).
"""
from __future__ import annotations

import csv
import json
import os
import os.path as osp
import random
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.utils.data as tdata

NUM_FRAMES = 16          # distill_utils/dataset.py:15
FRAME_GAP = 4            # distill_utils/dataset.py:16
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


@dataclass(frozen=True)
class FolderSpec:
    """How one dataset family is laid out on disk and how a clip's frames are chosen."""
    index: str                       # 'csv-splits' (UCF/HMDB), 'csv-kinetics', 'json-ssv2'
    index_file: str                  # file name under the dataset root ('{split}' is substituted)
    frames_root: str                 # sub-directory holding one folder per video
    pick: str                        # 'window' (random start + stride), 'segments' (one frame per 1/16), 'all'


SPECS = {
    'UCF101': FolderSpec('csv-splits', 'ucf101_splits1.csv', 'jpegs_112', 'window'),
    'HMDB51': FolderSpec('csv-splits', 'hmdb51_splits.csv', 'jpegs_112', 'window'),
    'miniUCF101': FolderSpec('csv-splits', 'ucf50_splits1.csv', 'jpegs_112', 'window'),
    'miniHMDB51': FolderSpec('csv-splits', 'hmdb25_splits.csv', 'jpegs_112', 'window'),          # distill_utils/dataset.py:469-568
    'Kinetics400': FolderSpec('csv-kinetics', '{split}.csv', '{split}', 'all'),
    'SSv2': FolderSpec('json-ssv2', 'annot_{split}.json', 'frame', 'all'),
}


class FrameTransform:
    """``ToTensor`` + ``Normalize`` and, when the target size is not the stored 112x112, ``Resize((100, 80))`` +
    ``RandomCrop(im_size)`` in front (utils.py:164-174 and its twins).  torchvision is not needed: a PIL image goes
    through PIL's own bilinear resize (what torchvision calls for PIL inputs), a crop whose offsets are drawn as
    ``RandomCrop.get_params`` draws them (``torch.randint`` for the row, then the column; none when nothing is cut), and
    ``uint8/255`` followed by ``(x - mean)/std`` in fp32."""

    def __init__(self, im_size: Tuple[int, int], mean=IMAGENET_MEAN, std=IMAGENET_STD, stored=(112, 112)):
        self.im_size = tuple(im_size)
        self.resize = None if self.im_size == tuple(stored) else (100, 80)
        self.mean = torch.tensor(mean, dtype=torch.float32)
        self.std = torch.tensor(std, dtype=torch.float32)

    def draw(self, height: int, width: int):
        """The random part of one call: the crop origin (None when the transform has no crop)."""
        if self.resize is None:
            return None
        h, w = self.resize
        th, tw = self.im_size
        if h < th or w < tw:
            raise ValueError("Required crop size %s is larger than input image size %s" % ((th, tw), (h, w)))
        if (h, w) == (th, tw):
            return (0, 0)
        i = int(torch.randint(0, h - th + 1, size=(1,)).item())
        j = int(torch.randint(0, w - tw + 1, size=(1,)).item())
        return (i, j)

    def pixels(self, image, crop) -> np.ndarray:
        """PIL image -> (H, W, 3) uint8 after resize / crop."""
        from PIL import Image
        if image.mode != 'RGB':
            image = image.convert('RGB')
        if self.resize is not None:
            image = image.resize((self.resize[1], self.resize[0]), Image.BILINEAR)
            i, j = crop
            image = image.crop((j, i, j + self.im_size[1], i + self.im_size[0]))
        return np.asarray(image, dtype=np.uint8)

    def normalise(self, u8: torch.Tensor) -> torch.Tensor:
        """(..., H, W, 3) uint8 -> (..., 3, H, W) fp32; the host-side twin of ``vd_frames_normalize``."""
        x = u8.movedim(-1, -3).to(torch.float32).div(255)
        return x.sub_(self.mean.view(3, 1, 1)).div_(self.std.view(3, 1, 1))

    def __call__(self, image) -> torch.Tensor:
        crop = self.draw(image.height, image.width)
        return self.normalise(torch.from_numpy(self.pixels(image, crop).copy()))


@dataclass
class ClipDraw:
    """Everything random about one item, drawn up front so that the file reads can run on worker threads."""
    files: List[str]
    flip: bool
    crops: List[Optional[Tuple[int, int]]]


class FrameFolderVideos(tdata.Dataset):
    """``dataset[i] -> (clip (T, 3, H, W) fp32, label)`` over a folder of per-video frame directories.

    ``family`` picks the layout (``SPECS``); ``split`` is 'train' / 'test' ('val' for Kinetics400 / SSv2).  Labels
    are positions in the sorted set of label strings of THIS split, ``labels`` / ``targets`` list them per item, and the
    start frame of a training item is drawn once and kept (``self.start``), as in the reference."""

    def __init__(self, family: str, path: str, split: str, transform: Optional[FrameTransform] = None, sample: str = 'random'):
        if family not in SPECS:
            raise ValueError("unknown dataset family: %s" % family)
        self.family, self.spec, self.split, self.sample = family, SPECS[family], split, sample
        self.transform = transform if transform is not None else FrameTransform((112, 112))
        self.root = path
        names, label_strs, self.skipped = self._read_index(path, split)
        self.video_dirs = names
        self.label_strs = label_strs
        self.class_strs = sorted(set(label_strs))
        self.class_2_idx = {s: i for i, s in enumerate(self.class_strs)}
        self.labels = [self.class_2_idx[s] for s in label_strs]
        self.targets = self.labels
        self.start = [-1] * len(self.video_dirs)

    # -- index files -----------------------------------------------------------------------------------------------------
    def _expected_frames(self, path: str) -> int:
        tail = path.split("/")[-1]
        return 8 if tail in ("kinetics_64x64x8", "SSv2_64x8") else NUM_FRAMES       # dataset.py:81-84, 843-846

    def _read_index(self, path: str, split: str):
        spec = self.spec
        dirs, labels, skipped = [], [], 0
        if spec.index == 'csv-splits':
            frames_root = osp.join(path, spec.frames_root)
            with open(osp.join(path, spec.index_file)) as fp:
                for row in csv.DictReader(fp):
                    if row["split"] != split:
                        continue
                    dirs.append(osp.join(frames_root, row["folder_name"]))
                    labels.append(row["label"])
            return dirs, labels, 0
        want = self._expected_frames(path)

        def usable(d):
            return osp.exists(d) and len(os.listdir(d)) == want
        if spec.index == 'csv-kinetics':
            csv_split = "validate" if split == "val" else split
            with open(osp.join(path, "%s.csv" % csv_split)) as fp:
                for row in csv.DictReader(fp):
                    if row["split"] != csv_split:
                        raise AssertionError("row of split %r in %s.csv" % (row["split"], csv_split))
                    name = "%s_%06d_%06d" % (row["youtube_id"], int(row["time_start"]), int(row["time_end"]))
                    d = osp.join(path, split, name)
                    if not usable(d):
                        d = osp.join(path, "replacement", name)
                    if not usable(d):
                        skipped += 1
                        continue
                    dirs.append(d)
                    labels.append(row["label"])
            return dirs, labels, skipped
        with open(osp.join(path, spec.index_file.format(split=split))) as fp:
            for item in json.load(fp):
                d = osp.join(path, spec.frames_root, item['id'])
                if not usable(d):
                    skipped += 1
                    continue
                dirs.append(d)
                labels.append(item["class"])
        return dirs, labels, skipped

    def __len__(self) -> int:
        return len(self.video_dirs)

    # -- one item: the draws, then the reads -----------------------------------------------------------------------------
    def draw(self, index: int) -> ClipDraw:
        """Consumes the global generators exactly as the reference's ``__getitem__`` does: ``np.random.randint`` for the
        start (first visit of a training item, every visit of a test item), segment picks for 'split-random', then
        ``random.random()`` for the flip, then the per-frame crop origins."""
        path = self.video_dirs[index]
        if self.spec.pick == 'all':
            files = [osp.join(path, f) for f in os.listdir(path)]        # directory order, as the reference
            flip = False
        else:
            length = len(os.listdir(path))
            skip = length // NUM_FRAMES if length < NUM_FRAMES * FRAME_GAP else FRAME_GAP
            if self.start[index] == -1 or self.split == "test":
                self.start[index] = int(np.random.randint(1, length - (NUM_FRAMES - 1) * skip))
            first = self.start[index]
            numbers = list(range(first, first + NUM_FRAMES * skip, skip))
            if self.family == 'miniUCF101' and self.sample == 'split-random':
                seg = length // 16
                bounds = [(k * seg, (k + 1) * seg if k < 15 else length) for k in range(16)]
                numbers = [int(np.random.randint(lo, hi)) + 1 for lo, hi in bounds]
            files = [osp.join(path, "frame{:06d}.jpg".format(n)) for n in numbers]
            flip = random.random() > 0.5
        crops = [self.transform.draw(0, 0) for _ in files]
        return ClipDraw(files, flip, crops)

    def read_u8(self, d: ClipDraw) -> np.ndarray:
        """-> (T, H, W, 3) uint8: decode, flip, resize / crop.  Thread-safe (no generator is touched)."""
        from PIL import Image
        frames = []
        for f, crop in zip(d.files, d.crops):
            with Image.open(f) as im:
                if d.flip:
                    im = im.transpose(Image.FLIP_LEFT_RIGHT)
                frames.append(self.transform.pixels(im, crop))
        return np.stack(frames, 0)

    def __getitem__(self, index: int):
        u8 = self.read_u8(self.draw(index))
        return self.transform.normalise(torch.from_numpy(u8)), self.labels[index]


# family -> (FolderSpec, how the one frame is chosen)
_STILL_SPECS = {
    'staticHMDB51': FolderSpec('csv-splits', 'hmdb51_splits.csv', 'jpegs_112', 'still'),
    'staticUCF101': FolderSpec('csv-splits', 'ucf101_splits1.csv', 'jpegs_112', 'still-part'),
    'staticUCF50': FolderSpec('csv-splits', 'ucf50_splits1_max.csv', 'jpegs_112', 'still-part'),
    'singleKinetics400': FolderSpec('csv-kinetics', '{split}.csv', '{split}', 'listed'),
    'singleSSv2': FolderSpec('json-ssv2', 'annot_{split}.json', 'frame', 'listed'),
}


class StillFrameVideos(FrameFolderVideos):
    """One frame per item.  The ``static*`` families draw a frame NUMBER with ``np.random.randint`` -- anywhere in the
    video (staticHMDB51, dataset.py:644), in the ``split_id``-th of ``split_num`` equal parts (staticUCF101 :730,
    staticUCF50 ``split_mode='mean'`` :823), or between the per-video cut points of the index file's ``split_index``
    column (staticUCF50 ``split_mode='feature'``, four parts, :824-830) -- then the flip, then ONE transform call, and
    return the frame stacked ``frames`` times, or as a (3, H, W) image when ``frames == 1`` (:626-633).  The ``single*``
    families (Kinetics400 / SSv2 frame sets) pick a listed file with ``random.randint`` and do not flip (:69-77, :938-947).
    A ``split_id`` outside ``split_num`` reads as 0 (:660, :746); the last frame of a video is never drawn, as in the
    reference (``randint``'s upper bound is the frame count)."""

    def __init__(self, family: str, path: str, split: str, transform: Optional[FrameTransform] = None, frames: int = NUM_FRAMES,
                 split_num: int = 1, split_id: int = 0, split_mode: str = 'mean'):
        if family not in _STILL_SPECS:
            raise ValueError("unknown still-frame family: %s" % family)
        self.family, self.spec, self.split, self.sample = family, _STILL_SPECS[family], split, 'random'
        self.transform = transform if transform is not None else FrameTransform((112, 112))
        self.root = path
        self.frames, self.split_num, self.split_mode = int(frames), int(split_num), split_mode
        self.split_id = 0 if split_id >= split_num else int(split_id)
        if family == 'staticUCF50' and split_mode not in ('mean', 'feature'):
            raise SystemExit("split_mode error!")                       # the reference exits on the first item (:831-833)
        self.split_lists: List[List[int]] = []
        names, label_strs, self.skipped = self._read_index(path, split)
        if family == 'staticUCF50':
            with open(osp.join(path, self.spec.index_file)) as fp:
                for row in csv.DictReader(fp):
                    if row["split"] == split:
                        self.split_lists.append(sorted(int(v) for v in row['split_index'].strip('][').split(', ')))
        self.video_dirs = names
        self.label_strs = label_strs
        self.class_strs = sorted(set(label_strs))
        self.class_2_idx = {s: i for i, s in enumerate(self.class_strs)}
        self.labels = [self.class_2_idx[s] for s in label_strs]
        self.targets = self.labels
        self.start = [-1] * len(self.video_dirs)                        # kept for shape; a still item redraws every visit

    def _expected_frames(self, path: str) -> int:
        return 8                                                        # dataset.py:20, 899 (both frame sets are 8 long)

    @property
    def image_items(self) -> bool:
        """Items are (3, H, W) images rather than (frames, 3, H, W) clips."""
        return self.spec.pick == 'listed' or self.frames == 1

    def _frame_range(self, index: int, length: int) -> Tuple[int, int]:
        if self.spec.pick == 'still':
            return 1, length
        if self.family == 'staticUCF50' and self.split_mode == 'feature':
            cuts = self.split_lists[index]
            if self.split_id == 0:
                return 1, cuts[0] + 1
            if self.split_id == 3:
                return cuts[2] + 1, length
            return cuts[self.split_id - 1] + 1, cuts[self.split_id] + 1
        part = length // self.split_num
        return part * self.split_id + 1, part * (self.split_id + 1)

    def draw(self, index: int) -> ClipDraw:
        path = self.video_dirs[index]
        listing = os.listdir(path)
        if self.spec.pick == 'listed':
            files = [osp.join(path, listing[random.randint(0, len(listing) - 1)])]
            flip = False
        else:
            lo, hi = self._frame_range(index, len(listing))
            number = int(np.random.randint(lo, hi))
            self.start[index] = number
            files = [osp.join(path, "frame{:06d}.jpg".format(number))]
            flip = random.random() > 0.5
        return ClipDraw(files, flip, [self.transform.draw(0, 0)])

    def __getitem__(self, index: int):
        image = self.transform.normalise(torch.from_numpy(self.read_u8(self.draw(index))))[0]
        if self.image_items:
            return image, self.labels[index]
        return torch.stack([image] * self.frames, 0), self.labels[index]


def staticHMDB51(path, split, transform=None, frames=NUM_FRAMES):
    return StillFrameVideos('staticHMDB51', path, split, transform, frames)


def staticUCF101(path, split, transform=None, frames=NUM_FRAMES, split_num=1, split_id=0):
    return StillFrameVideos('staticUCF101', path, split, transform, frames, split_num, split_id)


def staticUCF50(path, split, transform=None, frames=NUM_FRAMES, split_num=1, split_id=0, split_mode='mean'):
    return StillFrameVideos('staticUCF50', path, split, transform, frames, split_num, split_id, split_mode)


def singleKinetics400(path, split, transform=None):
    return StillFrameVideos('singleKinetics400', path, split, transform)


def singleSSv2(path, split, transform=None):
    return StillFrameVideos('singleSSv2', path, split, transform)


def UCF101(path, split, transform=None):
    return FrameFolderVideos('UCF101', path, split, transform)


def HMDB51(path, split, transform=None):
    return FrameFolderVideos('HMDB51', path, split, transform)


def miniUCF101(path, split, transform=None, sample='random'):
    return FrameFolderVideos('miniUCF101', path, split, transform, sample)


def miniHMDB51(path, split, transform=None):
    return FrameFolderVideos('miniHMDB51', path, split, transform)


def Kinetics400(path, split, transform=None):
    return FrameFolderVideos('Kinetics400', path, split, transform)


def SSv2(path, split, transform=None):
    return FrameFolderVideos('SSv2', path, split, transform)


# name -> (family, sub-directory of data_path, num_classes, fixed im_size or None)   utils.py:132-236
_DATASETS = {
    'Kinetics400': ('Kinetics400', 'Kinetics', 400, (64, 64)),
    'Kinetics400_long': ('Kinetics400', 'kinetics_112x112x16', 400, (112, 112)),
    'UCF101': ('UCF101', 'UCF101', 101, None),
    'HMDB51': ('HMDB51', 'HMDB51', 51, None),
    'miniUCF101': ('miniUCF101', 'UCF101', 50, None),
}                                                   # (the reference has a video ``SSv2`` class but no get_dataset branch for it)


# still-image names -> (family, sub-directory, num_classes, fixed im_size or None, frames, passes the split arguments)   utils.py:237-454
_STILL_DATASETS = {
    'staticHMDB51': ('staticHMDB51', 'HMDB51', 51, None, NUM_FRAMES, False),
    'staticUCF101': ('staticUCF101', 'UCF101', 101, None, NUM_FRAMES, False),
    'staticUCF50': ('staticUCF50', 'UCF101', 50, None, NUM_FRAMES, False),
    'singleHMDB51': ('staticHMDB51', 'HMDB51', 51, None, 1, False),
    'singleUCF50': ('staticUCF50', 'UCF101', 50, None, 1, True),
    'singleUCF101': ('staticUCF101', 'UCF101', 101, None, 1, True),
    'singleKinetics400': ('singleKinetics400', 'Kinetics', 400, (64, 64), 1, False),
    'singleSSv2': ('singleSSv2', 'SSv2', 174, (64, 64), 1, False),
}


def get_dataset(dataset: str, data_path: str, num_workers: int = 0, img_size=(112, 112), split_num: int = 1, split_id: int = 0,
                split_mode: str = 'mean'):
    """The video and still-frame branches of the reference's ``get_dataset`` (utils.py:21, 132-454, 507-508): same 9-tuple
    ``(channel, im_size, num_classes, class_names, mean, std, dst_train, dst_test, testloader)``.  Unknown names end the
    process as the reference does (utils.py:505).  The split arguments reach only ``singleUCF50`` / ``singleUCF101``
    (utils.py:380-381, 411-412); ``singleHMDB51`` resizes + crops only for 64x64 targets (utils.py:345), the others for any
    size that is not the stored 112x112."""
    if dataset in _STILL_DATASETS:
        family, sub, num_classes, fixed, frames, splits = _STILL_DATASETS[dataset]
        im_size = tuple(fixed) if fixed is not None else tuple(img_size)
        path = data_path + "/" + sub
        assert os.path.exists(path), path
        listed = _STILL_SPECS[family].pick == 'listed'
        plain = listed or (dataset == 'singleHMDB51' and im_size != (64, 64))
        transform = FrameTransform(im_size, stored=im_size if plain else (112, 112))
        kw = dict(split_num=split_num, split_id=split_id) if splits else {}
        if splits and family == 'staticUCF50':
            kw['split_mode'] = split_mode
        dst_train = StillFrameVideos(family, path, "train", transform, frames, **kw)
        dst_test = StillFrameVideos(family, path, "val" if listed else "test", transform, frames, **kw)
        testloader = tdata.DataLoader(dst_test, batch_size=64, shuffle=False, num_workers=num_workers)
        return 3, im_size, num_classes, None, list(IMAGENET_MEAN), list(IMAGENET_STD), dst_train, dst_test, testloader
    if dataset not in _DATASETS:
        raise SystemExit('unknown dataset: %s' % dataset)
    family, sub, num_classes, fixed = _DATASETS[dataset]
    im_size = tuple(fixed) if fixed is not None else tuple(img_size)
    path = data_path + "/" + sub
    assert os.path.exists(path), path
    listed = SPECS[family].pick == 'all'            # pre-resized frame sets: no Resize / RandomCrop branch
    transform = FrameTransform(im_size, stored=im_size if listed else (112, 112))
    test_split = "val" if listed else "test"
    dst_train = FrameFolderVideos(family, path, "train", transform)
    dst_test = FrameFolderVideos(family, path, test_split, transform)
    testloader = tdata.DataLoader(dst_test, batch_size=64, shuffle=False, num_workers=num_workers)
    return 3, im_size, num_classes, None, list(IMAGENET_MEAN), list(IMAGENET_STD), dst_train, dst_test, testloader


# ------------------------------------------------------------------------------------------------------------------------
# HBM-resident preload
# ------------------------------------------------------------------------------------------------------------------------
def frames_normalize(u8: torch.Tensor, out: torch.Tensor, mean: Sequence[float], std: Sequence[float]) -> torch.Tensor:
    """(n, H, W, 3) uint8 on the device -> ``out`` (n, 3, H, W) fp32 by ``vd_frames_normalize`` on the current stream."""
    import ctypes
    from . import hip
    n, h, w, c = u8.shape
    assert c == 3 and u8.dtype == torch.uint8 and u8.is_contiguous() and u8.is_cuda
    assert out.dtype == torch.float32 and out.is_contiguous() and out.numel() == u8.numel() and out.device == u8.device
    m = (ctypes.c_float * 3)(*[float(v) for v in mean])
    s = (ctypes.c_float * 3)(*[float(v) for v in std])
    with torch.cuda.device(u8.device):
        hip.check(hip.lib().vd_frames_normalize(hip.ptr(u8), hip.ptr(out), ctypes.c_int64(n), int(h), int(w), m, s, hip.stream_ptr(u8.device)),
                  "vd_frames_normalize")
    return out


def preload(dataset: FrameFolderVideos, device, indices: Optional[Sequence[int]] = None, workers: int = 8, chunk: int = 64):
    """Every item of ``dataset`` (or of ``indices``, in that order) decoded once and left in HBM.

    -> ``(clips (N, T, 3, H, W) fp32 on ``device`` -- (N, 3, H, W) for the image items of ``StillFrameVideos`` --, labels
    (N,) int64 on the host)``; the same values
    ``torch.stack([dataset[i][0] for i in indices])`` produces (the draws happen here, in index order; only the file reads
    run on the ``workers`` threads), with 1/4 of the bytes crossing PCIe and the normalisation done by the device.
    The copy of chunk k+1 overlaps the decode of chunk k+2 and the normalisation of chunk k (pinned double buffer, one
    copy stream)."""
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("dataset.preload: the preload targets HBM; use dataset[i] for host tensors")
    idx = list(range(len(dataset))) if indices is None else [int(i) for i in indices]
    labels = torch.tensor([dataset.labels[i] for i in idx], dtype=torch.int64)
    if not idx:
        return torch.empty((0, 0, 3, 0, 0), device=device), labels
    draws = [dataset.draw(i) for i in idx]                     # generator order = item order
    tf = dataset.transform
    first = dataset.read_u8(draws[0])
    t, h, w, _ = first.shape
    clips = torch.empty((len(idx), t, 3, h, w), dtype=torch.float32, device=device)
    copy_stream = torch.cuda.Stream(device=device)
    main = torch.cuda.current_stream(device)
    stage = [torch.empty((chunk, t, h, w, 3), dtype=torch.uint8).pin_memory() for _ in range(2)]
    dev_u8 = [torch.empty((chunk, t, h, w, 3), dtype=torch.uint8, device=device) for _ in range(2)]
    landed = [torch.cuda.Event() for _ in range(2)]          # H2D of the buffer finished
    drained = [torch.cuda.Event() for _ in range(2)]         # normalise kernel has read the device buffer
    used = [False, False]

    def decode_into(buf, k, item):
        got = first if item == 0 else dataset.read_u8(draws[item])
        if got.shape != (t, h, w, 3):
            raise ValueError("clip %s has shape %s, the first one %s" % (draws[item].files[0], got.shape, (t, h, w, 3)))
        buf[k] = torch.from_numpy(got)

    with ThreadPoolExecutor(max_workers=max(1, workers)) as pool:
        for ci, lo in enumerate(range(0, len(idx), chunk)):
            b = ci & 1
            n = min(chunk, len(idx) - lo)
            if used[b]:
                landed[b].synchronize()                        # the pinned buffer may be overwritten
            list(pool.map(lambda k: decode_into(stage[b], k, lo + k), range(n)))
            with torch.cuda.stream(copy_stream):
                if used[b]:
                    copy_stream.wait_event(drained[b])
                dev_u8[b][:n].copy_(stage[b][:n], non_blocking=True)
                landed[b].record(copy_stream)
            main.wait_event(landed[b])
            frames_normalize(dev_u8[b][:n].view(n * t, h, w, 3), clips[lo:lo + n].view(n * t, 3, h, w), tf.mean.tolist(), tf.std.tolist())
            drained[b].record(main)
            used[b] = True
    main.synchronize()
    if isinstance(dataset, StillFrameVideos):                   # one decoded frame per item: an image, or that frame repeated on the device
        clips = clips[:, 0] if dataset.image_items else clips.expand(-1, dataset.frames, -1, -1, -1).contiguous()
    return clips, labels


def indices_class(labels: Sequence[int], num_classes: int) -> List[List[int]]:
    """``indices_class[c]`` = item numbers of class c in dataset order (distill_baseline.py:76-81)."""
    out: List[List[int]] = [[] for _ in range(num_classes)]
    for i, lab in enumerate(labels):
        out[int(lab)].append(i)
    return out


class DeviceBatches:
    """Mini-batches of an HBM-resident tensor set: what ``DataLoader(TensorDataset(clips, labels), batch_size, shuffle)``
    yields (buffer.py:62, utils.py:861), without a per-item host loop — one ``index_select`` per batch on the device.
    The shuffle consumes the global torch generator exactly as the DataLoader does (one int64 for the iterator's base
    seed, one for the sampler's seed, then ``randperm`` from a generator seeded with the latter), so a seeded run sees the
    same batches either way; the last batch is short."""

    def __init__(self, clips: torch.Tensor, labels: torch.Tensor, batch_size: int, shuffle: bool = False, generator=None):
        assert clips.shape[0] == labels.shape[0]
        self.clips, self.labels = clips, labels.to(clips.device)
        self.batch_size, self.shuffle, self.generator = int(batch_size), bool(shuffle), generator

    def __len__(self) -> int:
        return (self.clips.shape[0] + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n = self.clips.shape[0]
        torch.empty((), dtype=torch.int64).random_(generator=self.generator)          # the iterator's base seed
        if self.shuffle:
            if self.generator is None:
                gen = torch.Generator()
                gen.manual_seed(int(torch.empty((), dtype=torch.int64).random_().item()))
            else:
                gen = self.generator
            order = torch.randperm(n, generator=gen)
        else:
            order = torch.arange(n)
        order = order.to(self.clips.device)
        for lo in range(0, n, self.batch_size):
            idx = order[lo:lo + self.batch_size]
            yield self.clips.index_select(0, idx), self.labels.index_select(0, idx)
        if self.shuffle and self.generator is not None:
            torch.randperm(n, generator=self.generator)      # RandomSampler ends an epoch with a second (unused) permutation
