"""Data path (reference dataset.py:26-104 `VideoDataset`, :107-162 `VideoDatasetTwoRes`): frames stored as images inside
ZIP partitions -> clips.

Same on-disk format, same sampling (the torch random-number calls are made in the same order, so a seeded run picks the
same frames as the reference), but the arithmetic moved: `__getitem__` returns the decoded frames as uint8
[T, H, W, C] -- what the JPEG / PNG decoder produces -- and the conversion to the network's float layout
(2 * x / 255 - 1, channels first, optional horizontal flip) is ONE device pass over the whole batch
(`lvg.video_io.video_from_uint8`, csrc/video_io.hip) after the bytes crossed PCIe: 4x fewer host->device bytes than the
float32 clips the reference collates, and no per-frame float work in the loader workers. `to_video` is that step;
`reference_item` reproduces the reference's `__getitem__` output on the CPU (parity tests)."""

import json
from dataclasses import dataclass
from pathlib import Path, PurePosixPath
from typing import Any, Optional
from zipfile import ZipFile

import numpy as np
import torch
from PIL import Image
from torch.utils.data import Dataset

from . import video_io


def _read_partition_indices(root: Path) -> dict:
    """{partition name: {clip path: [frame file names]}} from the `frame_paths.json` inside every ZIP partition."""
    index = {}
    for archive in root.glob('*.zip'):
        with ZipFile(archive) as zf:
            index[archive.stem] = json.loads(zf.read('frame_paths.json'))
    return index


def _eligible_clips(index: dict, min_frames: int) -> list:
    """(partition, clip path, frame names) of every clip with at least `min_frames` frames, in sorted order."""
    clips = []
    for part in sorted(index):
        for clip in sorted(index[part]):
            names = index[part][clip]
            if len(names) >= min_frames:
                clips.append((part, clip, names))
    return clips


@dataclass
class VideoDataset(Dataset):
    dataset_dir: str
    seq_length: int
    height: int
    width: int
    min_spacing: int = 1
    max_spacing: int = 1
    min_video_length: Optional[int] = None
    x_flip: bool = False

    def __post_init__(self):
        if self.seq_length < 1:
            raise AssertionError('seq_length must be >= 1')
        self.dataset_path = Path(self.dataset_dir) / f'{self.height:04d}x{self.width:04d}'
        if not self.dataset_path.is_dir():
            raise AssertionError(self.dataset_path)
        # a clip must hold seq_length frames at the smallest spacing (and min_video_length, if that is longer)
        shortest = (self.seq_length - 1) * self.min_spacing + 1
        self.min_video_length = max(self.min_video_length or 1, shortest)
        self.frame_paths = _read_partition_indices(self.dataset_path)
        self.video_paths = _eligible_clips(self.frame_paths, self.min_video_length)
        self._zipfiles = {}

    def sample_frame_names(self, frame_names):
        """(frame names of one clip, spacing). Two torch.randint draws -- spacing, then start -- in the reference's order
        and with its bounds (dataset.py:59-71), so a seeded loader picks the same frames."""
        count = len(frame_names)
        widest = 1
        if self.seq_length > 1:
            widest = min(self.max_spacing, (count - 1) // (self.seq_length - 1))
        spacing = int(torch.randint(self.min_spacing, widest + 1, size=()))
        covered = 1 + spacing * (self.seq_length - 1)
        first = int(torch.randint(count - covered + 1, size=()))
        return frame_names[first:first + covered:spacing], spacing

    def read_frame_bytes(self, partition_name: str, frame_path: str) -> np.ndarray:
        """Decoded frame [H, W, C] uint8."""
        if partition_name not in self._zipfiles:
            self._zipfiles[partition_name] = ZipFile(self.dataset_path.joinpath(f'{partition_name}.zip'))
        with self._zipfiles[partition_name].open(frame_path, 'r') as fp:
            frame = np.array(Image.open(fp))
        return frame[:, :, None] if frame.ndim == 2 else frame

    def _clip(self, index: int):
        partition_name, clip_path, frame_names = self.video_paths[index]
        frame_names, spacing = self.sample_frame_names(frame_names)
        paths = [str(PurePosixPath(clip_path).joinpath(name)) for name in frame_names]
        return partition_name, paths, spacing

    def __getitem__(self, index: int) -> dict:
        partition_name, paths, spacing = self._clip(index)
        frames = torch.from_numpy(np.stack([self.read_frame_bytes(partition_name, p) for p in paths]))      # [T, H, W, C] uint8
        flip = bool(self.x_flip and torch.rand(()).item() < 0.5)
        return dict(frames=frames, flip=flip, spacing=spacing)

    def __len__(self) -> int:
        return len(self.video_paths)

    def __getstate__(self):
        return dict(self.__dict__, _zipfiles={})

    # -- the float side ------------------------------------------------------------------------------------------------
    @staticmethod
    def to_video(batch: dict, device=None, dtype: torch.dtype = torch.float32, key: str = 'frames') -> torch.Tensor:
        """Collated batch (frames [N, T, H, W, C] uint8, flip [N]) -> video [N, C, T, H, W] in [-1, 1] on `device`."""
        frames = batch[key]
        if device is not None:
            frames = frames.to(device, non_blocking=True)
        flip = batch.get('flip')
        flip = None if flip is None else torch.as_tensor(flip, dtype=torch.uint8)
        return video_io.video_from_uint8(frames, flip, dtype=dtype)

    def reference_item(self, index: int) -> dict:
        """What the reference's `VideoDataset.__getitem__` returns (float32 [C, T, H, W] video in [-1, 1]), on the CPU."""
        item = self[index]
        video = video_io.video_from_uint8(item['frames'][None], torch.tensor([item['flip']], dtype=torch.uint8))[0]
        return dict(video=video, spacing=item['spacing'])


@dataclass
class VideoDatasetTwoRes(Dataset):
    """The same clip at two resolutions (reference dataset.py:107-162); one sampling decision, one flip decision."""
    dataset_dir: str
    seq_length: int
    lr_height: int
    lr_width: int
    hr_height: int
    hr_width: int
    min_spacing: int = 1
    max_spacing: int = 1
    min_video_length: Optional[int] = None
    x_flip: bool = False

    def __post_init__(self):
        common = (self.min_spacing, self.max_spacing, self.min_video_length)
        self.lr_dataset = VideoDataset(self.dataset_dir, self.seq_length, self.lr_height, self.lr_width, *common, x_flip=self.x_flip)
        self.hr_dataset = VideoDataset(self.dataset_dir, self.seq_length, self.hr_height, self.hr_width, *common, x_flip=self.x_flip)
        assert self.lr_dataset.video_paths == self.hr_dataset.video_paths

    def __getitem__(self, index: int) -> dict:
        partition_name, paths, spacing = self.lr_dataset._clip(index)
        lr = torch.from_numpy(np.stack([self.lr_dataset.read_frame_bytes(partition_name, p) for p in paths]))
        hr = torch.from_numpy(np.stack([self.hr_dataset.read_frame_bytes(partition_name, p) for p in paths]))
        flip = bool(self.x_flip and torch.rand(()).item() < 0.5)
        return dict(lr_frames=lr, hr_frames=hr, flip=flip, spacing=spacing)

    def __len__(self) -> int:
        return len(self.lr_dataset)

    @staticmethod
    def to_videos(batch: dict, device=None, dtype: torch.dtype = torch.float32):
        return (VideoDataset.to_video(batch, device, dtype, key='lr_frames'), VideoDataset.to_video(batch, device, dtype, key='hr_frames'))
