"""Tiny on-disk video dataset in the reference's format + what the REFERENCE's `VideoDataset` / `VideoDatasetTwoRes`
(dataset.py:26-162) return for it under fixed torch seeds. Run in the build container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_dataset.py [/root/reference]

Writes tests/golden/tiny_dataset/<HHHH>x<WWWW>/part{0,1}.zip (PNG frames: lossless, so the decoder does not matter) and
tests/golden/dataset.npz."""

import io
import json
import os
import sys
import zipfile

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

import torch  # noqa: E402
from PIL import Image  # noqa: E402
import dataset as ref_dataset  # noqa: E402

assert os.path.realpath(ref_dataset.__file__).startswith(os.path.realpath(REF))
ROOT = os.path.join(HERE, 'tiny_dataset')
CLIPS = {'part0': {'clipA': 9, 'sub/clipB': 5}, 'part1': {'clipC': 12, 'clipD': 2}}     # frames per clip


def frame(part, clip, i, h, w):
    """deterministic RGB content that identifies (clip, frame, row, column, channel)"""
    s = sum(map(ord, part + clip))
    y, x, c = np.meshgrid(np.arange(h), np.arange(w), np.arange(3), indexing='ij')
    return ((s * 7 + i * 31 + y * 13 + x * 5 + c * 83 + (x * y) % 11) % 256).astype(np.uint8)


def build(h, w):
    d = os.path.join(ROOT, f'{h:04d}x{w:04d}')
    os.makedirs(d, exist_ok=True)
    for part, clips in CLIPS.items():
        index = {}
        with zipfile.ZipFile(os.path.join(d, part + '.zip'), 'w', zipfile.ZIP_STORED) as zf:
            for clip, n in clips.items():
                names = [f'{i:05d}.png' for i in range(n)]
                index[clip] = names
                for i, name in enumerate(names):
                    buf = io.BytesIO()
                    Image.fromarray(frame(part, clip, i, h, w)).save(buf, format='PNG')
                    zf.writestr(zipfile.ZipInfo(f'{clip}/{name}', date_time=(2020, 1, 1, 0, 0, 0)), buf.getvalue())
            zf.writestr(zipfile.ZipInfo('frame_paths.json', date_time=(2020, 1, 1, 0, 0, 0)), json.dumps(index))


if __name__ == '__main__':
    build(8, 12)
    build(16, 24)
    out = {}
    ds = ref_dataset.VideoDataset(ROOT, seq_length=4, height=8, width=12, min_spacing=1, max_spacing=3, x_flip=True)
    out['n_clips'] = np.array(len(ds))
    torch.manual_seed(123)
    for k in range(6):
        item = ds[k % len(ds)]
        out[f'one_{k}_video'] = item['video'].numpy()
        out[f'one_{k}_spacing'] = np.array(item['spacing'])
    two = ref_dataset.VideoDatasetTwoRes(ROOT, seq_length=3, lr_height=8, lr_width=12, hr_height=16, hr_width=24, max_spacing=2, x_flip=True)
    torch.manual_seed(7)
    for k in range(4):
        item = two[k % len(two)]
        out[f'two_{k}_lr'] = item['lr_video'].numpy()
        out[f'two_{k}_hr'] = item['hr_video'].numpy()
        out[f'two_{k}_spacing'] = np.array(item['spacing'])
    np.savez_compressed(os.path.join(HERE, 'dataset.npz'), **out)
    print('clips', len(ds), 'items written', len(out))
