"""Image ingest for the benchmark harness.

* ``ImageFolderDataset``: PNG/JPEG files -> float32 CHW in [0, 1] (``ToTensor``: uint8 / 255), the Kodak path of
  ``cbench/data/datasets/torchvision_datasets.py:80-88``.
* ``RandomImageDataset``: the synthetic generator of ``configs/datasets/images/random_image_generator.py:12-15``
  (image i = ``torch.manual_seed(i); torch.rand(3, H, W)``).
"""
import os

import numpy as np
import torch


class ImageFolderDataset:
    EXT = (".png", ".jpg", ".jpeg", ".bmp", ".ppm")

    def __init__(self, root, max_num=None):
        self.files = sorted(os.path.join(root, f) for f in os.listdir(root) if f.lower().endswith(self.EXT))
        if max_num is not None:
            self.files = self.files[:max_num]

    def __len__(self):
        return len(self.files)

    def __getitem__(self, i):
        from PIL import Image
        with Image.open(self.files[i]) as im:
            a = np.array(im.convert("RGB"), dtype=np.uint8)  # own, writable copy
        return torch.from_numpy(a).permute(2, 0, 1).contiguous().float().div_(255.0)


class RandomImageDataset:
    def __init__(self, num=8, size=(3, 256, 256)):
        self.num, self.size = num, tuple(size)

    def __len__(self):
        return self.num

    def __getitem__(self, i):
        if not 0 <= i < self.num:
            raise IndexError(i)
        torch.manual_seed(i)
        return torch.rand(*self.size)


def batched(dataset, batch_size=1):
    """Minimal DataLoader: stacks ``batch_size`` equally sized images (the reference tests with batch_size=1)."""
    buf = []
    for i in range(len(dataset)):
        buf.append(dataset[i])
        if len(buf) == batch_size:
            yield torch.stack(buf)
            buf = []
    if buf:
        yield torch.stack(buf)
