"""Query/reference image sets (IO side of lens/src/dataset.py:54-127).

The reference's dataset item is the full rate-coded raster ([T,1,H,W] floats, 6-16 MB per query,
made in 8 worker processes).  Here an item is just the uint8 event-count frame: normalisation,
raster and centre-pixel selection happen inside the feature kernel from the frame and the constant
threshold codes (lens_amd.engine.build_threshold_codes).  CSV handling follows the reference:
first column = file name, second = index; every `skip`-th row, then the first `max_samples`.
"""
import csv
import os

import numpy as np
import torch
from PIL import Image


class CustomImageDataset(torch.utils.data.Dataset):
    def __init__(self, annotations_file, img_dir, kernel_size=None, transform=None, target_transform=None,
                 skip=1, max_samples=None, test=True, is_spiking=False, time_window=33):
        with open(annotations_file, newline="") as fh:
            rows = [r for r in csv.reader(fh)][1:]
        rows = rows[::skip]
        if max_samples is not None:
            rows = rows[:max_samples]
        self.rows = rows
        self.img_dir = img_dir
        self.kernel_size, self.time_window, self.is_spiking, self.test = kernel_size, time_window, is_spiking, test

    def __len__(self):
        return len(self.rows)

    def __getitem__(self, idx):
        name, label = self.rows[idx][0], int(self.rows[idx][1])
        path = os.path.join(self.img_dir, name)
        if not os.path.exists(path):
            raise FileNotFoundError(f"No file found for index {idx} at {path}.")
        img = np.array(Image.open(path))
        if img.ndim != 2 or img.dtype != np.uint8:
            raise ValueError(f"{path}: expected an 8-bit single-channel image, got {img.dtype} {img.shape}")
        return torch.from_numpy(img), label, path, []

    def frames(self, workers=8, pin=False):
        """All frames as one uint8 tensor [Q,H,W] on the host.  PNG decoding runs in `workers` threads straight
        into one preallocated (optionally pinned) buffer -- the role of the reference's DataLoader(num_workers=8),
        without the 6-16 MB/query raster it ships between processes."""
        n = len(self)
        if n == 0:
            return torch.zeros((0, 0, 0), dtype=torch.uint8)
        first = self[0][0]
        out = torch.empty((n,) + tuple(first.shape), dtype=torch.uint8, pin_memory=bool(pin and torch.cuda.is_available()))
        out[0] = first

        def load(i):
            img = self[i][0]
            if img.shape != first.shape:
                raise ValueError(f"{self.rows[i][0]}: frame size {tuple(img.shape)} differs from {tuple(first.shape)}")
            out[i] = img
        if workers > 1 and n > 2:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(workers) as ex:
                list(ex.map(load, range(1, n)))
        else:
            for i in range(1, n):
                load(i)
        return out

    def frames_to_device(self, device, workers=8):
        """Decode into pinned memory and issue one asynchronous host-to-device copy."""
        return self.frames(workers=workers, pin=True).to(device, non_blocking=True)
