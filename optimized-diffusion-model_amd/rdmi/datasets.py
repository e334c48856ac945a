"""Training-data side of the path: the reference's GTOHaloImageDataset (RD/datasets.py:82-98) pads each
67-vector to 81 values, normalises with (x - 0.4652) / 0.1811 and reshapes to [1,9,9] per item on 4 CPU workers.
Here the table lives in HBM and a batch is ONE HIP kernel (gather + pad + normalise + label), fed by device-side indices.

The reference reads a pickled numpy array; this module takes the array itself or a .npy file (numpy.load with
allow_pickle=False) -- pickles are never loaded.
"""
import numpy as np
import torch

from . import _native


class GTOHaloImageDataset:
    """Same items as RD/datasets.py:82-98 (`ds[i]` -> (img [1,9,9], label [1])), plus whole batches on the device."""
    mean = 0.4652
    std = 0.1811

    def __init__(self, data, device, image_size=9, image_width=None):
        if isinstance(data, str):
            data = np.load(data, allow_pickle=False)
        data = np.ascontiguousarray(np.asarray(data, dtype=np.float32))
        if data.ndim != 2:
            raise ValueError('expected a [num_items, vector_length] table, got shape %r' % (data.shape,))
        self.H, self.W = image_size, image_width or image_size
        if data.shape[1] > self.H * self.W:
            raise ValueError('vectors of %d values do not fit a %dx%d image' % (data.shape[1], self.H, self.W))
        self.data = torch.from_numpy(data).to(device)

    def __len__(self):
        return self.data.shape[0]

    def batch(self, idx):
        """idx: int64 tensor [B] on the device (or None for the whole table) -> (images [B,1,H,W], labels [B,1])."""
        img, lab = _native.gto_pack(self.data, idx, self.H * self.W, self.mean, self.std)
        return img.view(-1, 1, self.H, self.W), lab.view(-1, 1)

    def __getitem__(self, i):
        img, lab = self.batch(torch.tensor([int(i) % len(self)], dtype=torch.int64, device=self.data.device))
        return img[0], lab[0]

    def sample_batch(self, batch_size, generator=None):
        """Uniform with replacement, indices drawn on the device (the reference's DataLoader shuffles per epoch)."""
        idx = torch.randint(len(self), (batch_size,), device=self.data.device, generator=generator)
        return self.batch(idx)

    def epoch(self, batch_size, generator=None, drop_last=True, rank=0, world_size=1):
        """One shuffled pass in batches, sharded like DistributedSampler (rank r takes every world_size-th index)."""
        perm = torch.randperm(len(self), device=self.data.device, generator=generator)[rank::world_size]
        stop = len(perm) - (len(perm) % batch_size if drop_last else 0)
        for s in range(0, stop, batch_size):
            yield self.batch(perm[s:s + batch_size].contiguous())
