"""Voxelised 3-D objects, one `.npz` per sample (reference: the `TopoDataset3D` dataset defined in
IBN/poisson-3d/parametric/IBN_3D.py:76-106; each file holds 'arr_0' of shape (1, n, n, n), 1 inside the object).  A sample is
(source, sink, forcing): the object indicator, the outer-boundary mask and zeros."""
import os

import numpy as np
import torch
from torch.utils import data

from .. import faces


class TopoDataset3D(data.Dataset):
    """The first 100 directory entries are the training split, the next 25 the validation split (directory order, as the
    reference)."""

    def __init__(self, data_path, domain_size=32, mode='train'):
        self.samples = data_path
        ids = os.listdir(self.samples)
        self.list_IDs = ids[:100] if mode == 'train' else ids[100:125]
        self.domain = np.ones((domain_size, domain_size, domain_size))
        self.domain_size = domain_size
        self.bc2 = faces(self.domain.shape, "all")

    def __len__(self):
        return len(self.list_IDs)

    def __getitem__(self, index):
        sample = np.load(os.path.join(self.samples, str(self.list_IDs[index])))['arr_0']
        source = torch.FloatTensor(sample)
        sink = torch.FloatTensor(np.expand_dims(self.bc2, axis=0))
        return source, sink, torch.zeros_like(source)


def write_blob_objects(dirname, n_objects=16, domain_size=32, seed=0):
    """Synthetic stand-in for the reference's topology library: unions of a few ellipsoids away from the boundary."""
    g = np.random.RandomState(seed)
    n = domain_size
    t = (np.arange(n) + 0.5) / n
    zz, yy, xx = np.meshgrid(t, t, t, indexing="ij")
    os.makedirs(dirname, exist_ok=True)
    for k in range(n_objects):
        vox = np.zeros((n, n, n), dtype=np.float32)
        for _ in range(g.randint(1, 4)):
            c = 0.3 + 0.4 * g.rand(3)
            r = 0.08 + 0.12 * g.rand(3)
            vox[((zz - c[0]) / r[0]) ** 2 + ((yy - c[1]) / r[1]) ** 2 + ((xx - c[2]) / r[2]) ** 2 < 1.0] = 1.0
        np.savez(os.path.join(dirname, f"object_{k:04d}.npz"), vox[None])
