"""Voxelised 3-D objects (reference: DiffNet/datasets/single_instances/voxels.py:8-64).

On-disk format: `<prefix>inouts.raw` = one uint8 per voxel (254 = inside), Fortran order; `<prefix>VoxelConfig.txt` = a
header line, then bounding-box min (3 floats), max (3 floats), divisions (3 ints), grid size (3 floats), and the two
voxel counts."""
import numpy as np

from .. import GridDataset, faces


def load_raw(fileName, **kwargs):
    with open(fileName + 'VoxelConfig.txt', 'r') as cfg:
        cfg.readline()
        bBoxMin = np.array([float(v) for v in cfg.readline().split()])
        bBoxMax = np.array([float(v) for v in cfg.readline().split()])      # parsed for format checking; not returned
        numDiv = np.array([int(v) for v in cfg.readline().split()])
        gridSize = np.array([float(v) for v in cfg.readline().split()])
        int(cfg.readline())
        int(cfg.readline())
    del bBoxMax
    inOut = np.fromfile(fileName + 'inouts.raw', dtype=np.dtype('uint8'))
    inOut = (inOut / 254.0 > 0.25).astype(float)
    return np.reshape(inOut, numDiv, order='F'), numDiv, gridSize, bBoxMin


class VoxelIMBackRAW(GridDataset):
    """The object is pasted at offset 32 into a cube of ones (domain = 0 inside it, u = 1 there); u = 0 on all six faces."""

    def __init__(self, filename, domain_size=64):
        vox = load_raw(filename)[0]
        n = domain_size
        self.domain = np.ones((n, n, n))
        self.domain[32:32 + vox.shape[0], 32:32 + vox.shape[1], 32:32 + vox.shape[2]] = 1 - vox
        self.bc1 = np.zeros_like(self.domain)
        self.bc1[(1 - self.domain).astype('bool')] = 1
        self.bc2 = faces((n, n, n), "all")
        self.n_samples = 100
