"""Boundary point clouds of immersed objects (reference: the `PointClouds` dataset defined in
IBN/poisson-2d/parametric/IBN_2D.py:35-84; on-disk format `point_cloud.npz` / `normals.npz`, key 'arr_0', shapes
(samples, points, 2|3)).  A sample is (points | normals | area) per boundary point, the zero forcing and the outer-boundary
sink mask; the inside/outside field is computed on the GPU by `compute_winding_nodes` (dn_winding_nodes)."""
import numpy as np
import torch
from torch.utils import data

from .. import faces


def segment_area(pc):
    """Per-point weight of a closed polyline: half the SQUARED length of the two adjacent segments (the reference sums
    squared differences without a square root; reproduced as is), end points wrapping around."""
    sq = lambda a, b: np.sum((a - b) ** 2, -1) * 0.5
    area = np.zeros((pc.shape[0], pc.shape[1], 1))
    area[:, 1:-1, 0] = sq(pc[:, 1:-1], pc[:, 0:-2]) + sq(pc[:, 2:], pc[:, 1:-1])
    area[:, 0, 0] = sq(pc[:, 1], pc[:, 0]) + sq(pc[:, -1], pc[:, 0])
    area[:, -1, 0] = sq(pc[:, -1], pc[:, -2]) + sq(pc[:, -1], pc[:, 0])
    return area


class PointClouds(data.Dataset):
    """`data_path` is a path prefix (the reference concatenates file names onto it); the first 1250 shapes are the
    validation split.  Points are scaled by 0.5 and shifted by (0.25, 0.5) as in the reference (whose preceding
    normalisation divides by `np.max(...).any()`, i.e. by True -- a no-op that is kept)."""

    n_val = 1250

    def __init__(self, data_path, type='train', domain_size=32):
        points = np.load(data_path + 'point_cloud.npz')['arr_0']
        normals = np.load(data_path + 'normals.npz')['arr_0']
        if type == 'val':
            points, normals = points[:self.n_val], normals[:self.n_val]
        elif type == 'train':
            points, normals = points[self.n_val:], normals[self.n_val:]
        points = points.copy()
        points[:, :, 0] = points[:, :, 0] / np.max(points[:, :, 0]).any()
        points[:, :, 1] = points[:, :, 1] / np.max(points[:, :, 1]).any()
        points = points * 0.5
        points[:, :, 0] += 0.25
        points[:, :, 1] += 0.5
        self.domain = np.ones((domain_size, domain_size))
        self.domain_size = domain_size
        self.n_samples = points.shape[0]
        self.normals = normals[:, :, :2]
        self.pc = points
        self.area = segment_area(self.pc)
        self.bc2 = faces((domain_size, domain_size), "all")

    def __len__(self):
        return self.n_samples

    def __getitem__(self, index):
        inputs = np.concatenate((self.pc[index], self.normals[index], self.area[index]), -1)
        forcing = np.zeros_like(self.domain)
        return torch.FloatTensor(inputs), torch.FloatTensor(forcing).unsqueeze(0), torch.FloatTensor(self.bc2).unsqueeze(0)


def write_star_shapes(prefix, n_shapes=64, n_points=200, seed=0):
    """Synthetic stand-in for the reference's shape library: closed star-shaped curves r(t) = r0 (1 + sum a_k cos(k t + p_k))
    in the unit box, with outward unit normals, written in the reference's npz layout.  Returns (points, normals)."""
    g = np.random.RandomState(seed)
    t = np.linspace(0.0, 2.0 * np.pi, n_points, endpoint=False)
    pts = np.zeros((n_shapes, n_points, 2))
    nrm = np.zeros((n_shapes, n_points, 3))
    for s in range(n_shapes):
        r, dr = np.full_like(t, 0.25), np.zeros_like(t)
        for k in range(2, 5):
            a, p = 0.12 * g.rand() / k, 2 * np.pi * g.rand()
            r = r + 0.25 * a * np.cos(k * t + p)
            dr = dr - 0.25 * a * k * np.sin(k * t + p)
        x, y = 0.5 + r * np.cos(t), 0.5 + r * np.sin(t)
        tx, ty = dr * np.cos(t) - r * np.sin(t), dr * np.sin(t) + r * np.cos(t)      # tangent; outward normal = (ty, -tx)
        ln = np.sqrt(tx * tx + ty * ty)
        pts[s, :, 0], pts[s, :, 1] = x, y
        nrm[s, :, 0], nrm[s, :, 1] = ty / ln, -tx / ln
    # invert the dataset's affine map p -> 0.5 p + (0.25, 0.5), so that the loaded shapes sit where they were drawn
    raw = np.stack([(pts[:, :, 0] - 0.25) / 0.5, (pts[:, :, 1] - 0.5) / 0.5], -1)
    np.savez(prefix + 'point_cloud.npz', raw)
    np.savez(prefix + 'normals.npz', nrm)
    return raw, nrm
