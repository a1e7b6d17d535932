"""Structured cuboid node grids in the reference's (P, N, M) = (z, y, x) layout
(DiffNet/cuboid_mesh.py:8-19)."""
import numpy as np


class CuboidMesh:
    def __init__(self, arg=None):
        pass

    @staticmethod
    def meshgrid_3d(x_1d, y_1d, z_1d):
        """Returns x_3d, y_3d, z_3d, each of shape (len(z), len(y), len(x))."""
        x_1d, y_1d, z_1d = (np.asarray(a) for a in (x_1d, y_1d, z_1d))
        shape = (z_1d.shape[0], y_1d.shape[0], x_1d.shape[0])
        x_3d = np.broadcast_to(x_1d[None, None, :], shape).copy()
        y_3d = np.broadcast_to(y_1d[None, :, None], shape).copy()
        z_3d = np.broadcast_to(z_1d[:, None, None], shape).copy()
        return x_3d, y_3d, z_3d
