"""Families of immersed objects, one image file per sample (reference: DiffNet/datasets/parametric/images.py:9-133)."""
import os

import numpy as np

from .. import StackedDataset
from ..single_instances.images import object_channels


class _ImageFolder(StackedDataset):
    sink_faces = ("all",)
    extra_faces = None

    def __init__(self, dirname, domain_size=64):
        samples = []
        for fname in sorted(os.listdir(dirname)):
            chans = list(object_channels(os.path.join(dirname, fname), self.sink_faces))
            if self.extra_faces is not None:
                from .. import faces
                chans.append(faces(chans[0].shape, *self.extra_faces))
            samples.append(np.array(chans))
        self.dataset = np.array(samples)
        self.n_samples = self.dataset.shape[0]


class ImageIMBack(_ImageFolder):
    """[domain, bc1 = object, bc2 = outer boundary], zero forcing (images.py:9-50)."""


class ImageIMBackObject(_ImageFolder):
    """Same channels, unit forcing (images.py:52-93)."""

    forcing_value = 1.0


class ImageIMBackNeumann(_ImageFolder):
    """Sinks on the first row / column only; the last row / column are returned as a fourth channel bc3
    (images.py:95-133)."""

    sink_faces = ((1, 0), (0, 0))
    extra_faces = ((0, -1), (1, -1))
