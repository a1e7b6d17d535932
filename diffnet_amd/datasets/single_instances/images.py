"""Immersed objects read from a binary image (reference: DiffNet/datasets/single_instances/images.py:9-79)."""
import os

import numpy as np

from .. import GridDataset, faces

IMAGE_EXTENSIONS = ('.png', '.jpg', '.bmp', '.tiff')


def object_channels(filename, sink_faces=("all",)):
    """[domain, bc1, bc2] from a grey image: non-zero pixels are the object (domain = 0, u = 1 there), the named outer
    faces are sinks."""
    import PIL.Image
    if os.path.splitext(filename)[1] not in IMAGE_EXTENSIONS:
        raise ValueError('invalid extension; extension not supported')
    img = (np.asarray(PIL.Image.open(filename).convert('L')) > 0).astype('float')
    domain = 1 - img
    bc1 = np.zeros_like(domain)
    bc1[(1 - domain).astype('bool')] = 1
    return domain, bc1, faces(domain.shape, *sink_faces)


class ImageIMBack(GridDataset):
    """Zero forcing (images.py:9-42)."""

    def __init__(self, filename, domain_size=64):
        self.domain, self.bc1, self.bc2 = object_channels(filename)
        self.n_samples = 100


class Disk(ImageIMBack):
    """Same geometry handling, unit forcing (images.py:44-79)."""

    def __init__(self, filename, domain_size=64):
        super().__init__(filename, domain_size)
        self.forcing = np.ones_like(self.domain)
