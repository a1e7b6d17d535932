from diffnet_amd.datasets.single_instances.voxels import *  # noqa: F401,F403
