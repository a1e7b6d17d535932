from diffnet_amd.datasets.parametric.topo3d import *  # noqa: F401,F403
