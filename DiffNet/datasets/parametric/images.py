from diffnet_amd.datasets.parametric.images import *  # noqa: F401,F403
