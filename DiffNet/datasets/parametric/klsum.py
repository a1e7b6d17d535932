from diffnet_amd.datasets.parametric.klsum import *  # noqa: F401,F403
