from diffnet_amd.datasets.parametric.pointclouds import *  # noqa: F401,F403
