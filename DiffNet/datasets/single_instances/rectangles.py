from diffnet_amd.datasets.single_instances.rectangles import *  # noqa: F401,F403
