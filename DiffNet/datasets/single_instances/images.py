from diffnet_amd.datasets.single_instances.images import *  # noqa: F401,F403
