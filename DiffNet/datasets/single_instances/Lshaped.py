from diffnet_amd.datasets.single_instances.Lshaped import *  # noqa: F401,F403
