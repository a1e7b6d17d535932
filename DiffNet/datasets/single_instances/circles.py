from diffnet_amd.datasets.single_instances.circles import *  # noqa: F401,F403
