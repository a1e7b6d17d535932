from diffnet_amd.datasets.single_instances.cuboids import *  # noqa: F401,F403
