from diffnet_amd.datasets.single_instances.klsum import *  # noqa: F401,F403
