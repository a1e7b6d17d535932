from diffnet_amd.networks.unets import *  # noqa: F401,F403
