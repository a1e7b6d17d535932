from diffnet_amd.networks.autoencoders import *  # noqa: F401,F403
