from diffnet_amd.networks.wgan3d import *  # noqa: F401,F403
