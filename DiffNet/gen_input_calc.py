from diffnet_amd.gen_input_calc import *  # noqa: F401,F403
