from diffnet_amd.base import PDE, HAVE_LIGHTNING  # noqa: F401
