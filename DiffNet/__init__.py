"""Import-path alias: `DiffNet.*` resolves to the MI355X implementation in `diffnet_amd`, so scripts written
against the reference package (`from DiffNet.DiffNetFEM import DiffNet2DFEM`, ...) run unchanged."""
from diffnet_amd import __version__  # noqa: F401
