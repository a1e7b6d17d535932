from diffnet_amd.fem import DiffNet2DFEM, DiffNet3DFEM, DiffNetFEM, gauss_pt_eval  # noqa: F401
