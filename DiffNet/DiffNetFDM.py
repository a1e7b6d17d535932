from diffnet_amd.fdm import DiffNetFDM, get_deriv_kernels, get_sobel_correction_matrix  # noqa: F401
