"""ctypes binding of libdiffnet_hip.so (the C ABI declared in include/diffnet_hip.h).

The library is the product: there is no CPU fallback.  `lib()` raises if the shared object is
missing or does not export the full ABI.  Loading needs no GPU (symbols are checked on CPU boxes
too); compute entry points need device pointers.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DN_LIB_PATH") or os.path.join(HERE, "libdiffnet_hip.so")     # DN_LIB_PATH: a variant build (tools/variant_build.sh)
ABI_VERSION = 9

DN_E = {-1: "DN_E_BADARG", -2: "DN_E_UNSUPPORTED", -3: "DN_E_WORKSPACE", -4: "DN_E_HANDOVER"}


class DnMesh(C.Structure):
    _fields_ = [("nsd", C.c_int32), ("degree", C.c_int32), ("ngp", C.c_int32), ("batch", C.c_int32),
                ("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32),
                ("scale", C.c_float * 3), ("gpw", C.c_float * 4),
                ("basis", (C.c_float * 4) * 4), ("dbasis", (C.c_float * 4) * 4)]


class DnDirichlet(C.Structure):
    _fields_ = [("mask", C.c_void_p), ("field", C.c_void_p), ("value", C.c_float),
                ("mask_kind", C.c_int32), ("mask_batched", C.c_int32), ("field_batched", C.c_int32),
                ("box_faces", C.c_int32), ("row_words", C.c_int32)]


MASK_F32, MASK_U8, MASK_BITS, MASK_BOX = 0, 1, 2, 3
FACE_XLO, FACE_XHI, FACE_YLO, FACE_YHI, FACE_ZLO, FACE_ZHI = 1, 2, 4, 8, 16, 32


class DnPoissonArgs(C.Structure):
    _fields_ = [("u", C.c_void_p), ("nu", C.c_void_p), ("f", C.c_void_p), ("f_gp", C.c_void_p),
                ("nu_batched", C.c_int32), ("f_batched", C.c_int32),
                ("bc", DnDirichlet * 2),
                ("alpha", C.c_float), ("beta", C.c_float), ("c", C.c_float), ("wscale", C.c_float),
                ("out_scale", C.c_float),
                ("out", C.c_void_p), ("energy", C.c_void_p), ("sumsq", C.c_void_p),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
                ("energy_f32", C.c_void_p), ("energy_scale", C.c_double), ("strip_select", C.c_int32), ("accumulate_sums", C.c_int32), ("defer_sums", C.c_int32),
                ("fold_prev", C.c_void_p), ("f_is_load", C.c_int32)]


class DnFsdtArgs(C.Structure):
    _fields_ = [("w", C.c_void_p), ("phi_x", C.c_void_p), ("phi_y", C.c_void_p),
                ("bc_mask", C.c_void_p), ("mask_is_u8", C.c_int32), ("mask_batched", C.c_int32),
                ("bc_field", C.c_void_p * 3), ("bc_field_batched", C.c_int32 * 3), ("bc_value", C.c_float * 3),
                ("D11", C.c_float), ("D12", C.c_float), ("D22", C.c_float), ("D66", C.c_float), ("A44", C.c_float),
                ("A55", C.c_float), ("q", C.c_float), ("wscale", C.c_float),
                ("out", C.c_void_p * 3), ("sumsq", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
                ("in_scale", C.c_void_p), ("norms", C.c_void_p), ("in_num", C.c_void_p), ("in_den", C.c_void_p),
                ("defer_sums", C.c_int32), ("den_ticket", C.c_int32), ("den_workspace", C.c_void_p)]


I32x3 = C.c_int32 * 3

# name -> (restype, argtypes); must list every symbol of include/diffnet_hip.h
SYMBOLS = {
    "dn_abi_version": (C.c_int, []),
    "dn_build_info": (C.c_char_p, []),
    "dn_config_set": (C.c_int, [C.c_char_p, C.c_char_p]),
    "dn_config_get": (C.c_char_p, [C.c_char_p]),
    "dn_probe_stream": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "dn_probe_march": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "dn_probe_tile": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "dn_poisson_workspace_bytes": (C.c_int64, [C.POINTER(DnMesh)]),
    "dn_poisson_apply": (C.c_int, [C.POINTER(DnMesh), C.POINTER(DnPoissonArgs), C.c_void_p]),
    "dn_poisson_finish_sums": (C.c_int, [C.POINTER(DnMesh), C.POINTER(DnPoissonArgs), C.c_void_p]),
    "dn_workspace_status": (C.c_int, [C.c_void_p, C.c_void_p]),
    "dn_gauss_pt_eval_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, I32x3, C.c_int32,
                                       C.c_int32, C.c_int32, C.c_void_p]),
    "dn_gauss_pt_eval_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, I32x3, C.c_int32,
                                       C.c_int32, C.c_int32, C.c_void_p]),
    "dn_assemble": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, I32x3, C.c_int32, C.c_int32, C.c_int32,
                              C.c_void_p]),
    "dn_assemble_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, I32x3, C.c_int32, C.c_int32,
                                  C.c_void_p]),
    "dn_fdm_stencil_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_float), C.c_int32,
                                     C.c_float, C.c_float, C.c_void_p]),
    "dn_fdm_stencil_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_float), C.c_int32,
                                     C.c_float, C.c_float, C.c_void_p]),
    "dn_fdm_fused_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_float), C.c_int32,
                                   C.c_float, C.c_float, C.c_void_p]),
    "dn_fdm_fused_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_float), C.c_int32,
                                   C.c_float, C.c_float, C.c_void_p]),
    "dn_fsdt_workspace_bytes": (C.c_int64, [C.POINTER(DnMesh)]),
    "dn_fsdt_apply": (C.c_int, [C.POINTER(DnMesh), C.POINTER(DnFsdtArgs), C.c_void_p]),
    "dn_upconv_out_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int64, C.c_int64, C.c_int64]),
    "dn_upconv_out_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int,
                                    C.c_void_p, C.c_int64, C.c_void_p]),
    "dn_upconv_out_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                    C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    "dn_upconv3d_out_workspace_bytes": (C.c_int64, [C.c_int64] * 5),
    "dn_upconv3d_out_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                      C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    "dn_upconv3d_out_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                      C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    "dn_conv3d_k4s2_wrw_workspace_bytes": (C.c_int64, [C.c_int64] * 6),
    "dn_conv3d_k4s2_wrw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                     C.c_void_p, C.c_int64, C.c_void_p]),
    "dn_conv2d_k4s2_down": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int64] * 5 + [C.c_void_p]),
    "dn_conv2d_k4s2_up": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int64] * 5 + [C.c_void_p]),
    "dn_conv2d_k4s2_wrw_workspace_bytes": (C.c_int64, [C.c_int64] * 5),
    "dn_conv2d_k4s2_wrw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int64] * 5 + [C.c_void_p, C.c_int64, C.c_void_p]),
    "dn_conv3d_k4s2_down": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int64] * 6 + [C.c_void_p]),
    "dn_conv3d_k4s2_up": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int64] * 6 + [C.c_void_p]),
    "dn_conv3d_k4s2_workspace_bytes": (C.c_int64, [C.c_int32] + [C.c_int64] * 6),
    "dn_conv3d_k4s2_down_ws": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int64] * 6 + [C.c_void_p, C.c_int64, C.c_void_p]),
    "dn_conv3d_k4s2_up_ws": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int64] * 6 + [C.c_void_p, C.c_int64, C.c_void_p]),
    "dn_conv2d_valid_fwd": (C.c_int, [C.c_void_p] * 4 + [C.c_int64] * 6 + [C.c_void_p]),
    "dn_conv2d_valid_bwd_data": (C.c_int, [C.c_void_p] * 3 + [C.c_int64] * 6 + [C.c_void_p]),
    "dn_conv2d_valid_bwd_weight": (C.c_int, [C.c_void_p] * 4 + [C.c_int64] * 6 + [C.c_void_p]),
    "dn_instnorm_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int64]),
    "dn_instnorm_act_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_float, C.c_float,
                                      C.c_void_p, C.c_int64, C.c_void_p]),
    "dn_instnorm_act_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_float,
                                      C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    "dn_pack_mask_bits": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "dn_unpack_mask_bits": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "dn_winding_nodes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                   C.c_void_p]),
}

_LIB = None


class DiffNetHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raises DiffNetHipError when the library is unusable."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise DiffNetHipError(
            f"{LIB_PATH} is missing: build it with `python -m diffnet_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback for the FEM hot path.")
    try:
        h = C.CDLL(LIB_PATH)
    except OSError as e:   # pragma: no cover - depends on the box
        raise DiffNetHipError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(h, name)
        except AttributeError as e:
            raise DiffNetHipError(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.restype = res
        fn.argtypes = args
    v = h.dn_abi_version()
    if v != ABI_VERSION:
        raise DiffNetHipError(f"ABI mismatch: library {v}, binding {ABI_VERSION}")
    _LIB = h
    return h


_HIP = None


def hip_runtime():
    """ctypes handle of the HIP runtime torch runs on (the instance mapped into this process), for the few calls torch does not expose
    with the flags needed: events without the system-scope fence of a default event (1.8 us per record, tools/event_cost.py)."""
    global _HIP
    if _HIP is None:
        path = next((ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64.so" in ln), None)
        if path is None:
            raise DiffNetHipError("the HIP runtime is not loaded in this process (no GPU build of torch?)")
        h = C.CDLL(path)
        h.hipEventCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
        h.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
        h.hipStreamWaitEvent.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
        h.hipEventSynchronize.argtypes = [C.c_void_p]
        h.hipEventDestroy.argtypes = [C.c_void_p]
        _HIP = h
    return _HIP


def new_event():
    """A HIP event for stream ordering only: hipEventDisableTiming | hipEventDisableSystemFence."""
    e = C.c_void_p()
    rc = hip_runtime().hipEventCreateWithFlags(C.byref(e), C.c_uint(0x2 | 0x20000000))
    if rc != 0:
        raise DiffNetHipError(f"hipEventCreateWithFlags: hipError_t {rc}")
    return e


def check(rc, what):
    if rc == 0:
        return
    if rc < 0:
        if rc == -4:
            raise DiffNetHipError(f"{what}: DN_E_HANDOVER (a chained-strip launch ran a bounded LDS hand-over poll to its limit: its results are NaN)")
        raise DiffNetHipError(f"{what}: {DN_E.get(rc, rc)} (argument not supported by the HIP kernels)")
    raise DiffNetHipError(f"{what}: hipError_t {rc}")


# Python-side mirror of the switches (the library reads DN_<KEY> once at load, then only dn_config_set changes them): lets host code
# that is traced by torch.compile look a switch up without a foreign call
CONFIG_MIRROR = {k[3:]: v for k, v in os.environ.items() if k.startswith("DN_")}


def config_set(key, value):
    """Set a tuning / A-B switch of the library (include/diffnet_hip.h: dn_config_set); "" or None clears it."""
    rc = lib().dn_config_set(key.encode(), (value or "").encode())
    if rc != 0:
        raise DiffNetHipError(f"dn_config_set: unknown switch {key!r} or value too long")
    CONFIG_MIRROR[key] = value or ""
    import sys
    ops = sys.modules.get(__package__ + ".ops")
    if ops is not None:
        ops.call_cache_clear()      # launch plans (and with them workspace sizes) may have changed


def config_get(key):
    v = lib().dn_config_get(key.encode())
    if v is None:
        raise DiffNetHipError(f"dn_config_get: unknown switch {key!r}")
    return v.decode()
