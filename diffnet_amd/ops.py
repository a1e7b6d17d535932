"""Torch-facing wrappers around the C ABI (include/diffnet_hip.h): device pointers + the current HIP
stream are handed to libdiffnet_hip.so; autograd.Function classes pair each forward kernel with
its adjoint kernel so user `loss()` bodies compose with ordinary torch ops.

No CPU fallback: every op requires fp32 CUDA(HIP) tensors and raises otherwise.
"""
import ctypes as C
import math

import torch
from torch.autograd.function import once_differentiable

from . import _lib
from ._lib import DnDirichlet, DnFsdtArgs, DnMesh, DnPoissonArgs, I32x3, DiffNetHipError


def _require(t, name, ndim=None, strict=False):
    """`strict` (prepared launches): the tensor itself must be usable -- a hidden .contiguous() copy would be read by every later
    launch instead of the caller's (since updated) tensor."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise DiffNetHipError(f"{name} is on {t.device}: the FEM ops run on the GPU only (no CPU fallback); move it with .cuda()")
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    if ndim is not None and t.dim() != ndim:
        raise ValueError(f"{name} must have {ndim} dims, got shape {tuple(t.shape)}")
    if strict and not t.is_contiguous():
        raise DiffNetHipError(f"{name} is not contiguous: a prepared launch (PoissonPlan) keeps the pointers it was given; pass a contiguous tensor")
    return t.contiguous()


def _raw_stream(device):
    """The current HIP stream of `device` as an integer handle (torch.cuda.current_stream() builds a Stream object: ~8 us per call,
    more than the rest of a cached launch)."""
    return torch._C._cuda_getCurrentRawStream(device.index if device.index is not None else torch.cuda.current_device())


def _stream(t):
    return C.c_void_p(_raw_stream(t.device))


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


# ------------------------------------------------------------------------------------------------
# generic gauss_pt_eval / assembly
# ------------------------------------------------------------------------------------------------
def _sizes_xyz(t, nsd):
    s = list(t.shape[2:])          # (z, y, x) order in memory
    s = s[::-1] + [1] * (3 - nsd)  # -> (x, y, z)
    return I32x3(*s)


def _gpe_fwd(x, tables, nsd, nbf, stride):
    x = _require(x, "tensor", nsd + 2)
    if x.shape[1] != 1:
        raise ValueError(f"gauss_pt_eval expects a single-channel field (B,1,...), got {tuple(x.shape)}")
    G = tables.shape[0]
    out_sp = [(n - nbf) // stride + 1 for n in x.shape[2:]]
    if min(out_sp) < 1:
        raise ValueError(f"field {tuple(x.shape)} smaller than the {nbf}-node element")
    out = torch.empty((x.shape[0], G, *out_sp), dtype=torch.float32, device=x.device)
    rc = _lib.lib().dn_gauss_pt_eval_fwd(_p(x), _p(tables), _p(out), x.shape[0], nsd, _sizes_xyz(x, nsd), nbf, stride, G, _stream(x))
    _lib.check(rc, "dn_gauss_pt_eval_fwd")
    return out


def _gpe_bwd(gout, tables, shape, nsd, nbf, stride):
    gout = _require(gout, "grad_output")
    gin = torch.empty(shape, dtype=torch.float32, device=gout.device)
    sizes = I32x3(*(list(shape[2:])[::-1] + [1] * (3 - nsd)))
    rc = _lib.lib().dn_gauss_pt_eval_bwd(_p(gout), _p(tables), _p(gin), shape[0], nsd, sizes, nbf, stride, tables.shape[0], _stream(gout))
    _lib.check(rc, "dn_gauss_pt_eval_bwd")
    return gin


class _GpeFn(torch.autograd.Function):
    """gauss_pt_eval in eager mode: the launch behind a plain autograd.Function (the registered operator's dispatch costs ~75 us of host time per
    call -- tools/bench_ops.py: 157 us for a forward + backward whose two kernels take 60 -- and an unchanged reference loss() makes eight such
    calls per step).  Linear: the backward is the adjoint operator, whose backward is this one again."""

    @staticmethod
    def forward(ctx, x, tables, nsd, nbf, stride):
        ctx.save_for_backward(tables)
        ctx.meta = (tuple(x.shape), nsd, nbf, stride)
        return _gpe_fwd(x, tables, nsd, nbf, stride)

    @staticmethod
    def backward(ctx, g):
        (tables,) = ctx.saved_tensors
        shape, nsd, nbf, stride = ctx.meta
        return _GaussPtEvalT.apply(g, tables, shape, nsd, nbf, stride), None, None, None, None


class _GpeTFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gout, tables, shape, nsd, nbf, stride):
        ctx.save_for_backward(tables)
        ctx.meta = (nsd, nbf, stride)
        return _gpe_bwd(gout, tables, tuple(shape), nsd, nbf, stride)

    @staticmethod
    def backward(ctx, gg):
        (tables,) = ctx.saved_tensors
        nsd, nbf, stride = ctx.meta
        return _GaussPtEval.apply(gg, tables, nsd, nbf, stride), None, None, None, None, None


def _wants_grad(t):
    return torch.is_grad_enabled() and isinstance(t, torch.Tensor) and t.requires_grad


class _GaussPtEval:
    """`gauss_pt_eval`: linear in the field, its backward is the adjoint operator `_GaussPtEvalT`, whose backward is this one again --
    differentiable to any order like the reference's conv formulation.  Under torch.compile / torch.export the registered operator
    diffnet_mi::gauss_pt_eval_fwd (diffnet_amd/torch_ops.py: an ordinary graph node); in eager mode the same launch behind a plain
    autograd.Function, or directly when nothing requires a gradient."""

    @staticmethod
    def apply(x, tables, nsd, nbf, stride):
        if torch.compiler.is_compiling():
            from . import torch_ops
            return torch_ops.gauss_pt_eval_fwd(x, tables, nsd, nbf, stride)
        if _wants_grad(x):
            return _GpeFn.apply(x, tables, nsd, nbf, stride)
        return _gpe_fwd(x, tables, nsd, nbf, stride)


class _GaussPtEvalT:
    """Adjoint of `_GaussPtEval` (element / Gauss-point cotangents -> nodal field): diffnet_mi::gauss_pt_eval_bwd under torch.compile."""

    @staticmethod
    def apply(gout, tables, shape, nsd, nbf, stride):
        if torch.compiler.is_compiling():
            from . import torch_ops
            return torch_ops.gauss_pt_eval_bwd(gout, tables, list(shape), nsd, nbf, stride)
        if _wants_grad(gout):
            return _GpeTFn.apply(gout, tables, tuple(shape), nsd, nbf, stride)
        return _gpe_bwd(gout, tables, tuple(shape), nsd, nbf, stride)


def stack_tables(N, nsd):
    """(G, nbf^nsd) contiguous fp32 table from a list / ParameterList of (1,1,*nbf) kernels."""
    ts = [t.detach().reshape(-1) for t in N]
    return torch.stack(ts, 0).contiguous().float()


def gauss_pt_eval(tensor, N, nsd=2, stride=1):
    """Drop-in for DiffNet/DiffNetFEM.py:7-18 on the GPU: all len(N) Gauss-point channels in one launch.
    `N`: sequence of (1,1,*nbf) kernels, or an already stacked (G, nbf^nsd) tensor."""
    if nsd not in (1, 2, 3):
        raise UnboundLocalError("nsd must be 1, 2 or 3")   # the reference fails with UnboundLocalError here
    if isinstance(N, torch.Tensor) and N.dim() == 2:
        tables = N
        nbf = round(N.shape[1] ** (1.0 / nsd))
    else:
        nbf = N[0].shape[-1]
        tables = stack_tables(N, nsd)
    if tables.device != tensor.device:
        tables = tables.to(tensor.device)
    return _GaussPtEval.apply(tensor, tables, nsd, nbf, stride)


def _assemble_raw(r_split, nsd, nbf, base):
    r_split = _require(r_split, "R_split", nsd + 2)
    stride = nbf - 1
    if r_split.shape[1] != nbf ** nsd:
        raise ValueError(f"R_split must have {nbf ** nsd} local-basis channels, got {r_split.shape[1]}")
    node_sp = [n * stride + 1 for n in r_split.shape[2:]]
    if base is not None:
        out = _require(base, "Aglobal", nsd + 2).clone()
        if list(out.shape[2:]) != node_sp or out.shape[0] != r_split.shape[0]:
            raise ValueError("Aglobal shape does not match R_split")
    else:
        out = torch.empty((r_split.shape[0], 1, *node_sp), dtype=torch.float32, device=r_split.device)
    rc = _lib.lib().dn_assemble(_p(r_split), _p(out), r_split.shape[0], nsd, _sizes_xyz(out, nsd), nbf, stride,
                                1 if base is not None else 0, _stream(out))
    _lib.check(rc, "dn_assemble")
    return out


def _assemble_bwd_raw(gout, shape, nsd, nbf):
    gout = _require(gout, "grad_output")
    gs = torch.empty(shape, dtype=torch.float32, device=gout.device)
    rc = _lib.lib().dn_assemble_bwd(_p(gout), _p(gs), shape[0], nsd, _sizes_xyz(gout, nsd), nbf, nbf - 1, _stream(gout))
    _lib.check(rc, "dn_assemble_bwd")
    return gs


def assemble(r_split, nsd, nbf=2, out=None):
    """Element->node assembly (Q1_2D/3D_vector_assembly of the reference scripts, any degree):
    returns `out + scatter_add(r_split)` (out = zeros when omitted).  Deterministic gather form; registered operator
    diffnet_mi::assemble (linear: its backward is the per-element gather, whose backward is the assembly)."""
    if torch.compiler.is_compiling():
        from . import torch_ops
        return torch_ops.assemble(r_split, nsd, nbf) if out is None else torch_ops.assemble_onto(r_split, out, nsd, nbf)
    if _wants_grad(r_split) or _wants_grad(out):
        return _AssembleFn.apply(r_split, out, nsd, nbf)
    return _assemble_raw(r_split, nsd, nbf, out)


class _AssembleFn(torch.autograd.Function):
    """Assembly in eager mode (see _GpeFn): backward = the per-element gather, whose backward is the assembly again."""

    @staticmethod
    def forward(ctx, r_split, base, nsd, nbf):
        ctx.meta = (tuple(r_split.shape), nsd, nbf, base is not None)
        return _assemble_raw(r_split, nsd, nbf, base)

    @staticmethod
    def backward(ctx, g):
        shape, nsd, nbf, has_base = ctx.meta
        gr = _assemble_gather(g, shape, nsd, nbf) if ctx.needs_input_grad[0] else None
        return gr, (g if has_base and ctx.needs_input_grad[1] else None), None, None


class _AssembleGatherFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g, shape, nsd, nbf):
        ctx.meta = (nsd, nbf)
        return _assemble_bwd_raw(g.contiguous(), tuple(shape), nsd, nbf)

    @staticmethod
    def backward(ctx, gg):
        nsd, nbf = ctx.meta
        return assemble(gg, nsd, nbf), None, None, None


def _assemble_gather(g, shape, nsd, nbf):
    if _wants_grad(g):
        return _AssembleGatherFn.apply(g, tuple(shape), nsd, nbf)
    return _assemble_bwd_raw(g.contiguous(), tuple(shape), nsd, nbf)


# ------------------------------------------------------------------------------------------------
# fused Poisson operator
# ------------------------------------------------------------------------------------------------
class Dirichlet:
    """u <- where(mask > 0.5, value, u); `value` a python float or a nodal field tensor."""

    def __init__(self, mask, value=0.0):
        self.mask = mask
        self.value = value


class PackedMask:
    """A Dirichlet mask held in HBM with ONE BIT per node (C ABI: DN_MASK_BITS, include/diffnet_hip.h) instead of the reference's
    fp32 image (`DiffNet/datasets/parametric/images.py:30`): 1/8 B per node in the loss instead of 4.  Build it once when the
    dataset is placed on the device: `PackedMask.pack(mask)` with mask (B|1, 1, *N) float32 / uint8 / bool on the GPU.  Usable
    wherever a mask is: the 2-D Q1 fused kernels read the bits directly, every other path gets the (cached) uint8 image."""

    def __init__(self, bits, shape):
        self.bits = bits                    # int32 (B, rows per sample, row_words)
        self.shape = tuple(shape)           # (B, 1, *N) of the image
        self._u8 = None

    @property
    def row_words(self):
        return int(self.bits.shape[-1])

    @staticmethod
    def pack(mask):
        if not isinstance(mask, torch.Tensor) or not mask.is_cuda:
            raise DiffNetHipError("PackedMask.pack needs a CUDA tensor")
        m = mask.to(torch.uint8) if mask.dtype == torch.bool else mask
        if m.dtype not in (torch.float32, torch.uint8):
            raise TypeError("Dirichlet mask must be float32, uint8 or bool")
        m = m.contiguous()
        nx = m.shape[-1]
        rows = m.numel() // nx
        rw = (nx + 31) // 32
        bits = torch.empty((m.shape[0], rows // m.shape[0], rw), dtype=torch.int32, device=m.device)
        rc = _lib.lib().dn_pack_mask_bits(m.data_ptr(), _lib.MASK_U8 if m.dtype == torch.uint8 else _lib.MASK_F32, rows, nx, rw,
                                          bits.data_ptr(), _stream(m))
        _lib.check(rc, "dn_pack_mask_bits")
        return PackedMask(bits, m.shape)

    def image(self):
        """The uint8 image (B|1, 1, *N), unpacked once."""
        if self._u8 is None:
            out = torch.empty(self.shape, dtype=torch.uint8, device=self.bits.device)
            nx = self.shape[-1]
            rc = _lib.lib().dn_unpack_mask_bits(self.bits.data_ptr(), out.numel() // nx, nx, self.row_words, out.data_ptr(), _stream(out))
            _lib.check(rc, "dn_unpack_mask_bits")
            self._u8 = out
        return self._u8


class LoadVector:
    """The forcing of a sample as its ASSEMBLED load vector b_a = sum_e sum_g w_g N_a(g) f_g instead of nodal values f (include/diffnet_hip.h:
    dn_poisson_args.f_is_load).  The reference interpolates the nodal forcing to the Gauss points in every loss evaluation
    (IBN/poisson-2d/parametric/IBN_2D.py:128-130); the forcing of a sample does not change between training steps, so -- like
    `PackedMask.pack` for the Dirichlet masks -- it can be assembled ONCE when the dataset is placed on the device:
    `LoadVector.assemble(geom, f)`.  Passed as the `f` argument of energy_loss / energy_loss_and_grad / residual* / PoissonPlan; the
    kernel then spends one FMA per node on the forcing instead of the element arithmetic (3-D Q1 256^3: 114 -> 92 us).  Results equal
    the nodal-forcing call to rounding.  Taken by the 3-D Q1 two-element kernel only (even nx, 2-point rule, constant-value uint8 /
    float32 mask images); anything else raises DiffNetHipError (DN_E_UNSUPPORTED)."""

    def __init__(self, tensor):
        if not (isinstance(tensor, torch.Tensor) and tensor.is_cuda and tensor.dtype == torch.float32 and tensor.is_contiguous()):
            raise DiffNetHipError("LoadVector needs a contiguous float32 CUDA tensor (B|1, 1, *N)")
        self.tensor = tensor
        self.shape = tensor.shape

    @staticmethod
    def assemble(geom, f):
        """b = (rule weights, no Jacobian factor) mass-type assembly of the nodal forcing f (B|1, 1, *N): one launch of the fused operator
        itself with u = 0 (out = -beta * b)."""
        if not (isinstance(f, torch.Tensor) and f.is_cuda):
            raise DiffNetHipError("LoadVector.assemble needs a CUDA tensor")
        f = f.detach().to(torch.float32).contiguous()
        zero = torch.zeros_like(f)
        out, _ = poisson_apply(geom, zero, None, f, None, (), alpha=0.0, beta=-1.0, c=0.0, wscale=1.0, out_scale=1.0, want_out=True, want_sums=False)
        return LoadVector(out)


class BoxFaces:
    """Dirichlet condition on faces of the domain box, derived from the geometry (C ABI: DN_MASK_BOX): no mask array is read at
    all.  The reference builds these masks as images (`IBN/poisson-2d/parametric/IBN_2D.py:69-73`, `rectangles.py:16,232-233`).
    faces: "all" or any of "xlo", "xhi", "ylo", "yhi", "zlo", "zhi" (x = last tensor axis)."""
    _BITS = {"xlo": _lib.FACE_XLO, "xhi": _lib.FACE_XHI, "ylo": _lib.FACE_YLO, "yhi": _lib.FACE_YHI, "zlo": _lib.FACE_ZLO, "zhi": _lib.FACE_ZHI}

    def __init__(self, faces="all"):
        if isinstance(faces, str):
            faces = list(self._BITS) if faces == "all" else [faces]
        self.bits = 0
        for f in faces:
            self.bits |= self._BITS[f]
        self._img = {}

    def image(self, node_shape, device):
        """The uint8 image (1, 1, *N) of the selected faces (cached per shape and device)."""
        key = (tuple(node_shape), str(device))
        img = self._img.get(key)
        if img is None:
            nsd = len(node_shape)
            img = torch.zeros((1, 1, *node_shape), dtype=torch.uint8, device=device)
            for ax, name in zip(range(nsd - 1, -1, -1), "xyz"):       # x is the last tensor axis
                idx = [slice(None)] * (nsd + 2)
                if self.bits & self._BITS[name + "lo"]:
                    idx[2 + ax] = 0; img[tuple(idx)] = 1
                if self.bits & self._BITS[name + "hi"]:
                    idx[2 + ax] = -1; img[tuple(idx)] = 1
            self._img[key] = img
        return img


def _mask_image(m, like):
    """Any mask form -> tensor image usable in torch.where against `like` (B,1,*N)."""
    if isinstance(m, PackedMask):
        return m.image()
    if isinstance(m, BoxFaces):
        return m.image(like.shape[2:], like.device)
    return m


# ---- mask images packed on first use (round 4) ------------------------------------------------------------------------------------
# The reference keeps Dirichlet masks as fp32 images and hands them to loss() every step (IBN_2D.py:69-73, 119-121); read as they are they
# cost the fused 2-D kernel 4 B per node and launch (72 against 54-56 us at 512^2 x 64), as uint8 images 1 B.  A one-shot call
# (energy_loss / energy_loss_and_grad / residual*) therefore packs a mask image to one bit per node the FIRST time it sees it and finds the
# bits again on later calls: the cache is keyed on the mask's storage (address, offset, shape, strides, dtype) AND its version counter -- an
# in-place write to the mask (or to any view of its storage) changes the version and the image is packed again -- and an entry holds the
# mask tensor itself, so that the address cannot be handed to another tensor while the entry lives.  Small LRU; `AUTO_PACK_MASKS = False`
# turns it off.  Prepared launches (PoissonPlan) take what they are given: pack explicitly with PackedMask.pack there.
AUTO_PACK_MASKS = True
_PACK_CACHE = __import__("collections").OrderedDict()
_PACK_CACHE_MAX = 8
_PACK_STATS = {"hit": 0, "pack": 0}


def _packed_on_first_use(m):
    key = (m.untyped_storage().data_ptr(), m.storage_offset(), tuple(m.shape), tuple(m.stride()), m.dtype)
    with _WS_LOCK:
        ent = _PACK_CACHE.get(key)
        if ent is not None and ent[1] == m._version:
            _PACK_CACHE.move_to_end(key)
            _PACK_STATS["hit"] += 1
            return ent[2]
    pm = PackedMask.pack(m)
    _PACK_STATS["pack"] += 1
    with _WS_LOCK:
        _PACK_CACHE[key] = (m, m._version, pm)
        _PACK_CACHE.move_to_end(key)
        while len(_PACK_CACHE) > _PACK_CACHE_MAX:
            _PACK_CACHE.popitem(last=False)
    return pm


def _auto_pack(geom, u, f_gp, dl):
    """The conditions of a one-shot 2-D Q1 call with their mask IMAGES replaced by (cached) bit-packed masks, where the compact kernels
    take the whole call: every condition a constant value on a PackedMask / BoxFaces / contiguous float32, uint8 or bool CUDA image."""
    if not (AUTO_PACK_MASKS and geom.nsd == 2 and geom.deg == 1 and f_gp is None and len(dl) > 0 and isinstance(u, torch.Tensor) and u.is_cuda):
        return dl
    if _lib.CONFIG_MIRROR.get("Q1_RULE_KERNEL"):
        return dl
    want = (1, *geom.node_shape)
    for d in dl:
        m = d.mask
        if isinstance(d.value, torch.Tensor):
            return dl
        if isinstance(m, (PackedMask, BoxFaces)):
            continue
        if not (isinstance(m, torch.Tensor) and m.is_cuda and m.device == u.device and m.is_contiguous() and not m.requires_grad and
                m.dtype in (torch.float32, torch.uint8, torch.bool) and m.dim() == u.dim() and tuple(m.shape[1:]) == want and
                m.shape[0] in (1, u.shape[0])):
            return dl
    return [d if isinstance(d.mask, (PackedMask, BoxFaces)) else Dirichlet(_packed_on_first_use(d.mask), d.value) for d in dl]


def _norm_dirichlet(dirichlet):
    out = []
    for d in dirichlet or ():
        if isinstance(d, Dirichlet):
            out.append(d)
        else:
            m, v = d
            out.append(Dirichlet(m, v))
    if len(out) > 2:
        raise ValueError("at most two Dirichlet conditions per call are supported; merge masks beforehand")
    return out


_threading = __import__("threading")
_WS = {}
_WS_LOCK = _threading.Lock()
_FSDT_LAUNCH_LOCK = _threading.Lock()
_POISSON_WS_BYTES = {}


_SIDE_STREAMS = {}


def _side_stream(dev):
    """One side stream per device for the deferred final reductions (PoissonPlan(async_sums=True))."""
    s = _SIDE_STREAMS.get(dev.index)
    if s is None:
        s = _SIDE_STREAMS[dev.index] = torch.cuda.Stream(dev)
    return s


def _workspace(dev, nbytes):
    """The reduction workspace of (device, current stream): launches on one stream are ordered, so they may share it; a launch on
    another stream gets its own (the ABI asks for one workspace per concurrently used stream).  A prepared launch (PoissonPlan)
    keeps the workspace of the stream it was prepared on and must be launched on that stream."""
    key = (dev.index, _raw_stream(dev))
    with _WS_LOCK:
        ws = _WS.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = torch.zeros(max(nbytes, 1 << 16), dtype=torch.uint8, device=dev)   # ABI: zero-filled once
            _WS[key] = ws
    return ws


def workspace_status(device=None):
    """Health check of the shared launch workspaces of `device` (every stream that has used one): synchronises those streams and raises
    DiffNetHipError if a launch ran a bounded LDS hand-over poll to its limit (chained-strip plans; such a launch's outputs and sums are
    NaN -- never silently wrong).  Cheap enough for once per epoch; not for the launch path."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    with _WS_LOCK:
        items = [(k, w) for k, w in _WS.items() if k[0] == dev.index]
    for (_, stream), ws in items:
        _lib.check(_lib.lib().dn_workspace_status(C.c_void_p(ws.data_ptr()), C.c_void_p(stream)), "dn_workspace_status")


# ---- one-shot calls: prepared launches cached behind the public API ------------------------------------------------------------
# energy_loss / energy_loss_and_grad / residual* evaluate the operator on the same buffers step after step (a training loop's network
# output lands in the same allocator block every iteration).  Validation and the argument structs of such a call are kept in a small
# LRU keyed on everything they depend on -- pointers, shapes, condition forms, coefficients, stream -- so that a repeat costs the key,
# the output allocations and one ctypes call instead of the full preparation (22 -> ~8 us of host time: small meshes are bounded by
# the device, not by Python).  An entry holds NO reference to the caller's tensors (a pointer + shape is all a launch needs, and the
# caller passes live tensors to every call); outputs are allocated fresh per call (a cached call never overwrites an earlier result).
_CALL_CACHE = __import__("collections").OrderedDict()
_CALL_CACHE_MAX = 16
_CALL_STATS = {"hit": 0, "miss": 0, "uncached": 0}


def _tkey(t):
    """What a launch depends on for a tensor argument that is used as it is; None when it would need a conversion copy."""
    if t is None:
        return 0
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.is_contiguous()):
        return None
    return (t.data_ptr(), t.dtype, tuple(t.shape))


def _call_key(geom, u, nu, f, f_gp, dl, scal, want_out, want_sums, out):
    parts = [geom.key, u.device.index, _raw_stream(u.device), scal, want_out, want_sums, isinstance(f, LoadVector)]
    if isinstance(f, LoadVector):
        f = f.tensor
    for t in (u, nu, f, f_gp, out):
        k = _tkey(t)
        if k is None or (k != 0 and k[1] != torch.float32):
            return None
        parts.append(k)
    for d in dl:
        m, v = d.mask, d.value
        if isinstance(m, PackedMask):
            mk = ("bits", m.bits.data_ptr(), m.shape)
        elif isinstance(m, BoxFaces):
            mk = ("box", m.bits)
        else:
            mk = _tkey(m)
            if mk is None or mk == 0 or mk[1] not in (torch.float32, torch.uint8):
                return None
        if isinstance(v, torch.Tensor):
            vk = _tkey(v)
            if vk is None or vk[1] != torch.float32 or len(vk[2]) != u.dim():
                return None
        else:
            vk = float(v)
        parts.append((mk, vk))
    return tuple(parts)


def poisson_apply(geom, u, nu=None, f=None, f_gp=None, dirichlet=(), alpha=1.0, beta=1.0, c=1.0, wscale=1.0,
                  out_scale=1.0, want_out=True, want_sums=True, loss_scale=None, out=None):
    """One launch of dn_poisson_apply.  Returns (out | None, sums | None) where sums is a float64 device
    tensor [energy, sum(out_unscaled^2)].  With `loss_scale` a third value is returned: the 0-dim float32 tensor
    energy * loss_scale written by the same launch.  See include/diffnet_hip.h for the operator definition."""
    dl = _auto_pack(geom, u, f_gp, _norm_dirichlet(dirichlet))
    key = None
    if isinstance(u, torch.Tensor) and u.is_cuda:
        key = _call_key(geom, u, nu, f, f_gp, dl, (float(alpha), float(beta), float(c), float(wscale), float(out_scale),
                                                   None if loss_scale is None else float(loss_scale)), want_out, want_sums, out)
    if key is None:            # something needs a conversion copy (or is invalid: the full preparation says what)
        _CALL_STATS["uncached"] += 1
        return PoissonPlan(geom, u, nu, f, f_gp, dl, alpha, beta, c, wscale, out_scale, want_out, want_sums, loss_scale, out, strict=False).launch()
    with _WS_LOCK:
        ent = _CALL_CACHE.get(key)
        if ent is not None:
            _CALL_CACHE.move_to_end(key)
    if ent is None:
        _CALL_STATS["miss"] += 1
        plan = PoissonPlan(geom, u, nu, f, f_gp, dl, alpha, beta, c, wscale, out_scale, want_out, want_sums, loss_scale, out, strict=True)
        res = plan.launch()
        # keep the structs, the workspace and the condition objects' own cached images; drop the caller's tensors and this call's outputs
        with _WS_LOCK:
            live_ws = list(_WS.values())
        plan.keep = [t for t in plan.keep if any(t is w for w in live_ws)] + [d.mask for d in dl if not isinstance(d.mask, torch.Tensor)]
        plan.result = None
        plan.fresh = (want_out and out is None, want_sums, loss_scale is not None)
        with _WS_LOCK:
            _CALL_CACHE[key] = plan
            while len(_CALL_CACHE) > _CALL_CACHE_MAX:
                _CALL_CACHE.popitem(last=False)
        return res
    _CALL_STATS["hit"] += 1
    a = ent.args
    new_out, has_sums, has_loss = ent.fresh
    o = out
    sums = loss32 = None
    if new_out:
        o = torch.empty_like(u)
    if has_sums:
        sums = torch.empty(2, dtype=torch.float64, device=u.device)
    if has_loss:
        loss32 = torch.empty((), dtype=torch.float32, device=u.device)
    # the entry's argument struct is shared by every caller that hits this key: the output pointers are patched and the launch is issued
    # under the entry's lock, so two Python threads on the same (stream, buffers) cannot interleave them
    with ent.lock:
        if new_out:
            a.out = o.data_ptr()
        if has_sums:
            a.energy = sums.data_ptr()
            a.sumsq = a.energy + 8
        if has_loss:
            a.energy_f32 = loss32.data_ptr()
        rc = ent._fn(ent._mesh_ref, ent._args_ref, C.c_void_p(key[2]))
    if rc:
        _lib.check(rc, "dn_poisson_apply")
    return (o, sums, loss32) if has_loss else (o, sums)


def call_cache_clear():
    """Forget the cached prepared calls (after dn_config_set of a launch-plan switch: the workspace size depends on the plan)."""
    with _WS_LOCK:
        _CALL_CACHE.clear()
        _FSDT_CACHE.clear()
        _PACK_CACHE.clear()
    _POISSON_WS_BYTES.clear()
    _FSDT_WS_BYTES.clear()


class PoissonPlan:
    """A dn_poisson_apply call prepared once for FIXED buffers: validation, the argument structs, the outputs and the workspace are
    set up at construction; `launch()` is one ctypes call (~3 us of host time instead of ~25).  For loops that evaluate the operator
    on the same tensors every step -- the slab-parallel path (diffnet_amd/slab.py), whose per-rank kernels are shorter than the
    host-side preparation -- and for anything captured into a HIP graph.  Outputs are overwritten by every launch."""

    def __init__(self, geom, u, nu=None, f=None, f_gp=None, dirichlet=(), alpha=1.0, beta=1.0, c=1.0, wscale=1.0,
                 out_scale=1.0, want_out=True, want_sums=True, loss_scale=None, out=None, strict=True, strip_select=0, continues=None,
                 async_sums=False, loss_out=None, pipelined_sums=False):
        """pipelined_sums (round 4): the launch leaves per-workgroup partial sums in a workspace of its own and does NOT form its scalars; the
        NEXT launch in the loop does, at its start (`nxt.fold(this)`: dn_poisson_args.fold_prev -- the first workgroup of that launch adds the
        partials up before its own march), and `finish_sums()` closes the last evaluation of a loop with the one-workgroup kernel.  The
        final reduction (a ~3 us serial tail of a ~56 us launch: every workgroup has finished, the last arriver adds 2048 partials) leaves the
        critical path without a side stream or an event; the price: the loss of step k is ready when launch k + 1 has run (a training loop
        only logs it).  The gradient is final when its own launch ends, as always.
        async_sums: the launch leaves per-workgroup partial sums; the final scalars are formed by a one-workgroup kernel on a SIDE stream
        (dn_poisson_finish_sums), i.e. under whatever the launch stream runs next -- the in-kernel final reduction is a ~3 us serial tail
        of every launch.  The gradient is ready in launch-stream order as always; the sums / the loss are ready on the side stream: call
        `wait_sums()` (makes the current stream wait for them) before consuming them there, or consume them under `sums_stream`.  Such a
        plan owns its reduction workspace.
        strip_select 1 / 2: the launch covers only the first + last strip of the marched axis / all the others (include/diffnet_hip.h:
        split evaluation); `continues`: the PoissonPlan of the first launch of such a pair -- this one writes into its outputs and adds
        its sums to that launch's."""
        # strict: every tensor argument must be usable as it is (contiguous float32 fields, float32 / uint8 mask images) -- a conversion
        # would be a one-time copy that later launches keep reading after the caller has updated the original in place
        if continues is not None:
            out = continues.result[0]
        self.mesh, self.args, self.keep, self.result = _prepare_poisson(geom, u, nu, f, f_gp, dirichlet, alpha, beta, c, wscale, out_scale,
                                                                        want_out, want_sums, loss_scale, out, strict=strict,
                                                                        reuse=None if continues is None else continues.result)
        if loss_out is not None:
            # the float32 loss goes into the caller's one-element tensor (e.g. a slot of a buffer that one collective reduces for several steps)
            if loss_scale is None or continues is not None:
                raise ValueError("loss_out needs loss_scale and is not for the second launch of a split evaluation")
            if not (isinstance(loss_out, torch.Tensor) and loss_out.is_cuda and loss_out.dtype == torch.float32 and loss_out.numel() == 1
                    and loss_out.device == u.device):
                raise DiffNetHipError("loss_out must be a one-element float32 tensor on the device of u")
            self.args.energy_f32 = loss_out.data_ptr()
            self.keep.append(loss_out)
            self.result = (self.result[0], self.result[1], loss_out)
        self.args.strip_select = int(strip_select)
        self.args.accumulate_sums = int(continues is not None)
        self.async_sums = bool(async_sums) and self.args.workspace is not None and self.args.workspace != 0
        self.pipelined_sums = bool(pipelined_sums) and not self.async_sums and self.args.workspace is not None and self.args.workspace != 0
        self._folds = None
        if self.pipelined_sums:
            self.args.defer_sums = 1
            ws = torch.zeros(int(self.args.workspace_bytes), dtype=torch.uint8, device=u.device)      # its own: the partials wait for the next launch
            self.keep.append(ws)
            self.args.workspace = ws.data_ptr()
        if self.async_sums:
            self.args.defer_sums = 1
            ws = torch.zeros(int(self.args.workspace_bytes), dtype=torch.uint8, device=u.device)      # its own: the partials wait for the side stream
            self.keep.append(ws)
            self._own_ws = ws
            self.args.workspace = ws.data_ptr()
            self.sums_stream = _side_stream(u.device)
            self._side = self.sums_stream.cuda_stream
            self._ev_main, self._ev_done = _lib.new_event(), _lib.new_event()
            self._hip = _lib.hip_runtime()
            self._finish = _lib.lib().dn_poisson_finish_sums
            self._launched = False
        self.device = u.device
        self.stream = _raw_stream(u.device)      # the reduction workspace belongs to this stream
        self._fn = _lib.lib().dn_poisson_apply
        self._mesh_ref, self._args_ref = C.byref(self.mesh), C.byref(self.args)
        self.lock = _threading.Lock()            # guards the shared argument struct of a cached call (ops.poisson_apply)
        if self.async_sums:
            # the partials, the sums and the loss are allocated on the launch stream but read / written by the finish kernel on the side
            # stream: tell the caching allocator, or a dropped plan's blocks could be handed to launch-stream work while that kernel is pending
            for t in [self._own_ws] + [r for r in self.result[1:] if isinstance(r, torch.Tensor)]:
                t.record_stream(self.sums_stream)

    def __del__(self):
        if getattr(self, "async_sums", False):
            try:
                hip = self._hip
                if getattr(self, "_launched", False):
                    hip.hipEventSynchronize(self._ev_done)       # the finish kernel still reads the plan's workspace
                hip.hipEventDestroy(self._ev_main)
                hip.hipEventDestroy(self._ev_done)
            except Exception:                                     # interpreter shutdown: the runtime may already be gone
                pass

    def check(self):
        """Health check of the plan's workspace (synchronises the stream): raises if a launch that used it ran a bounded hand-over poll
        to its limit (its results are NaN then; include/diffnet_hip.h: dn_workspace_status)."""
        if self.args.workspace:
            _lib.check(_lib.lib().dn_workspace_status(C.c_void_p(self.args.workspace), C.c_void_p(_raw_stream(self.device))), "dn_workspace_status")

    def launch(self):
        cur = _raw_stream(self.device)
        if self.args.workspace and cur != self.stream:
            raise DiffNetHipError("PoissonPlan.launch: prepared on another stream (its reduction workspace is per stream); prepare one plan per stream")
        if self.async_sums:
            hip, side = self._hip, C.c_void_p(self._side)
            if self._launched:
                hip.hipStreamWaitEvent(C.c_void_p(cur), self._ev_done, 0)      # the previous evaluation's partials and scalars have been consumed
            rc = self._fn(self._mesh_ref, self._args_ref, C.c_void_p(cur))
            if rc:
                _lib.check(rc, "dn_poisson_apply")
            hip.hipEventRecord(self._ev_main, C.c_void_p(cur))
            hip.hipStreamWaitEvent(side, self._ev_main, 0)
            rc = self._finish(self._mesh_ref, self._args_ref, side)
            if rc:
                _lib.check(rc, "dn_poisson_finish_sums")
            hip.hipEventRecord(self._ev_done, side)
            self._launched = True
            return self.result
        rc = self._fn(self._mesh_ref, self._args_ref, C.c_void_p(cur))
        if rc:
            _lib.check(rc, "dn_poisson_apply")
        return self.result

    def fold(self, prev):
        """This launch closes the evaluation of `prev` (a pipelined_sums plan on the same mesh whose launch precedes this one on the stream):
        its first workgroup adds up prev's partial sums and writes prev's scalars.  fold(None) clears."""
        if prev is None:
            self.args.fold_prev = None
            self._folds = None
            return self
        if not getattr(prev, "pipelined_sums", False):
            raise DiffNetHipError("PoissonPlan.fold: the other plan must be prepared with pipelined_sums=True")
        self._folds = prev                                    # keeps its argument struct and workspace alive
        self.args.fold_prev = C.addressof(prev.args)
        return self

    def finish_sums(self):
        """pipelined_sums: form this plan's scalars from the partial sums of its last launch now (one small kernel on the current stream)
        -- for the last evaluation of a loop, which no further launch folds."""
        if not self.pipelined_sums:
            return self.result
        rc = _lib.lib().dn_poisson_finish_sums(self._mesh_ref, self._args_ref, C.c_void_p(_raw_stream(self.device)))
        if rc:
            _lib.check(rc, "dn_poisson_finish_sums")
        return self.result

    def wait_sums(self):
        """async_sums: make the current stream wait for the sums / the loss of the last launch (no-op otherwise)."""
        if self.async_sums and self._launched:
            self._hip.hipStreamWaitEvent(C.c_void_p(_raw_stream(self.device)), self._ev_done, 0)


def _prepare_poisson(geom, u, nu, f, f_gp, dirichlet, alpha, beta, c, wscale, out_scale, want_out, want_sums, loss_scale, out, strict=False, reuse=None):
    """Validation + argument structs of one dn_poisson_apply call: (mesh, args, tensors to keep alive, result tuple)."""
    nsd = geom.nsd
    u = _require(u, "u", nsd + 2, strict)
    B = u.shape[0]
    node_shape = tuple(geom.node_shape)            # (ny, nx) or (nz, ny, nx)
    if tuple(u.shape[1:]) != (1, *node_shape):
        raise ValueError(f"u has shape {tuple(u.shape)}, expected (B,1,{','.join(map(str, node_shape))})")
    keep = [u]
    args = DnPoissonArgs()
    args.u = u.data_ptr()

    def field(t, name, allow_gp=False):
        t = _require(t, name, nsd + 2, strict)
        if t.shape[0] not in (1, B):
            raise ValueError(f"{name} batch {t.shape[0]} does not broadcast to {B}")
        keep.append(t)
        return t, int(t.shape[0] == B)

    if nu is not None:
        nu, nb_ = field(nu, "nu")
        if tuple(nu.shape[1:]) != (1, *node_shape):
            raise ValueError("nu must be a nodal field")
        args.nu, args.nu_batched = nu.data_ptr(), nb_
    if f is not None and f_gp is not None:
        raise ValueError("give either nodal f or f_gp, not both")
    if isinstance(f, LoadVector):
        args.f_is_load = 1
        f = f.tensor
    if f is not None:
        f, fb = field(f, "f")
        if tuple(f.shape[1:]) != (1, *node_shape):
            raise ValueError("f must be a nodal field")
        args.f, args.f_batched = f.data_ptr(), fb
    if f_gp is not None:
        f_gp, fb = field(f_gp, "f_gp")
        if tuple(f_gp.shape[1:]) != (geom.ngp_total, *geom.elem_shape):
            raise ValueError(f"f_gp must have shape (B|1,{geom.ngp_total},{geom.elem_shape})")
        args.f_gp, args.f_batched = f_gp.data_ptr(), fb
    dl = _norm_dirichlet(dirichlet)
    # bit-packed / geometry-derived conditions: read directly by the 2-D Q1 kernels with nodal or absent forcing when every
    # condition of the call is one of them with a constant value; otherwise they are expanded to their (cached) uint8 images
    compact = (nsd == 2 and geom.deg == 1 and f_gp is None and len(dl) > 0 and _lib.lib().dn_config_get(b"Q1_RULE_KERNEL") in (None, b"") and
               all(isinstance(d.mask, (PackedMask, BoxFaces)) and not isinstance(d.value, torch.Tensor) for d in dl))
    # 3-D (round 4): BoxFaces are taken as they are (no image, no load) by the two-element Q1 kernel -- exact 2-point rule, even nx, nodal /
    # absent forcing, 8-byte aligned fields, constant values, at most one mask image beside them; anything else gets the expanded image
    box3d = False
    if nsd == 3 and geom.deg == 1 and f_gp is None and any(isinstance(d.mask, BoxFaces) for d in dl):
        mesh0 = geom.mesh_struct(B)
        imgs = [d.mask for d in dl if not isinstance(d.mask, BoxFaces)]
        cfg = _lib.lib().dn_config_get
        box3d = (mesh0.ngp == 2 and mesh0.nx % 2 == 0 and len(imgs) <= 1 and cfg(b"Q1_3D_T16") in (None, b"") and cfg(b"Q1_3D_E1") in (None, b"") and
                 all(not isinstance(d.value, torch.Tensor) for d in dl) and
                 all(isinstance(t, torch.Tensor) and t.is_cuda and t.is_contiguous() and t.dtype in (torch.uint8, torch.float32) and
                     t.data_ptr() % 8 == 0 for t in imgs) and
                 all(t is None or t.data_ptr() % 8 == 0 for t in (u, nu, f, out)))
    for k, d in enumerate(dl):
        m = d.mask
        if box3d and isinstance(m, BoxFaces):
            bc = args.bc[k]
            bc.value = float(d.value)
            bc.mask_kind, bc.box_faces = _lib.MASK_BOX, m.bits
            continue
        if compact:
            bc = args.bc[k]
            bc.value = float(d.value)
            if isinstance(m, BoxFaces):
                bc.mask_kind, bc.box_faces = _lib.MASK_BOX, m.bits
            else:
                if tuple(m.shape[1:]) != (1, *node_shape) or m.shape[0] not in (1, B) or m.bits.device != u.device:
                    raise ValueError(f"packed Dirichlet mask of shape {m.shape} does not match u")
                keep.append(m.bits)
                bc.mask, bc.mask_kind, bc.row_words = m.bits.data_ptr(), _lib.MASK_BITS, m.row_words
                bc.mask_batched = int(m.shape[0] == B)
            continue
        m = _mask_image(m, u)
        if not isinstance(m, torch.Tensor) or not m.is_cuda:
            raise DiffNetHipError("Dirichlet mask must be a CUDA tensor")
        if strict and (m.dtype == torch.bool or not m.is_contiguous()):
            raise DiffNetHipError("PoissonPlan: a bool or non-contiguous Dirichlet mask would be converted into a one-time copy; pass a "
                                  "contiguous float32 / uint8 image (or a PackedMask / BoxFaces)")
        if m.dtype == torch.bool:
            m = m.to(torch.uint8)
        if m.dtype not in (torch.float32, torch.uint8):
            raise TypeError("Dirichlet mask must be float32, uint8 or bool")
        m = m.contiguous()
        if m.dim() != nsd + 2 or tuple(m.shape[1:]) != (1, *node_shape) or m.shape[0] not in (1, B):
            raise ValueError(f"Dirichlet mask shape {tuple(m.shape)} does not match u")
        keep.append(m)
        bc = args.bc[k]
        bc.mask = m.data_ptr()
        bc.mask_kind = _lib.MASK_U8 if m.dtype == torch.uint8 else _lib.MASK_F32
        bc.mask_batched = int(m.shape[0] == B)
        if isinstance(d.value, torch.Tensor):
            v = d.value
            if v.dim() == nsd:
                v = v[(None,) * 2]
            if strict and not v.is_cuda:
                raise DiffNetHipError("PoissonPlan: the Dirichlet value field must live on the GPU (a prepared launch would keep a one-time copy)")
            v = _require(v.to(u.device) if not v.is_cuda else v, "Dirichlet value", nsd + 2, strict)
            if tuple(v.shape[1:]) != (1, *node_shape) or v.shape[0] not in (1, B):
                raise ValueError("Dirichlet value field shape does not match u")
            keep.append(v)
            bc.field = v.data_ptr()
            bc.field_batched = int(v.shape[0] == B)
        else:
            bc.value = float(d.value)
    args.alpha, args.beta, args.c, args.wscale, args.out_scale = alpha, beta, c, wscale, out_scale
    mesh = geom.mesh_struct(B)
    if out is not None:
        if not want_out or out.shape != u.shape or out.dtype != torch.float32 or out.device != u.device or not out.is_contiguous():
            raise ValueError("out= must be a contiguous float32 tensor of u's shape on u's device")
    elif want_out:
        out = torch.empty_like(u)
    sums = None
    if out is not None:
        args.out = out.data_ptr()
    if want_sums or (mesh.nsd == 3 and mesh.degree > 1):       # (3-D Q2 / Q3: the workspace also holds the element vectors)
        key = (mesh.nsd, mesh.degree, mesh.ngp, mesh.nx, mesh.ny, mesh.nz, B)
        nbytes = _POISSON_WS_BYTES.get(key)
        if nbytes is None:
            nbytes = _lib.lib().dn_poisson_workspace_bytes(C.byref(mesh))
            if nbytes < 0:
                _lib.check(int(nbytes), "dn_poisson_workspace_bytes")
            _POISSON_WS_BYTES[key] = nbytes
        ws = _workspace(u.device, nbytes)
        keep.append(ws)
        args.workspace, args.workspace_bytes = ws.data_ptr(), ws.numel()
    if want_sums:
        sums = reuse[1] if reuse is not None else torch.empty(2, dtype=torch.float64, device=u.device)
        args.energy = sums.data_ptr()
        args.sumsq = sums.data_ptr() + 8
    loss32 = None
    if loss_scale is not None:
        if not want_sums:
            raise ValueError("loss_scale needs want_sums=True")
        loss32 = reuse[2] if reuse is not None else torch.empty((), dtype=torch.float32, device=u.device)
        args.energy_f32, args.energy_scale = loss32.data_ptr(), float(loss_scale)
    keep += [t for t in (out, sums, loss32) if t is not None]
    return mesh, args, keep, ((out, sums, loss32) if loss_scale is not None else (out, sums))


# ---- composed (operator-level) forms: differentiable wrt EVERY tensor input ------------------------------------
# The fused kernels return the gradient wrt u only.  When a coefficient (nu, f, f_gp) or a Dirichlet value field requires a
# gradient -- inverse / parametric-coefficient problems, which the reference differentiates through plain autograd
# (IBN_2D.py:116-134) -- the same loss is evaluated on the single-launch HIP operators (`_GaussPtEval` and its adjoint) plus
# torch elementwise ops, so autograd covers all of them.  Slower (Gauss-point tensors are materialised), never silently wrong.
def _wants_grad(*ts):
    return torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in ts)


def _extra_grad_inputs(nu, f, f_gp, dirichlet):
    return _wants_grad(nu, f, f_gp, *[d.value for d in dirichlet])


def _dirichlet_cond(m):
    return (m > 0.5) if m.is_floating_point() else (m != 0)


def _apply_dirichlet(u, dirichlet):
    for d in dirichlet:
        v = d.value
        if isinstance(v, torch.Tensor):
            v = v[(None,) * (u.dim() - v.dim())] if v.dim() < u.dim() else v
            v = v.to(u.device).expand_as(u)
        else:
            v = torch.full_like(u, float(v))
        u = torch.where(_dirichlet_cond(_mask_image(d.mask, u)), v, u)
    return u


def _geom_tables(geom, device):
    """Stacked (G, nbf^nsd) N / dN_d tables and the nd weights of a FemGeometry on `device` (cached)."""
    cache = geom.__dict__.setdefault("_dev_tables", {})
    hit = cache.get(str(device))
    if hit is None:
        from .tables import nd_tables
        K, _, w = nd_tables(geom.nsd, geom.deg, geom.gpx_1d, geom.gpw_1d, geom.hs)
        names = ["N_gp"] + ["dN_%s_gp" % a for a in "xyz"[:geom.nsd]]
        hit = ({n: torch.from_numpy(K[n].reshape(K[n].shape[0], -1).copy()).to(device) for n in names}, torch.from_numpy(w).to(device))
        cache[str(device)] = hit
    return hit


def composed_energy(geom, u, nu=None, f=None, f_gp=None, dirichlet=(), c=1.0, jac=1.0):
    """Energy loss on the drop-in operators (the reference formulation, e.g. IBN_2D.py:116-134): differentiable wrt u, nu, f,
    f_gp and Dirichlet value fields."""
    T, w = _geom_tables(geom, u.device)
    nsd, nbf = geom.nsd, geom.deg + 1
    ev = lambda t, name: _GaussPtEval.apply(t, T[name], nsd, nbf, nbf - 1)
    ub = _apply_dirichlet(u, _norm_dirichlet(dirichlet))
    g2 = sum(ev(ub, "dN_%s_gp" % a) ** 2 for a in "xyz"[:nsd])
    wg = (w * jac).reshape((1, -1) + (1,) * nsd)
    dens = c * g2 if nu is None else c * ev(nu, "N_gp") * g2
    if f is not None or f_gp is not None:
        dens = dens - ev(ub, "N_gp") * (f_gp if f_gp is not None else ev(f, "N_gp"))
    return torch.sum(wg * dens) / (u.shape[0] * geom.nelem_total)


def composed_residual(geom, u, nu=None, f=None, f_gp=None, dirichlet=(), jac=1.0):
    """Assembled weak-form residual (zero on Dirichlet nodes) from the operator adjoints: R = sum_d gpe^T(W nu d_d u ; dN_d)
    - gpe^T(W f ; N).  Differentiable wrt every tensor input."""
    T, w = _geom_tables(geom, u.device)
    nsd, nbf = geom.nsd, geom.deg + 1
    dirichlet = _norm_dirichlet(dirichlet)
    ev = lambda t, name: _GaussPtEval.apply(t, T[name], nsd, nbf, nbf - 1)
    evT = lambda t, name: _GaussPtEvalT.apply(t.contiguous(), T[name], tuple(u.shape), nsd, nbf, nbf - 1)
    ub = _apply_dirichlet(u, dirichlet)
    wg = (w * jac).reshape((1, -1) + (1,) * nsd)
    nug = None if nu is None else ev(nu, "N_gp")
    R = None
    for a in "xyz"[:nsd]:
        q = wg * ev(ub, "dN_%s_gp" % a)
        q = q if nug is None else q * nug
        t = evT(q.expand(u.shape[0], *q.shape[1:]), "dN_%s_gp" % a)
        R = t if R is None else R + t
    if f is not None or f_gp is not None:
        fg = f_gp if f_gp is not None else ev(f, "N_gp")
        R = R - evT((wg * fg).expand(u.shape[0], -1, *fg.shape[2:]), "N_gp")
    for d in dirichlet:
        R = torch.where(_dirichlet_cond(_mask_image(d.mask, R)), torch.zeros_like(R), R)
    return R


class _PoissonFn(torch.autograd.Function):
    """(out, sums, loss) of one dn_poisson_apply launch, differentiable wrt u -- the eager counterpart of the registered operator
    diffnet_mi::poisson_apply (torch_ops.py), whose dispatch costs ~100 us per call (204 -> ~60 us for an energy_loss + backward at
    64^2, tools/host_profile.py); same formulas as torch_ops._pa_backward, and differentiable again (its backward calls `_fused`)."""

    @staticmethod
    def forward(ctx, u, nu, f, f_gp, geom, dl, alpha, beta, c, wscale, out_scale, loss_scale):
        out, sums, loss = poisson_apply(geom, u, nu, f, f_gp, dl, alpha=alpha, beta=beta, c=c, wscale=wscale, out_scale=out_scale,
                                        want_out=True, want_sums=True, loss_scale=loss_scale)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(out, nu)
        ctx.meta = (geom, dl, alpha, beta, c, wscale, out_scale, loss_scale)
        return out, sums, loss

    @staticmethod
    def backward(ctx, g_out, g_sums, g_loss):
        out, nu = ctx.saved_tensors
        geom, dl, alpha, beta, c, wscale, out_scale, loss_scale = ctx.meta
        hom = tuple(Dirichlet(d.mask, 0.0) for d in dl)       # J = s alpha M K M is symmetric: J^T v is the same launch, homogeneous conditions, no forcing

        def K(v, scale):
            return _fused(geom, v, nu, None, None, hom, alpha, 0.0, 0.0, wscale, scale, 0.0)[0]

        gu = None
        if g_out is not None:
            gu = K(g_out.contiguous(), out_scale)
        coef = None
        if g_loss is not None:
            coef = g_loss * loss_scale
        if g_sums is not None:
            coef = g_sums[0].to(torch.float32) if coef is None else coef + g_sums[0].to(torch.float32)
            t = K(out * (2.0 / out_scale), 1.0) * g_sums[1].to(torch.float32)
            gu = t if gu is None else gu + t
        if coef is not None:
            if alpha != 2.0 * c or beta != 1.0:
                if g_loss is not None or c != 0.0:
                    raise RuntimeError("poisson_apply: the energy output is differentiable only in the energy-loss form (alpha = 2c, beta = 1)")
            else:
                t = out * (coef / out_scale)
                gu = t if gu is None else gu + t
        return (gu,) + (None,) * 11


def _fused(geom, u, nu, f, f_gp, dirichlet, alpha, beta, c, wscale, out_scale, loss_scale):
    """(out, sums, loss) of one dn_poisson_apply launch, differentiable wrt u through all three outputs.  Under torch.compile /
    torch.export the registered operator diffnet_mi::poisson_apply (an ordinary graph node); in eager mode the same launch behind a
    plain autograd.Function, or directly when nothing requires a gradient."""
    if torch.compiler.is_compiling():
        from . import torch_ops
        if isinstance(f, LoadVector):
            raise DiffNetHipError("LoadVector forcing is an eager-mode argument (the registered operator takes tensors); pass nodal f under torch.compile")
        compact = (geom.nsd == 2 and geom.deg == 1 and f_gp is None and not _lib.CONFIG_MIRROR.get("Q1_RULE_KERNEL") and
                   all(not isinstance(d.value, torch.Tensor) for d in dirichlet))
        return torch_ops.poisson_apply(u, nu, f, f_gp, *torch_ops.dirichlet_args(dirichlet, u, compact), *torch_ops.geometry_args(geom),
                                       float(alpha), float(beta), float(c), float(wscale), float(out_scale), float(loss_scale))
    for d in dirichlet:
        if isinstance(d.mask, torch.Tensor) and d.mask.dtype == torch.int32:
            raise TypeError("Dirichlet mask images are float32, uint8 or bool; int32 tensors are reserved for bit-packed masks (wrap them in ops.PackedMask)")
    if torch.is_grad_enabled() and isinstance(u, torch.Tensor) and u.requires_grad:
        return _PoissonFn.apply(u, nu, f, f_gp, geom, tuple(dirichlet), float(alpha), float(beta), float(c), float(wscale), float(out_scale),
                                float(loss_scale))
    return poisson_apply(geom, u, nu, f, f_gp, dirichlet, alpha=alpha, beta=beta, c=c, wscale=wscale, out_scale=out_scale, want_out=True,
                         want_sums=True, loss_scale=loss_scale)


def energy_loss(geom, u, nu=None, f=None, f_gp=None, dirichlet=(), c=1.0, jac=1.0):
    """mean_{b,e} sum_g gpw_g*jac*(c*nu_g*|grad u|_g^2 - u_g*f_g).  One fused launch, differentiable wrt u; when nu / f /
    f_gp / a Dirichlet value field requires a gradient the composed operator form runs instead (see composed_energy)."""
    dirichlet = tuple(_norm_dirichlet(dirichlet))
    if _extra_grad_inputs(nu, f, f_gp, dirichlet):
        return composed_energy(geom, u, nu, f, f_gp, dirichlet, float(c), float(jac))
    scale = 1.0 / (u.shape[0] * geom.nelem_total)
    return _fused(geom, u, nu, f, f_gp, dirichlet, 2.0 * c, 1.0, c, jac, scale, scale)[2]


def energy_loss_and_grad(geom, u, nu=None, f=None, f_gp=None, dirichlet=(), c=1.0, jac=1.0, out=None):
    """(loss, dloss/du) from ONE kernel pass, outside autograd -- the form an optimiser loop wants and the
    form bench.py measures (16 algorithmic bytes per node: read u, nu, f; write grad).  `out`: optional preallocated
    gradient buffer (same shape as u)."""
    B = u.shape[0]
    scale = 1.0 / (B * geom.nelem_total)
    grad, _, loss = poisson_apply(geom, u.detach(), nu, f, f_gp, dirichlet, alpha=2.0 * c, beta=1.0, c=c, wscale=jac,
                                  out_scale=scale, want_out=True, want_sums=True, loss_scale=scale, out=out)
    return loss, grad


def residual(geom, u, nu=None, f=None, f_gp=None, dirichlet=(), jac=1.0):
    dirichlet = tuple(_norm_dirichlet(dirichlet))
    if _extra_grad_inputs(nu, f, f_gp, dirichlet):
        return composed_residual(geom, u, nu, f, f_gp, dirichlet, float(jac))
    return _fused(geom, u, nu, f, f_gp, dirichlet, 1.0, 1.0, 0.0, jac, 1.0, 0.0)[0]


def residual_loss(geom, u, nu=None, f=None, f_gp=None, dirichlet=(), jac=1.0):
    dirichlet = tuple(_norm_dirichlet(dirichlet))
    if _extra_grad_inputs(nu, f, f_gp, dirichlet):
        return torch.sum(composed_residual(geom, u, nu, f, f_gp, dirichlet, float(jac)) ** 2)
    return _fused(geom, u, nu, f, f_gp, dirichlet, 1.0, 1.0, 0.0, jac, 1.0, 0.0)[1][1].to(torch.float32)


_FSDT_WS_BYTES = {}
_FSDT_CACHE = __import__("collections").OrderedDict()


def _fsdt_key(geom, flds, bc, bc_values, consts, in_scale, in_num, in_den, flags):
    """Key of a cached prepared dn_fsdt_apply call (see _call_key); None when an argument needs a conversion copy."""
    parts = [geom.key, flds[0].device.index, _raw_stream(flds[0].device), consts, flags]
    for t in flds:
        k = _tkey(t)
        if k is None or k == 0 or k[1] != torch.float32:
            return None
        parts.append(k)
    kb = _tkey(bc)
    if kb is None or (kb != 0 and kb[1] not in (torch.float32, torch.uint8)):
        return None
    parts.append(kb)
    for v in bc_values:
        if isinstance(v, torch.Tensor) and v.numel() > 1:
            kv = _tkey(v)
            if kv is None or kv[1] != torch.float32:
                return None
            parts.append(kv)
        else:
            parts.append(float(v))
    for t in (in_scale, in_num, in_den):
        k = _tkey(t)
        if k is None or (k != 0 and (k[1] != torch.float32 or k[2] != (3,))):
            return None
        parts.append(k)
    return tuple(parts)


class DeferredNorms:
    """Handle of a dn_fsdt_apply launch that left its partial sums of squares in the stream's reduction workspace (`defer_norms=True`): pass it as
    `norms_from` to the NEXT FSDT launch on that stream, which forms the norms itself.  Any other reducing FSDT launch in between overwrites the
    partials -- the pair is meant to be issued back to back (elasticity.fsdt_loss_and_grad, FsdtPlan)."""

    _next = [1]

    def __init__(self, ws, ticket=None):
        self.ws = ws
        if ticket is None:                     # a fresh ticket per deferring launch (1 .. 2^31 - 1)
            ticket = DeferredNorms._next[0]
            DeferredNorms._next[0] = ticket % 0x7FFFFFFE + 1
        self.ticket = ticket


def fsdt_apply(geom, w, phi_x, phi_y, bc=None, bc_values=(0.0, 0.0, 0.0), D11=1.0, D12=0.0, D22=1.0, D66=1.0, A44=1.0, A55=1.0,
               q=0.0, wscale=1.0, want_out=True, want_sums=True, in_scale=None, want_norms=False, in_num=None, in_den=None,
               defer_norms=False, norms_from=None):
    """One launch of dn_fsdt_apply (include/diffnet_hip.h): the three assembled FSDT plate residuals of the fields
    (B,1,ny,nx) and / or the float64 device tensor of their three sums of squares.  `bc`: Dirichlet node mask (fp32,
    `>= 0.5`, or bool/uint8), per sample or shared; `bc_values[k]`: float or tensor the k-th field / residual takes there;
    `in_scale`: optional float32 device tensor of 3 factors applied to the fields as they are loaded; `in_num` / `in_den`: the same
    with the factors in_num[k] / in_den[k] (0 where in_den[k] <= 0) formed by the kernel; `want_norms`: a third result, the float32
    tensor of the three Frobenius norms written by the same launch.  Returns (outs | None, sums | None[, norms]).
    `defer_norms` (round 4; instead of want_sums / want_norms): the launch leaves per-workgroup partials only -- no arrival protocol, no final
    reduction at its end (4.5-6.6 us of a 1025^2 launch) -- and the third result is a DeferredNorms handle; `norms_from=handle` (with in_num, without
    in_den) makes THIS launch form the norms from those partials, use them as in_den, and, with want_norms, return them as its third result.
    Calls on the same buffers reuse their prepared argument structs (small LRU, fresh outputs per call: see poisson_apply)."""
    if geom.nsd != 2:
        raise DiffNetHipError("fsdt_apply: 2-D meshes only")
    if defer_norms and (want_sums or want_norms or norms_from is not None):
        raise ValueError("fsdt_apply: defer_norms replaces want_sums / want_norms and does not combine with norms_from")
    if norms_from is not None and (in_num is None or in_den is not None or want_sums):
        raise ValueError("fsdt_apply: norms_from goes with in_num, without in_den and without want_sums")
    consts = tuple(float(x) for x in (D11, D12, D22, D66, A44, A55, q, wscale))
    flds = (w, phi_x, phi_y)
    key = None
    if all(isinstance(t, torch.Tensor) and t.is_cuda for t in flds) and tuple(w.shape[1:]) == (1, *geom.node_shape) and w.shape == phi_x.shape == phi_y.shape:
        key = _fsdt_key(geom, flds, bc, bc_values, consts, in_scale, in_num, in_den,
                        (want_out, want_sums, want_norms, bool(defer_norms), None if norms_from is None else norms_from.ws.data_ptr()))
    ent = None
    if key is not None:
        with _WS_LOCK:
            ent = _FSDT_CACHE.get(key)
            if ent is not None:
                _FSDT_CACHE.move_to_end(key)
    if ent is None:
        _CALL_STATS["miss" if key is not None else "uncached"] += 1
        mesh, args, keep, shape = _prepare_fsdt(geom, w, phi_x, phi_y, bc, bc_values, consts, in_scale, in_num, in_den,
                                                (want_sums or want_norms or defer_norms) and norms_from is None, defer_norms, norms_from)
        with _WS_LOCK:
            live_ws = list(_WS.values())
        ent = (mesh, args, C.byref(mesh), C.byref(args), shape, [t for t in keep if any(t is x for x in live_ws)],
               next((t for t in keep if any(t is x for x in live_ws)), None))
        if key is not None:
            with _WS_LOCK:
                _FSDT_CACHE[key] = ent
                while len(_FSDT_CACHE) > _CALL_CACHE_MAX:
                    _FSDT_CACHE.popitem(last=False)
    else:
        _CALL_STATS["hit"] += 1
    mesh, args, mref, aref, shape = ent[:5]
    dev = w.device
    outs = sums = norms = None
    if want_out:
        o3 = torch.empty((3, *shape), dtype=torch.float32, device=dev)      # one allocation, three views
        outs = list(o3.unbind(0))
    if want_sums:
        sums = torch.empty(3, dtype=torch.float64, device=dev)
    if want_norms:
        norms = torch.empty(3, dtype=torch.float32, device=dev)
    with _FSDT_LAUNCH_LOCK:          # pointer patch + launch of the (possibly shared, cached) argument struct as one step
        if want_out:
            p0, step = o3.data_ptr(), 4 * o3[0].numel()
            args.out[0], args.out[1], args.out[2] = p0, p0 + step, p0 + 2 * step
        if want_sums:
            args.sumsq = sums.data_ptr()
        if want_norms:
            args.norms = norms.data_ptr()
        if defer_norms:
            handle = DeferredNorms(ent[6])
            args.defer_sums = handle.ticket
        if norms_from is not None:
            args.den_ticket = norms_from.ticket
        rc = _lib.lib().dn_fsdt_apply(mref, aref, _stream(w))
    if rc:
        _lib.check(rc, "dn_fsdt_apply")
    if defer_norms:
        return outs, None, handle
    return (outs, sums, norms) if want_norms else (outs, sums)


def _prepare_fsdt(geom, w, phi_x, phi_y, bc, bc_values, consts, in_scale, in_num, in_den, want_red, defer=False, norms_from=None):
    """Validation + argument struct of a dn_fsdt_apply call, outputs left unset: (mesh, args, tensors to keep alive, field shape)."""
    flds = [_require(t, n, 4) for t, n in ((w, "w"), (phi_x, "phi_x"), (phi_y, "phi_y"))]
    B = flds[0].shape[0]
    shape = (B, 1, *geom.node_shape)
    for t in flds:
        if tuple(t.shape) != shape:
            raise ValueError(f"fsdt_apply: field shape {tuple(t.shape)} != {shape}")
    keep = list(flds)
    args = DnFsdtArgs()
    args.w, args.phi_x, args.phi_y = (t.data_ptr() for t in flds)

    def batched(t, name):
        if tuple(t.shape[-2:]) != tuple(geom.node_shape) or t.numel() not in (B * geom.nnode_total, geom.nnode_total):
            raise ValueError(f"fsdt_apply: {name} shape {tuple(t.shape)} does not match the mesh {shape}")
        return 1 if (t.numel() == B * geom.nnode_total and B > 1) else 0

    if bc is not None:
        if not bc.is_cuda:
            raise DiffNetHipError("fsdt_apply: bc mask must be on the GPU")
        if bc.dtype in (torch.bool, torch.uint8):
            m = bc.to(torch.uint8).contiguous()
            args.mask_is_u8 = 1
        else:
            m = _require(bc, "bc")
            args.mask_is_u8 = 0
        args.mask_batched = batched(m, "bc")
        args.bc_mask = m.data_ptr()
        keep.append(m)
        for k, v in enumerate(bc_values):
            if isinstance(v, torch.Tensor) and v.numel() > 1:
                v = _require(v, f"bc_values[{k}]")
                args.bc_field_batched[k] = batched(v, f"bc_values[{k}]")
                args.bc_field[k] = v.data_ptr()
                keep.append(v)
            else:
                args.bc_value[k] = float(v)
    args.D11, args.D12, args.D22, args.D66, args.A44, args.A55, args.q, args.wscale = consts
    mesh = geom.mesh_struct(B)
    for name, t in (("in_scale", in_scale), ("in_num", in_num), ("in_den", in_den)):
        if t is not None:
            t = _require(t, name, 1)
            if t.numel() != 3:
                raise ValueError(f"{name} must hold 3 floats")
            setattr(args, name, t.data_ptr())
            keep.append(t)
    if (in_num is None) != (in_den is None and norms_from is None) or (in_num is not None and in_scale is not None):
        raise ValueError("fsdt_apply: in_num and in_den (or norms_from) go together, and not with in_scale")
    if norms_from is not None:
        args.den_workspace = norms_from.ws.data_ptr()
        args.den_ticket = norms_from.ticket
        keep.append(norms_from.ws)
    args.defer_sums = 1 if defer else 0          # (the caller patches the pair's ticket in: fsdt_apply per call, FsdtPlan once)
    if want_red:
        key = (mesh.nx, mesh.ny, mesh.degree, mesh.ngp, B)
        nbytes = _FSDT_WS_BYTES.get(key)
        if nbytes is None:
            nbytes = _lib.lib().dn_fsdt_workspace_bytes(C.byref(mesh))
            if nbytes < 0:
                _lib.check(int(nbytes), "dn_fsdt_workspace_bytes")
            _FSDT_WS_BYTES[key] = nbytes
        ws = _workspace(flds[0].device, nbytes)
        keep.append(ws)
        args.workspace, args.workspace_bytes = ws.data_ptr(), ws.numel()
    return mesh, args, keep, shape


class FsdtPlan:
    """The FSDT plate loss prepared once for FIXED buffers (the FSDT counterpart of PoissonPlan): `launch()` is two ctypes calls --
    dn_fsdt_apply on the fields (the three assembled residuals and, in the same launch, their Frobenius norms) and dn_fsdt_apply on the
    residuals as the VJP of sum_k weights[k] * ||R_k|| (the kernel forms weights[k] / ||R_k|| itself, 0 where a norm is 0) -- and returns
    (norms (3,) float32, [dL/dw, dL/dphi_x, dL/dphi_y]).  No autograd graph, no torch op, ~6 us of host time; outputs are overwritten by
    every launch.  `want_grad=False` prepares the first launch only.  Reference: e1_plate_bending_fsdt.py:128-232 (residuals, loss) and the
    backward pass autograd builds for it."""

    def __init__(self, geom, w, phi_x, phi_y, bc=None, bc_values=(0.0, 0.0, 0.0), weights=(1.0, 1.0, 1.0), want_grad=True,
                 D11=1.0, D12=0.0, D22=1.0, D66=1.0, A44=1.0, A55=1.0, q=0.0, wscale=1.0):
        if geom.nsd != 2:
            raise DiffNetHipError("FsdtPlan: 2-D meshes only")
        for t, n in ((w, "w"), (phi_x, "phi_x"), (phi_y, "phi_y")):
            _require(t, n, 4, True)          # strict: the plan keeps reading THESE buffers
        dev = w.device
        consts = tuple(float(x) for x in (D11, D12, D22, D66, A44, A55, q, wscale))
        self.mesh, self.args, self.keep, shape = _prepare_fsdt(geom, w, phi_x, phi_y, bc, bc_values, consts, None, None, None, True)
        self.residuals = torch.empty((3, *shape), dtype=torch.float32, device=dev)
        self.norms = torch.empty(3, dtype=torch.float32, device=dev)
        step = 4 * self.residuals[0].numel()
        for k in range(3):
            self.args.out[k] = self.residuals.data_ptr() + k * step
        self.args.norms = self.norms.data_ptr()
        self.grads = None
        self._refs = [(C.byref(self.mesh), C.byref(self.args))]
        if want_grad:
            self.weights = torch.tensor([float(x) for x in weights], dtype=torch.float32, device=dev)
            vconsts = consts[:6] + (0.0, consts[7])          # J = M K M, K symmetric: the VJP is the operator itself on the masked cotangents, q = 0
            R = list(self.residuals.unbind(0))
            # the first launch defers its sums: the second forms the norms from its partials (dn_fsdt_args.defer_sums / den_workspace) and writes them
            pair = DeferredNorms(next(t for t in self.keep if t.data_ptr() == self.args.workspace))
            self.vmesh, self.vargs, vkeep, _ = _prepare_fsdt(geom, R[0], R[1], R[2], bc, (0.0, 0.0, 0.0), vconsts, None, self.weights, None, False,
                                                             False, pair)
            self.vargs.norms = self.norms.data_ptr()
            self.args.norms = None
            self.args.defer_sums = pair.ticket
            self.keep += vkeep
            self.grads = torch.empty((3, *shape), dtype=torch.float32, device=dev)
            for k in range(3):
                self.vargs.out[k] = self.grads.data_ptr() + k * step
            self._refs.append((C.byref(self.vmesh), C.byref(self.vargs)))
        self.device = dev
        self.stream = _raw_stream(dev)           # the reduction workspace belongs to this stream
        self._fn = _lib.lib().dn_fsdt_apply
        self.result = (self.norms, None if self.grads is None else list(self.grads.unbind(0)))

    def launch(self):
        cur = _raw_stream(self.device)
        if cur != self.stream:
            raise DiffNetHipError("FsdtPlan.launch: prepared on another stream (its reduction workspace is per stream); prepare one plan per stream")
        s = C.c_void_p(cur)
        for mref, aref in self._refs:
            rc = self._fn(mref, aref, s)
            if rc:
                _lib.check(rc, "dn_fsdt_apply")
        return self.result


def compute_winding_nodes(points, normals, area, q):
    """Drop-in for `compute_winding_nodes` of IBN/poisson-2d/parametric/IBN_2D.py:89-104 (same argument shapes:
    points / normals (B,1,Npts,2), area (B,1,Npts,1) -- unused by the reference too --, q = stack((xx, yy)) (2,Ny,Nx));
    returns (B,1,Nx,Ny).  One HIP launch instead of a Python loop over grid columns."""
    pts = _require(points.reshape(points.size(0), -1, 2), "points")
    nrm = _require(normals.reshape(normals.size(0), -1, 2), "normals")
    nodes = _require(q if q.is_cuda else q.to(pts.device), "nodes", 3)
    B, npts = pts.shape[0], pts.shape[1]
    ny, nx = nodes.shape[1], nodes.shape[2]
    out = torch.empty((B, 1, nx, ny), dtype=torch.float32, device=pts.device)
    rc = _lib.lib().dn_winding_nodes(_p(pts), _p(nrm), _p(nodes), _p(out), B, npts, ny, nx, _stream(pts))
    _lib.check(rc, "dn_winding_nodes")
    return out
