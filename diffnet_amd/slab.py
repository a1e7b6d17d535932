"""Slab domain decomposition of the structured mesh across the GPUs of one node (one process per GPU,
torch.distributed over RCCL/xGMI; `gloo` on CPU for tests).

The reference has no domain decomposition (its only parallelism is Lightning DDP over the batch,
IBN/poisson-3d/parametric/IBN_3D.py:193-195); BASELINE.json's north_star asks for it: the slowest axis
(z in 3-D, y in 2-D) is cut into `world` contiguous slabs of element layers.  Rank r owns element layers
[e0, e1) and therefore node layers [e0, e1] INCLUSIVE: interface node layers are replicated on the two
neighbouring ranks (verified there against the oracle: per-rank tables are identical
to the global module's because h is unchanged).

Exchange steps per loss evaluation -- nothing else crosses GPUs:
  * forward : one all-reduce(sum) of the local energy (8 bytes); the loss is sum / (B * nel_GLOBAL), not a
    mean of per-rank means (slabs may differ by one layer);
  * backward: the gradient on an interface layer is the sum of both neighbours' contributions: one
    point-to-point exchange of a single node layer per interior face (256 KiB at 256^3), added in a fixed
    order (lower rank's part first) so both replicas are bitwise identical.
"""
import torch
import torch.distributed as dist


def slab_ranges(nel_slow, world):
    """Element-layer range [e0, e1) per rank: the first (nel % world) ranks get one extra layer."""
    base, extra = divmod(nel_slow, world)
    out, e0 = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out.append((e0, e0 + n))
        e0 += n
    if any(b <= a for a, b in out):
        raise ValueError(f"{nel_slow} element layers cannot be split over {world} ranks")
    return out


class SlabDecomposition:
    """Bookkeeping for one rank.  `sizes_xyz` / `lengths_xyz`: global nodes / lengths in the reference's
    kwarg order (X, Y[, Z]); `degree` = fem_basis_deg."""

    def __init__(self, nsd, sizes_xyz, lengths_xyz, rank, world, degree=1):
        self.nsd, self.rank, self.world, self.degree = nsd, rank, world, degree
        self.sizes, self.lengths = tuple(sizes_xyz[:nsd]), tuple(lengths_xyz[:nsd])
        slow = nsd - 1                                    # index of the slowest axis in (x, y, z) order
        nel_slow = (self.sizes[slow] - 1) // degree
        self.ranges = slab_ranges(nel_slow, world)
        self.e0, self.e1 = self.ranges[rank]
        self.n0, self.n1 = self.e0 * degree, self.e1 * degree            # owned node layers [n0, n1] inclusive
        self.nel_global = 1
        for s in self.sizes:
            self.nel_global *= (s - 1) // degree
        h = self.lengths[slow] / nel_slow
        self.local_sizes = self.sizes[:slow] + ((self.e1 - self.e0) * degree + 1,)
        self.local_lengths = self.lengths[:slow] + (h * (self.e1 - self.e0),)

    def local_kwargs(self, **extra):
        """kwargs for the per-rank DiffNet2DFEM/DiffNet3DFEM (same h as the global mesh)."""
        pad = (1,) * (3 - self.nsd)
        kw = dict(nsd=self.nsd, domain_sizes=self.local_sizes + pad, domain_lengths=self.local_lengths + tuple(float(p) for p in pad),
                  domain_size=self.local_sizes[0], domain_length=self.local_lengths[0], fem_basis_deg=self.degree)
        kw.update(extra)
        return kw

    def take(self, t):
        """Slice a global nodal tensor (B,1,[Nz,]Ny,Nx) down to this rank's slab (with the replicated interface layers)."""
        return t[:, :, self.n0:self.n1 + 1].contiguous()

    def owned_mask(self, like):
        """1 on node layers this rank is the *primary* owner of (interface layers belong to the lower rank): use it to
        combine per-rank nodal reductions without double counting."""
        m = torch.ones(self.n1 - self.n0 + 1, dtype=like.dtype, device=like.device)
        if self.rank > 0:
            m[0] = 0
        return m.reshape((1, 1, -1) + (1,) * (self.nsd - 1))


def exchange_interfaces(grad_local, dec, group=None):
    """Sum the two partial gradients of every interface node layer in place (lower rank's part first on both
    sides => bitwise identical replicas).  One isend/irecv pair per interior face."""
    if dec.world == 1:
        return grad_local
    ops, recv_lo, recv_hi = [], None, None
    lo, hi = dec.rank - 1, dec.rank + 1
    if lo >= 0:
        send_lo = grad_local[:, :, 0].contiguous()
        recv_lo = torch.empty_like(send_lo)
        ops += [dist.P2POp(dist.isend, send_lo, lo, group), dist.P2POp(dist.irecv, recv_lo, lo, group)]
    if hi < dec.world:
        send_hi = grad_local[:, :, -1].contiguous()
        recv_hi = torch.empty_like(send_hi)
        ops += [dist.P2POp(dist.isend, send_hi, hi, group), dist.P2POp(dist.irecv, recv_hi, hi, group)]
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    if recv_lo is not None:
        grad_local[:, :, 0] = recv_lo + grad_local[:, :, 0]        # lower rank's contribution first
    if recv_hi is not None:
        grad_local[:, :, -1] = grad_local[:, :, -1] + recv_hi
    return grad_local


def slab_energy_loss_and_grad(dec, local_sum_and_grad, batch, group=None):
    """Global energy loss and this rank's slab of its gradient.

    `local_sum_and_grad()` must return (energy_sum, grad) of the LOCAL slab where energy_sum is the un-normalised
    sum over local elements (0-dim tensor) and grad = d(energy_sum)/du_local * 1/(B*nel_global).  On the GPU that is
    `ops.poisson_apply(..., out_scale=1/(B*nel_global))` (HIP); the CPU tests inject the oracle.
    """
    esum, grad = local_sum_and_grad()
    esum = esum.clone().reshape(1)
    if dec.world > 1:
        dist.all_reduce(esum, op=dist.ReduceOp.SUM, group=group)
        exchange_interfaces(grad, dec, group)
    return (esum / (batch * dec.nel_global)).reshape(()), grad


class SlabPoisson:
    """Slab-parallel fused Poisson energy on the GPU: the per-rank FEM module + the two exchange steps."""

    def __init__(self, nsd, sizes_xyz, lengths_xyz, rank, world, ngp_1d=2, group=None, device=None):
        from . import DiffNet2DFEM, DiffNet3DFEM
        self.dec = SlabDecomposition(nsd, sizes_xyz, lengths_xyz, rank, world)
        cls = DiffNet3DFEM if nsd == 3 else DiffNet2DFEM
        self.fem = cls(None, **self.dec.local_kwargs(ngp_1d=ngp_1d))
        if device is not None:
            self.fem = self.fem.to(device)
        self.group = group

    def energy_loss_and_grad(self, u_local, nu=None, f=None, dirichlet=(), c=1.0, jac=1.0):
        from . import ops
        B = u_local.shape[0]
        scale = 1.0 / (B * self.dec.nel_global)

        def local():
            grad, sums = ops.poisson_apply(self.fem.geom, u_local, nu, f, None, dirichlet, alpha=2.0 * c, beta=1.0, c=c,
                                           wscale=jac, out_scale=scale, want_out=True, want_sums=True)
            return sums[0], grad

        loss, grad = slab_energy_loss_and_grad(self.dec, local, B, self.group)
        return loss.to(torch.float32), grad
