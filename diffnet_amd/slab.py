"""Slab domain decomposition of the structured mesh across the GPUs of one node (one process per GPU,
torch.distributed over RCCL/xGMI; `gloo` on CPU for tests).

The reference has no domain decomposition (its only parallelism is Lightning DDP over the batch,
IBN/poisson-3d/parametric/IBN_3D.py:193-195); BASELINE.json's north_star asks for it: the slowest axis
(z in 3-D, y in 2-D) is cut into `world` contiguous slabs of element layers.  Rank r owns element layers
[e0, e1) and therefore node layers [e0, e1] INCLUSIVE: interface node layers are replicated on the two
neighbouring ranks (verified there against the oracle: per-rank tables are identical
to the global module's because h is unchanged).

Exchange steps per loss evaluation -- nothing else crosses GPUs:
  * forward : one all-reduce(sum) of this rank's share of the loss (4 bytes: the kernel writes energy_local / (B * nel_GLOBAL) as fp32),
    asynchronous; the loss is sum / (B * nel_GLOBAL), not a mean of per-rank means (slabs may differ by one layer);
  * backward: the gradient on an interface layer is the sum of both neighbours' contributions: one
    point-to-point exchange of a single node layer per interior face (256 KiB at 256^3), added in a fixed
    order (lower rank's part first) so both replicas are bitwise identical.  The slab is evaluated in two launches -- the
    strips next to the faces first, then the interior (SlabPoisson) -- so the transfers are in flight (side stream,
    pre-allocated buffers) while the interior is computed, and nothing is computed twice.
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


class InterfaceExchange:
    """The backward exchange step: one node layer per interior face, point to point, with pre-allocated send / receive
    buffers.  `start()` can be called as soon as this rank's partial gradients of its two interface layers exist -- before
    the interior of the slab has been computed -- and runs the transfers on a side stream (GPU tensors), so they overlap the
    slab kernel; `finish()` makes the current stream wait for them and writes lower-rank-part + upper-rank-part into the
    interface layers, in that order on both sides, so the two replicas of a layer are bitwise identical."""

    def __init__(self, dec, group=None):
        self.dec, self.group = dec, group
        self.buf, self.dev, self.host = None, None, False
        self.side = None
        self.reqs = []

    def _buffers(self, like):
        if self.buf is None or self.buf[0].shape != like.shape or self.dev != like.device or self.buf[0].dtype != like.dtype:
            # gloo has no GPU point-to-point: a gloo group on GPU tensors (the one-GPU rehearsal of the N > 1 logic) stages the
            # layers through host buffers; RCCL ("nccl") moves them GPU to GPU over xGMI
            self.host = like.is_cuda and dist.get_backend(self.group) == "gloo"
            self.dev = like.device
            self.buf = [torch.empty_like(like, device="cpu" if self.host else like.device) for _ in range(4)]   # send_lo, recv_lo, send_hi, recv_hi
            self.side = torch.cuda.Stream(like.device) if (like.is_cuda and not self.host) else None
        return self.buf

    def start(self, part_lo, part_hi):
        """part_lo / part_hi: this rank's contribution to its first / last node layer, shape (B,1,[Ny,]Nx), either may be None
        on the outermost ranks."""
        dec = self.dec
        ref = part_lo if part_lo is not None else part_hi
        if dec.world == 1 or ref is None:
            return
        send_lo, recv_lo, send_hi, recv_hi = self._buffers(ref)
        # a contiguous device-resident part is sent from where it is (no staging copy); `sent` is what finish() adds to the received part
        direct = lambda p: p.is_contiguous() and not self.host
        self.sent = [None, None]
        ops = []
        if dec.rank > 0:
            self.sent[0] = part_lo if direct(part_lo) else send_lo.copy_(part_lo)
            ops += [dist.P2POp(dist.isend, self.sent[0], dec.rank - 1, self.group), dist.P2POp(dist.irecv, recv_lo, dec.rank - 1, self.group)]
        if dec.rank + 1 < dec.world:
            self.sent[1] = part_hi if direct(part_hi) else send_hi.copy_(part_hi)
            ops += [dist.P2POp(dist.isend, self.sent[1], dec.rank + 1, self.group), dist.P2POp(dist.irecv, recv_hi, dec.rank + 1, self.group)]
        if self.side is not None:
            self.side.wait_stream(torch.cuda.current_stream(ref.device))
            with torch.cuda.stream(self.side):
                self.reqs = dist.batch_isend_irecv(ops)
        else:
            self.reqs = dist.batch_isend_irecv(ops)

    def finish(self, grad_local):
        """Wait for the transfers and write both interface layers of `grad_local` (B,1,Nslow_local,...) in place."""
        dec = self.dec
        if dec.world == 1:
            return grad_local
        for req in self.reqs:
            req.wait()
        if self.side is not None:
            torch.cuda.current_stream(grad_local.device).wait_stream(self.side)
        self.reqs = []
        _, recv_lo, _, recv_hi = (b.to(grad_local.device) for b in self.buf) if self.host else self.buf
        sent_lo, sent_hi = (None if t is None else t.to(grad_local.device) for t in self.sent)
        if dec.rank > 0:
            torch.add(recv_lo, sent_lo, out=grad_local[:, :, 0])            # lower rank's contribution first
        if dec.rank + 1 < dec.world:
            torch.add(sent_hi, recv_hi, out=grad_local[:, :, -1])
        return grad_local


def exchange_interfaces(grad_local, dec, group=None):
    """Sum the two partial gradients of every interface node layer in place (one-shot form of `InterfaceExchange`)."""
    if dec.world == 1:
        return grad_local
    ex = InterfaceExchange(dec, group)
    ex.start(grad_local[:, :, 0] if dec.rank > 0 else None, grad_local[:, :, -1] if dec.rank + 1 < dec.world else None)
    return ex.finish(grad_local)


def slab_energy_loss_and_grad(dec, local_sum_and_grad, batch, group=None, interface_parts=None, exchange=None):
    """Global energy loss and this rank's slab of its gradient.

    `local_sum_and_grad()` must return (energy_sum, grad) of the LOCAL slab where energy_sum is the un-normalised
    sum over local elements (0-dim tensor) and grad = d(energy_sum)/du_local * 1/(B*nel_global).  On the GPU that is
    `ops.poisson_apply(..., out_scale=1/(B*nel_global))` (HIP); the CPU tests inject the oracle.

    `interface_parts()` (optional) returns this rank's partial gradients (part_lo, part_hi) of its first / last node layer,
    computed from the ONE element layer next to each face -- cheap, and available before the slab kernel has run: the
    layer exchange is then started first and overlaps the slab computation and the loss all-reduce.  Without it the
    layers are taken from the finished slab gradient (no overlap).
    """
    ex = exchange if exchange is not None else InterfaceExchange(dec, group)
    started = False
    if dec.world > 1 and interface_parts is not None:
        ex.start(*interface_parts())
        started = True
    esum, grad = local_sum_and_grad()
    esum = esum.clone().reshape(1)
    if dec.world > 1:
        work = dist.all_reduce(esum, op=dist.ReduceOp.SUM, group=group, async_op=True)
        if not started:
            ex.start(grad[:, :, 0] if dec.rank > 0 else None, grad[:, :, -1] if dec.rank + 1 < dec.world else None)
        ex.finish(grad)
        work.wait()
    return (esum / (batch * dec.nel_global)).reshape(()), grad


class SlabPoisson:
    """Slab-parallel fused Poisson energy on the GPU: the per-rank FEM module + the two exchange steps.  Per evaluation TWO launches of
    the fused kernel on the slab, nothing computed twice: first the strips next to the two faces across the decomposed axis (C ABI:
    dn_poisson_args.strip_select = 1) -- after it this rank's parts of its interface layers are final, and their exchange with the
    neighbouring ranks starts on a side stream --, then all the other strips (strip_select = 2, sums added to the first launch's),
    which the exchange overlaps; the 4-byte loss all-reduce (asynchronous); one small add per face.

    overlap: True = always the two-launch form, False = one launch over the slab, then the exchange; "auto" (default, round 4) = the launch
    plan decides per call: a slab whose kernel is shorter than the collectives' latency is not worth splitting -- two launches over half the
    strips each take nearly as long as one over all of them when a launch is the latency chain of ~10 layers (rank 3 of 8 of 256^3, B = 1:
    18.1 + 18.3 us against 24.7 us in one launch, profiles/r3_slab_timeline.txt) -- so the split is used from SPLIT_MIN_NODES slab nodes x
    batch on (256^3 over 8 ranks: B >= 4)."""

    SPLIT_MIN_NODES = 8_000_000

    def __init__(self, nsd, sizes_xyz, lengths_xyz, rank, world, ngp_1d=2, group=None, device=None, overlap="auto"):
        from . import DiffNet2DFEM, DiffNet3DFEM
        self.dec = SlabDecomposition(nsd, sizes_xyz, lengths_xyz, rank, world)
        cls = DiffNet3DFEM if nsd == 3 else DiffNet2DFEM
        self.fem = cls(None, **self.dec.local_kwargs(ngp_1d=ngp_1d))
        if device is not None:
            self.fem = self.fem.to(device)
        self.group = group
        self.overlap = overlap if world > 1 else False
        self.exchange = InterfaceExchange(self.dec, group)

    def _local_conditions(self, dirichlet, like):
        """Dirichlet conditions in the form the slab launches take: tensor images of the LOCAL slab.  A PackedMask (given for the local
        slab, like every other argument) is unpacked once; BoxFaces name faces of the GLOBAL box: the faces across the decomposed axis
        exist on the outermost ranks only -- an interface between two ranks is not a boundary."""
        from . import ops
        out = []
        slow = "xyz"[self.dec.nsd - 1]
        for x in ops._norm_dirichlet(dirichlet):
            m = x.mask
            if isinstance(m, ops.PackedMask):
                m = m.image()
            elif isinstance(m, ops.BoxFaces):
                local = ops.BoxFaces([])
                local.bits = m.bits
                if self.dec.rank > 0:
                    local.bits &= ~ops.BoxFaces._BITS[slow + "lo"]
                if self.dec.rank + 1 < self.dec.world:
                    local.bits &= ~ops.BoxFaces._BITS[slow + "hi"]
                m = local.image(like.shape[2:], like.device)
            out.append(ops.Dirichlet(m, x.value))
        return out

    def _plans(self, u_local, nu, f, dirichlet, c, jac, scale):
        """The step's launches prepared once per set of buffers (ops.PoissonPlan): at 256^3 over 8 ranks a rank's kernels take ~30 us,
        the host-side preparation of the dn_poisson_apply calls more.  Re-prepared when a buffer, shape or coefficient changes.
        Returns (first launch, second launch | None, local conditions): with the overlapped exchange the first launch covers the strips
        next to the faces, the second the rest; otherwise one launch covers the slab."""
        from . import ops
        dl = ops._norm_dirichlet(dirichlet)
        for name, t in (("u_local", u_local), ("nu", nu), ("f", f)) + tuple(("Dirichlet mask", x.mask) for x in dl) + tuple(("Dirichlet value", x.value) for x in dl):
            if isinstance(t, torch.Tensor) and t.is_cuda and (not t.is_contiguous() or t.dtype == torch.bool):
                # the prepared launches keep pointers: a hidden .contiguous() / .to(uint8) copy would go stale when the caller updates the original
                raise ops.DiffNetHipError(f"SlabPoisson: {name} must be a contiguous float32 / uint8 tensor (SlabDecomposition.take() returns one)")
        key = (u_local.data_ptr(), tuple(u_local.shape), None if nu is None else nu.data_ptr(), None if f is None else f.data_ptr(),
               tuple((x.mask.data_ptr() if isinstance(x.mask, torch.Tensor) else id(x.mask),
                      x.value.data_ptr() if isinstance(x.value, torch.Tensor) else float(x.value)) for x in dl), float(c), float(jac))
        if getattr(self, "_plan_key", None) != key:
            dec = self.dec
            local = self._local_conditions(dl, u_local)
            kw = dict(alpha=2.0 * c, beta=1.0, c=c, wscale=jac, out_scale=scale, want_out=True, want_sums=True, loss_scale=scale)
            split = (u_local.numel() >= self.SPLIT_MIN_NODES) if self.overlap == "auto" else bool(self.overlap)
            first = ops.PoissonPlan(self.fem.geom, u_local, nu, f, None, local, strip_select=1 if split else 0, **kw)
            rest = ops.PoissonPlan(self.fem.geom, u_local, nu, f, None, local, strip_select=2, continues=first, **kw) if split else None
            self._plan_key, self._plan = key, (first, rest, local)
        return self._plan

    def energy_loss_and_grad(self, u_local, nu=None, f=None, dirichlet=(), c=1.0, jac=1.0):
        """(global loss, this rank's slab of its gradient).  The gradient tensor is owned by the prepared launch and overwritten by the
        next evaluation on the same buffers."""
        B = u_local.shape[0]
        dec = self.dec
        scale = 1.0 / (B * dec.nel_global)
        first, rest, _ = self._plans(u_local, nu, f, dirichlet, c, jac, scale)
        # the loss is taken from the launches themselves: the kernel writes energy_local / (B * nel_global) as float32, the all-reduce sums
        # those shares (no clone / divide / cast kernels per step)
        ex = self.exchange
        grad, _, loss = first.launch()
        if rest is not None:
            # this rank's parts of its two interface layers are final: start their exchange (side stream), then compute the interior
            ex.start(grad[:, :, 0] if dec.rank > 0 else None, grad[:, :, -1] if dec.rank + 1 < dec.world else None)
            rest.launch()
        if dec.world > 1:
            work = dist.all_reduce(loss, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            if rest is None:
                ex.start(grad[:, :, 0] if dec.rank > 0 else None, grad[:, :, -1] if dec.rank + 1 < dec.world else None)
            ex.finish(grad)
            work.wait()
        return loss, grad
