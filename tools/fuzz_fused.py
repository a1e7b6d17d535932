#!/usr/bin/env python3
"""Randomised cross-check of the fused Poisson operator against the same loss composed from the single-launch operators behind autograd
(ops.composed_energy / composed_residual): random mesh sizes (around the tile / chunk / strip widths of the kernels), degrees, rules, batch sizes,
mask formats, one or two conditions, value fields, absent coefficients, launch-plan overrides.  usage: fuzz_fused.py [cases] [seed]"""
import os, random, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffnet_amd import BoxFaces, DiffNet2DFEM, DiffNet3DFEM, PackedMask, _lib, ops


def run(ncases=60, seed=0, verbose=True):
    """Returns the descriptions of the cases that disagree (or raised)."""
    dev = torch.device("cuda:0")
    rng = random.Random(seed)
    failures = []
    for case in range(ncases):
        _one(case, rng, dev, failures, verbose)
    return failures


def _one(case, rng, dev, failures, verbose):
    if True:
        nsd = rng.choice([2, 2, 3])
        deg = rng.choice([1, 1, 1, 2, 3])
        ngp = rng.choice([2, 3, 4]) if deg == 1 else rng.choice([3, 4])
        if nsd == 2:
            nel = [rng.choice([1, 2, 3, 7, 16, 31, 32, 33, 63, 64, 127, 128, 130, 255, 320, 384, 512]), rng.choice([1, 2, 3, 7, 16, 31, 32, 33, 63, 64, 127, 128, 130, 255])]
        else:
            nel = [rng.choice([1, 2, 3, 5, 14, 15, 16, 17, 29, 30, 31, 33, 47]) for _ in range(3)]
            if deg > 1:
                nel = [min(e, 6) for e in nel]
        sizes = tuple(e * deg + 1 for e in nel)
        lengths = tuple(rng.choice([1.0, 0.7, 1.3]) for _ in sizes)
        B = rng.choice([1, 1, 2, 3, 5])
        kw = dict(nsd=nsd, domain_sizes=sizes if nsd == 3 else sizes + (1,), domain_lengths=lengths if nsd == 3 else lengths + (1.0,), domain_size=sizes[0],
                  fem_basis_deg=deg, ngp_1d=ngp)
        m = (DiffNet3DFEM if nsd == 3 else DiffNet2DFEM)(None, **kw).to(dev)
        shape = (B, 1, *m.geom.node_shape)
        g = torch.Generator().manual_seed(case)
        u = torch.rand(shape, generator=g).to(dev)
        nu = (torch.rand(shape, generator=g) + 0.5).to(dev) if rng.random() < 0.8 else None
        fmode = rng.choice(["f", "f", "none", "fgp"])
        f = torch.rand(shape, generator=g).to(dev) if fmode == "f" else None
        fgp = torch.rand((B, m.geom.ngp_total, *m.geom.elem_shape), generator=g).to(dev) if fmode == "fgp" else None
        masks = []
        for k in range(rng.choice([0, 1, 1, 2])):
            mk = (torch.rand(shape if rng.random() < 0.7 else (1,) + shape[1:], generator=g) < 0.15)
            form = rng.choice(["f32", "u8", "bits", "bool"])
            val = rng.choice([0.0, 1.0, -0.3]) if rng.random() < 0.8 else torch.rand(shape, generator=g).to(dev)
            mt = {"f32": mk.float().to(dev), "u8": mk.to(torch.uint8).to(dev), "bool": mk.to(dev), "bits": mk.to(torch.uint8).to(dev)}[form]
            masks.append((form, mt, val))
        if nsd == 2 and rng.random() < 0.15:
            masks = [("box", BoxFaces(rng.choice(["all", ["xlo", "yhi"]])), 0.5)]
        def cond(pack):
            out = []
            for form, mt, val in masks:
                if form == "box":
                    out.append((mt if pack else mt.image(shape[2:], dev), val))
                elif form == "bits" and pack:
                    out.append((PackedMask.pack(mt), val))
                else:
                    out.append((mt, val))
            return out
        plan = ""
        if deg == 1 and rng.random() < 0.5:
            plan = (f"128,4,{rng.choice([1, 2, 3, 5, 8, 16])},{rng.choice([1, 4])}" if nsd == 2 and sizes[0] % 4 == 0 else
                    f"{rng.choice([64, 128, 256])},2,{rng.choice([1, 3, 8, 32])}") if nsd == 2 else f"16,16,{rng.choice([1, 2])},{rng.choice([1, 2, 5, 9])}"
        c, jac = rng.choice([1.0, 0.5]), rng.choice([1.0, 0.37])
        desc = f"case {case}: nsd={nsd} deg={deg} ngp={ngp} sizes={sizes} B={B} nu={'y' if nu is not None else 'n'} f={fmode} masks={[(a, 'field' if isinstance(v, torch.Tensor) else v) for a, _, v in masks]} plan={plan!r}"
        try:
            ur = u.clone().requires_grad_(True)
            ref = ops.composed_energy(m.geom, ur, nu, f, fgp, cond(False), c, jac)
            (gref,) = torch.autograd.grad(ref, ur)
            Rref = ops.composed_residual(m.geom, u, nu, f, fgp, cond(False), jac)
            _lib.config_set("PLAN2D" if nsd == 2 else "PLAN3D", plan)
            loss, grad = m.energy_loss_and_grad(u, nu, f, f_gp=fgp, dirichlet=cond(True), c=c, jac=jac)
            R = m.residual(u, nu, f, f_gp=fgp, dirichlet=cond(True), jac=jac)
            rl = m.residual_loss(u, nu, f, f_gp=fgp, dirichlet=cond(True), jac=jac)
            _lib.config_set("PLAN2D" if nsd == 2 else "PLAN3D", "")
            tol = 3e-5
            e1 = abs(float(loss) - float(ref.detach())) / (abs(float(ref.detach())) + 1e-6)
            e2 = float((grad - gref).abs().max()) / (float(gref.abs().max()) + 1e-12)
            e3 = float((R - Rref).abs().max()) / (float(Rref.abs().max()) + 1e-12)
            e4 = abs(float(rl) - float((Rref.double() ** 2).sum())) / (float((Rref.double() ** 2).sum()) + 1e-12)
            ok = e1 < tol and e2 < 3e-4 and e3 < 3e-4 and e4 < 1e-4
            if not ok:
                failures.append(desc + f"  loss {e1:.1e} grad {e2:.1e} R {e3:.1e} sumsq {e4:.1e}")
            if verbose:
                print(("ok   " if ok else "FAIL ") + desc + f"  loss {e1:.1e} grad {e2:.1e} R {e3:.1e} sumsq {e4:.1e}", flush=True)
        except Exception as ex:                        # noqa: BLE001
            _lib.config_set("PLAN2D" if nsd == 2 else "PLAN3D", "")
            failures.append("EXC  " + desc + f"  {type(ex).__name__}: {str(ex)[:200]}")
            print("EXC  " + desc + f"  {type(ex).__name__}: {str(ex)[:200]}", flush=True)


def run_fsdt(ncases=40, seed=0, verbose=True):
    """The fused FSDT plate residuals (dn_fsdt_apply: one launch, in-kernel sums and norms) against the same residuals composed from the
    single-launch operators (elasticity.fsdt_residuals_composed): random sizes, degrees, rules, masks, boundary values, launch plans."""
    from diffnet_amd.elasticity import _constants, fsdt_residuals_composed
    dev = torch.device("cuda:0")
    rng = random.Random(seed)
    failures = []
    for case in range(ncases):
        deg = rng.choice([1, 2, 2, 3])
        ngp = rng.choice([2, 3, 4]) if deg == 1 else rng.choice([3, 4])
        nel = [rng.choice([1, 2, 3, 5, 31, 63, 64, 65, 127, 190, 191, 192, 200]), rng.choice([1, 2, 3, 4, 7, 16, 33, 50])]
        sizes = tuple(e * deg + 1 for e in nel)
        B = rng.choice([1, 1, 2, 3])
        m = DiffNet2DFEM(None, nsd=2, domain_sizes=sizes + (1,), domain_lengths=(1.0, rng.choice([1.0, 0.6]), 1.0), domain_size=sizes[0], fem_basis_deg=deg,
                         ngp_1d=ngp).to(dev)
        shape = (B, 1, sizes[1], sizes[0])
        g = torch.Generator().manual_seed(1000 + case)
        flds = [torch.rand(shape, generator=g).to(dev) for _ in range(3)]
        mode = rng.choice(["f32", "u8", "f32"])
        bc = (torch.rand(shape if rng.random() < 0.6 else (1,) + shape[1:], generator=g) < 0.2).float().to(dev)
        bcm = bc if mode == "f32" else bc.to(torch.uint8)
        vals = [rng.choice([0.0, 0.1, -0.2]) for _ in range(3)]
        plan = rng.choice(["", "", "192,%d" % rng.choice([1, 2, 3, 7]), "64,%d,%d" % (rng.choice([1, 2, 3, 5]), rng.choice([2, 3, 4, 7, 12])), "128,%d" % rng.choice([2, 4])])
        kw = dict(E=rng.choice([1.0, 2.5]), v=0.25, h=rng.choice([0.1, 0.3]), K_s=rng.choice([1.0, 5.0 / 6.0]), q=rng.choice([1.0, 0.0, -2.0]))
        desc = f"fsdt case {case}: deg={deg} ngp={ngp} sizes={sizes} B={B} mask={mode} shared={bc.shape[0] == 1 and B > 1} values={vals} plan={plan!r}"
        try:
            ref = fsdt_residuals_composed(m, *flds, bc.expand(shape) if bc.shape[0] != B else bc, *vals, hx=m.hx, hy=m.hy, **kw)
            _lib.config_set("PLAN_FSDT", plan)
            got, sums, norms = ops.fsdt_apply(m.geom, *flds, bcm, tuple(vals), q=kw["q"], wscale=(0.5 * m.hx) * (0.5 * m.hy), want_norms=True,
                                              **_constants(kw["E"], kw["v"], kw["h"], kw["K_s"]))
            _lib.config_set("PLAN_FSDT", "")
            errs = [float((a - b).abs().max()) / (float(b.abs().max()) + 1e-12) for a, b in zip(got, ref)]
            serr = [abs(float(sums[k]) - float((ref[k].double() ** 2).sum())) / (float((ref[k].double() ** 2).sum()) + 1e-30) for k in range(3)]
            nerr = [abs(float(norms[k]) ** 2 - float(sums[k])) / (float(sums[k]) + 1e-30) for k in range(3)]
            ok = max(errs) < 3e-4 and max(serr) < 1e-4 and max(nerr) < 1e-5
            if not ok:
                failures.append(desc + f"  out {max(errs):.1e} sums {max(serr):.1e} norms {max(nerr):.1e}")
            if verbose:
                print(("ok   " if ok else "FAIL ") + desc + f"  out {max(errs):.1e} sums {max(serr):.1e} norms {max(nerr):.1e}", flush=True)
        except Exception as ex:                        # noqa: BLE001
            _lib.config_set("PLAN_FSDT", "")
            failures.append("EXC  " + desc + f"  {type(ex).__name__}: {str(ex)[:200]}")
            print("EXC  " + desc + f"  {type(ex).__name__}: {str(ex)[:200]}", flush=True)
    return failures


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    fails = run(n, seed)
    print(f"{n - len(fails)} of {n} cases agree")
    fails2 = run_fsdt(max(n // 2, 1), seed)
    print(f"{max(n // 2, 1) - len(fails2)} of {max(n // 2, 1)} FSDT cases agree")
    fails += fails2
    sys.exit(1 if fails else 0)
