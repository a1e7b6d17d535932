// 2-D Q1 fused Poisson kernel, per-Gauss-point marching form.  Included by poisson2d_q1_g{2,3,4}.hip with
// DN_NGP defined: one translation unit per Gauss order so the variants build in parallel.  It serves forcing given at the
// Gauss points (f_gp), which the closed-form kernel (poisson2d_q1_cf.hip) cannot collapse, and is the A/B reference for
// that kernel (DN_Q1_RULE_KERNEL=1).
//
// grid = (chunks_x, strips_y, B), block = T threads.  A thread owns E consecutive elements of a row.  Carried across the
// march: the lower node row as raw nodal values and, per element, the cotangents of that row's x-stage values produced
// by the element layer below (CT, CDX).  Per layer and element: x-stage of both rows (1 sub + NGP FMAs per field and row),
// the O(NGP) layer arithmetic of q1_layer_2d (y sums collapsed onto the rule's moments), then ONE x-stage transpose per
// completed row, whose result is the finished nodal value.  Two rows per loop trip with the two raw rows / cotangent
// sets swapping roles, so no state is copied.
//
// What is present (nu, nodal f, f at Gauss points, Dirichlet conditions) is a compile-time flag set FL: run-time
// "is this pointer null" tests inside the march make the compiler unswitch/duplicate the loop and inflate the
// register allocation (measured: 226 vs 111 VGPRs), and every element is kept in its own basic block (the
// `if (valid)` below is also a scheduling fence: in one block the scheduler interleaves the E element streams and
// the live temporaries double).  Earlier forms of this kernel -- staged lower-row state with a state copy per row, an
// LDS-DMA two-slot ring -- were measured slower and removed (numbers in profiles/README.md; code in the history).
#include <cstdlib>

#include "poisson_common.h"

namespace dn {

enum : int { FL_NU = 1, FL_F = 2, FL_FGP = 4, FL_BC = 8, FL_BC_U8C = 16 };   // FL_BC_U8C: uint8 masks with constant values only

#ifndef DN_Q1_2D_WAVES
#define DN_Q1_2D_WAVES 2
#endif
#ifndef DN_PRIO_ROT
#define DN_PRIO_ROT 3      // 0 = off (A/B switch); measured -7 % kernel time at the bench shape
#endif

template <int E>
struct RawRow2D {
    float u[E + 1], n[E + 1], f[E + 1];
    float keep[E];
    BcRaw<E> bc;
    uint32_t m8[2][2];
};

template <int NGP, int E, bool VEC, int FL>
__global__ void __launch_bounds__(256, DN_Q1_2D_WAVES) poisson2d_q1_raw_kernel(const PoissonParams p) {
    constexpr int NW = E;
    constexpr bool HAS_NU = (FL & FL_NU) != 0, HAS_F = (FL & FL_F) != 0, FGP = (FL & FL_FGP) != 0;
    constexpr bool BC_ANY = (FL & (FL_BC | FL_BC_U8C)) != 0, BC_U8C = (FL & FL_BC_U8C) != 0;
    const int T = blockDim.x;
    const int tid = threadIdx.x;
    const int chunk = blockIdx.x, strip = selected_strip(p, (int)blockIdx.y), b = blockIdx.z;
    const int q = chunk * (T - 1) + tid;
    const int ex0 = q * E;
    const int x0 = ex0;
    const bool col_owner = !(chunk > 0 && tid == 0);
    const int64_t nps = (int64_t)p.nx * p.ny;
    const unsigned eps = (unsigned)(p.nelx * p.nely);
    const SampleBases sb = sample_bases(p, b, nps);
    const float* fgp = FGP ? p.fgp + (p.f_batched ? (int64_t)b * eps * (NGP * NGP) : 0) : nullptr;
    const int R = p.rows_per_strip;
    const int ey_own = strip * R;
    const int ey_begin = ey_own > 0 ? ey_own - 1 : 0;
    const int ey_end = min(ey_own + R, p.nely);

    __shared__ float xch[2][256];
    __shared__ double red[8];
    __shared__ int last_flag;

    float CTA[E][NGP], CDXA[E], CTB[E][NGP], CDXB[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        CDXA[e] = CDXB[e] = 0.f;
#pragma unroll
        for (int i = 0; i < NGP; ++i) CTA[e][i] = CTB[e][i] = 0.f;
    }

    auto row_issue = [&](int yr, RawRow2D<E>& r) {
        const unsigned rowoff = (unsigned)min(yr, p.ny - 1) * (unsigned)p.nx;
        load_seg<NW, VEC>(sb.u, rowoff, x0, p.nx, r.u);
        if constexpr (HAS_NU) load_seg<NW, VEC>(sb.nu, rowoff, x0, p.nx, r.n);
        if constexpr (HAS_F) load_seg<NW, VEC>(sb.f, rowoff, x0, p.nx, r.f);
        if constexpr (BC_U8C) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (sb.mask[k] != nullptr) {
                    uint8_t t[NW + 1];
                    load_seg<NW, VEC>(reinterpret_cast<const uint8_t*>(sb.mask[k]), rowoff, x0, p.nx, t);
                    uint32_t w = 0u;
#pragma unroll
                    for (int n = 0; n < NW; ++n) w |= (uint32_t)t[n] << (8 * n);
                    r.m8[k][0] = w;
                    r.m8[k][1] = t[NW];
                }
            }
        } else if constexpr (BC_ANY) {
            bc_issue<NW, VEC>(p, sb, rowoff, x0, r.bc);
        }
    };
    // u <- where(mask, value, u) on a landed row, keep[] = 0 on its Dirichlet nodes
    auto row_bc = [&](RawRow2D<E>& r) {
#pragma unroll
        for (int n = 0; n < NW; ++n) r.keep[n] = 1.f;
        if constexpr (BC_U8C) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (sb.mask[k] != nullptr) {
                    const float val = p.bc[k].value;
#pragma unroll
                    for (int n = 0; n <= NW; ++n) {
                        const bool set = n < NW ? ((r.m8[k][0] >> (8 * n)) & 0xffu) != 0u : r.m8[k][1] != 0u;
                        r.u[n] = set ? val : r.u[n];
                        if (n < NW) r.keep[n] = set ? 0.f : r.keep[n];
                    }
                }
            }
        } else if constexpr (BC_ANY) {
            bc_apply<NW>(p, sb, r.bc, r.u, r.keep);
        }
    };

    float e1_acc = 0.f, e2_acc = 0.f, sq_acc = 0.f;
    int par = 0;

    auto emit_row = [&](const float (&o)[NW + 1], const float (&keep)[NW], int yr, bool owned_row) {
        xch[par][tid] = o[NW];
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // LDS-only barrier (loads stay in flight)
        const float left = (tid > 0) ? xch[par][tid - 1] : 0.f;
        par ^= 1;
        if (owned_row && col_owner) {
            float v[NW];
#pragma unroll
            for (int n = 0; n < NW; ++n) {
                const float t = (o[n] + (n == 0 ? left : 0.f)) * keep[n];
                sq_acc = fmaf(t, t, sq_acc);
                v[n] = t * p.out_scale;
            }
            if (sb.out) store_seg<NW, VEC>(sb.out, (unsigned)yr * (unsigned)p.nx, x0, p.nx, v);
        }
    };

    // one element layer between the lower row L (Dirichlet applied) and the freshly landed upper row U
    auto layer = [&](int ey, const RawRow2D<E>& L, RawRow2D<E>& U, const float (&CTi)[E][NGP], const float (&CDXi)[E], float (&CTo)[E][NGP],
                     float (&CDXo)[E]) {
        const bool own_layer = ey >= ey_own;
        const float cnt = (own_layer && col_owner) ? 1.f : 0.f;
        row_bc(U);
        float o[NW + 1], le1 = 0.f, le2 = 0.f;
#pragma unroll
        for (int n = 0; n <= NW; ++n) o[n] = 0.f;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (ex0 + e < p.nelx) {       // elements beyond the domain are skipped (and: scheduling fence, see header)
                float TU0[NGP], TN0[NGP], TF0[NGP], TU1[NGP], TN1[NGP], TF1[NGP];
                const float DX0 = L.u[e + 1] - L.u[e], DX1 = U.u[e + 1] - U.u[e];
#pragma unroll
                for (int i = 0; i < NGP; ++i) {
                    TU0[i] = fmaf(p.T.b[i][1], DX0, L.u[e]);
                    TU1[i] = fmaf(p.T.b[i][1], DX1, U.u[e]);
                    TN0[i] = TN1[i] = 1.f;
                    TF0[i] = TF1[i] = 0.f;
                }
                if constexpr (HAS_NU) {
                    const float d0 = L.n[e + 1] - L.n[e], d1 = U.n[e + 1] - U.n[e];
#pragma unroll
                    for (int i = 0; i < NGP; ++i) { TN0[i] = fmaf(p.T.b[i][1], d0, L.n[e]); TN1[i] = fmaf(p.T.b[i][1], d1, U.n[e]); }
                }
                if constexpr (HAS_F) {
                    const float d0 = L.f[e + 1] - L.f[e], d1 = U.f[e + 1] - U.f[e];
#pragma unroll
                    for (int i = 0; i < NGP; ++i) { TF0[i] = fmaf(p.T.b[i][1], d0, L.f[e]); TF1[i] = fmaf(p.T.b[i][1], d1, U.f[e]); }
                }
                float fg[NGP * NGP];
                if constexpr (FGP) {
                    const unsigned eo = (unsigned)ey * (unsigned)p.nelx + (unsigned)(ex0 + e);
#pragma unroll
                    for (int gi = 0; gi < NGP * NGP; ++gi) fg[gi] = fgp[eo + (unsigned)gi * eps];
                }
                float ct0[NGP], cdx0, e1, e2;
                q1_layer_2d<NGP, FGP>(p.T, TU0, TU1, DX0, DX1, TN0, TN1, TF0, TF1, fg, ct0, CTo[e], cdx0, CDXo[e], e1, e2);
                le1 += e1;
                le2 += e2;
                float ssum = 0.f, bsum = cdx0 + CDXi[e];
#pragma unroll
                for (int i = 0; i < NGP; ++i) {
                    const float t = ct0[i] + CTi[e][i];
                    ssum += t;
                    bsum = fmaf(p.T.b[i][1], t, bsum);
                }
                o[e + 1] += bsum;
                o[e] += ssum - bsum;
            } else {
                // define the outgoing cotangents on this path too, so the two sets never have to be live together
                CDXo[e] = 0.f;
#pragma unroll
                for (int i = 0; i < NGP; ++i) CTo[e][i] = 0.f;
            }
        }
        e1_acc = fmaf(cnt, le1, e1_acc);
        e2_acc = fmaf(cnt, le2, e2_acc);
        emit_row(o, L.keep, ey, own_layer);
    };

    auto set_prio = [&](int e) {
#if DN_PRIO_ROT
        switch (((ey_end - e) >> 1) & 3) {          // progress-dependent wave priority (see the staged form)
            case 0: __builtin_amdgcn_s_setprio(0); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            default: __builtin_amdgcn_s_setprio(3); break;
        }
#endif
    };

    RawRow2D<E> RA, RB;
    row_issue(ey_begin, RA);
    row_bc(RA);
    int ey = ey_begin;
    for (; ey + 1 < ey_end; ey += 2) {
        set_prio(ey);
        row_issue(ey + 1, RB);
        layer(ey, RA, RB, CTA, CDXA, CTB, CDXB);
        row_issue(ey + 2, RA);
        layer(ey + 1, RB, RA, CTB, CDXB, CTA, CDXA);
    }
    bool odd = false;
    if (ey < ey_end) {
        set_prio(ey);
        row_issue(ey + 1, RB);
        layer(ey, RA, RB, CTA, CDXA, CTB, CDXB);
        odd = true;
    }
    if (ey_end == p.nely) {       // the last strip owns the top boundary row of the domain: only the layer below contributes
        float o[NW + 1];
#pragma unroll
        for (int n = 0; n <= NW; ++n) o[n] = 0.f;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (ex0 + e < p.nelx) {
                float ssum = 0.f, bsum = odd ? CDXB[e] : CDXA[e];
#pragma unroll
                for (int i = 0; i < NGP; ++i) {
                    const float c = odd ? CTB[e][i] : CTA[e][i];
                    ssum += c;
                    bsum = fmaf(p.T.b[i][1], c, bsum);
                }
                o[e + 1] += bsum;
                o[e] += ssum - bsum;
            }
        }
        float keep[NW];
#pragma unroll
        for (int n = 0; n < NW; ++n) keep[n] = odd ? RB.keep[n] : RA.keep[n];
        emit_row(o, keep, p.ny - 1, true);
    }

    if (p.want_sums) finish_sums(p, e1_acc, e2_acc, sq_acc, tid, T, red, &last_flag);
}

// ---- dispatch over the compile-time flag set --------------------------------------------------------------
template <int NGP, int E, bool VEC, int FL>
static void launch_one(const PoissonParams& pp, const Geom2D& g, int batch, hipStream_t s) {
    hipLaunchKernelGGL((poisson2d_q1_raw_kernel<NGP, E, VEC, FL>), dim3(g.chunks, g.strips, batch), dim3(g.T), 0, s, pp);
}

template <int NGP, int E, bool VEC, int FLF>   // FLF: nu / f flags already fixed
static void launch_bc(const PoissonParams& pp, const Geom2D& g, int batch, hipStream_t s) {
    const bool any = pp.bc[0].mask || pp.bc[1].mask;
    bool u8c = any;
    for (int k = 0; k < 2; ++k)
        if (pp.bc[k].mask && (!pp.bc[k].mask_is_u8 || pp.bc[k].field)) u8c = false;
    if (!any) launch_one<NGP, E, VEC, FLF>(pp, g, batch, s);
    else if (u8c) launch_one<NGP, E, VEC, FLF | FL_BC_U8C>(pp, g, batch, s);
    else launch_one<NGP, E, VEC, FLF | FL_BC>(pp, g, batch, s);
}

template <int NGP, int E, bool VEC>
static void launch_flags(const PoissonParams& pp, const Geom2D& g, int batch, hipStream_t s) {
    const int f = pp.fgp ? 2 : (pp.f ? 1 : 0);
    if (pp.nu) {
        if (f == 0) launch_bc<NGP, E, VEC, FL_NU>(pp, g, batch, s);
        else if (f == 1) launch_bc<NGP, E, VEC, FL_NU | FL_F>(pp, g, batch, s);
        else launch_bc<NGP, E, VEC, FL_NU | FL_FGP>(pp, g, batch, s);
    } else {
        if (f == 0) launch_bc<NGP, E, VEC, 0>(pp, g, batch, s);
        else if (f == 1) launch_bc<NGP, E, VEC, FL_F>(pp, g, batch, s);
        else launch_bc<NGP, E, VEC, FL_FGP>(pp, g, batch, s);
    }
}

template <int NGP>
static int launch_q1_2d(const PoissonParams& pp, const Geom2D& g, int batch, bool vec, hipStream_t s) {
    if (g.E == 4 && vec) { launch_flags<NGP, 4, true>(pp, g, batch, s); return 0; }
    if (g.E == 2 && vec) { launch_flags<NGP, 2, true>(pp, g, batch, s); return 0; }
    if (g.E == 2) { launch_flags<NGP, 2, false>(pp, g, batch, s); return 0; }
    return DN_E_UNSUPPORTED;
}

#define DN_CAT2(a, b) a##b
#define DN_CAT(a, b) DN_CAT2(a, b)
int DN_CAT(launch_poisson2d_q1_g, DN_NGP)(const PoissonParams& pp, const Geom2D& g, int batch, bool vec, hipStream_t s) {
    return launch_q1_2d<DN_NGP>(pp, g, batch, vec, s);
}

}  // namespace dn
