// 2-D Q1 fused Poisson kernel, fully sum-factorised marching form.  Included by
// poisson2d_q1_g{2,3,4}.hip with DN_NGP defined: one translation unit per Gauss order so the variants build in
// parallel.
//
// grid = (chunks_x, strips_y, B), block = T threads.  A thread owns E consecutive elements of a row.  Carried
// across the march, per element: the x-stage values of the lower node row (TU/TN/TF at the x-Gauss points, DX) and
// the cotangents of that row's x-stage values produced by the element layer below (CT, CDX).  Per layer: x-stage
// of the new row (1 sub + NGP FMAs per field), the O(NGP) layer arithmetic of q1_layer_2d, then ONE x-stage
// transpose per completed row, whose result is the finished nodal value (no separate node accumulators).
//
// What is present (nu, nodal f, f at Gauss points, Dirichlet conditions) is a compile-time flag set FL: run-time
// "is this pointer null" tests inside the march make the compiler unswitch/duplicate the loop and inflate the
// register allocation (measured: 226 vs 111 VGPRs), and every element is kept in its own basic block (the
// `if (valid)` below is also a scheduling fence: in one block the scheduler interleaves the E element streams and
// the live temporaries double).
#include "poisson_common.h"

namespace dn {

enum : int { FL_NU = 1, FL_F = 2, FL_FGP = 4, FL_BC = 8, FL_BC_U8C = 16 };   // FL_BC_U8C: uint8 masks with constant values only

template <int NGP, int E>
struct RowState2D {
    float TU[E][NGP], TN[E][NGP], TF[E][NGP], DX[E];
    float keep[E];
};

template <int E>
struct RowRaw2D {
    float u[E + 1], n[E + 1], f[E + 1];
    BcRaw<E> bc;
    uint32_t m8[2][2];      // FL_BC_U8C: packed mask bytes (vector word + the shared node's byte) per condition
};

#ifndef DN_Q1_2D_WAVES
#define DN_Q1_2D_WAVES 2
#endif

template <int NGP, int E, bool VEC, int FL>
__global__ void __launch_bounds__(256, DN_Q1_2D_WAVES) poisson2d_q1_kernel(const PoissonParams p) {
    constexpr int NW = E;
    constexpr bool HAS_NU = (FL & FL_NU) != 0, HAS_F = (FL & FL_F) != 0, FGP = (FL & FL_FGP) != 0;
    constexpr bool BC_ANY = (FL & (FL_BC | FL_BC_U8C)) != 0, BC_U8C = (FL & FL_BC_U8C) != 0;
    const int T = blockDim.x;
    const int tid = threadIdx.x;
    const int chunk = blockIdx.x, strip = blockIdx.y, b = blockIdx.z;
    const int q = chunk * (T - 1) + tid;  // logical thread column (chunks overlap by one thread)
    const int ex0 = q * E;                // first element == first node of this thread
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

    RowState2D<NGP, E> SA, SB;
    float CT[E][NGP], CDX[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        CDX[e] = 0.f;
        SA.keep[e] = SB.keep[e] = 1.f;
#pragma unroll
        for (int i = 0; i < NGP; ++i) { CT[e][i] = 0.f; SA.TN[e][i] = SB.TN[e][i] = 1.f; SA.TF[e][i] = SB.TF[e][i] = 0.f; }
    }

    // issue the raw loads of node row yr (clamped to the domain: a prefetch past the last row is discarded)
    auto row_issue = [&](int yr, RowRaw2D<E>& r) {
        const unsigned rowoff = (unsigned)min(yr, p.ny - 1) * (unsigned)p.nx;
        load_seg<NW, VEC>(sb.u, rowoff, x0, p.nx, r.u);
        if constexpr (HAS_NU) load_seg<NW, VEC>(sb.nu, rowoff, x0, p.nx, r.n);
        if constexpr (HAS_F) load_seg<NW, VEC>(sb.f, rowoff, x0, p.nx, r.f);
        if constexpr (BC_U8C) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (sb.mask[k] != nullptr) {          // the second condition is optional (wave-uniform)
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
    // Dirichlet conditions + x-stage of a loaded row
    auto row_stage = [&](RowRaw2D<E>& r, RowState2D<NGP, E>& S) {
        if constexpr (BC_U8C) {
#pragma unroll
            for (int n = 0; n < NW; ++n) S.keep[n] = 1.f;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (sb.mask[k] != nullptr) {
                    const float val = p.bc[k].value;
#pragma unroll
                    for (int n = 0; n <= NW; ++n) {
                        const bool set = n < NW ? ((r.m8[k][0] >> (8 * n)) & 0xffu) != 0u : r.m8[k][1] != 0u;
                        r.u[n] = set ? val : r.u[n];
                        if (n < NW) S.keep[n] = set ? 0.f : S.keep[n];
                    }
                }
            }
        } else if constexpr (BC_ANY) {
            bc_apply<NW>(p, sb, r.bc, r.u, S.keep);
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
            S.DX[e] = r.u[e + 1] - r.u[e];
#pragma unroll
            for (int i = 0; i < NGP; ++i) S.TU[e][i] = fmaf(p.T.b[i][1], S.DX[e], r.u[e]);
            if constexpr (HAS_NU) {
                const float d = r.n[e + 1] - r.n[e];
#pragma unroll
                for (int i = 0; i < NGP; ++i) S.TN[e][i] = fmaf(p.T.b[i][1], d, r.n[e]);
            }
            if constexpr (HAS_F) {
                const float d = r.f[e + 1] - r.f[e];
#pragma unroll
                for (int i = 0; i < NGP; ++i) S.TF[e][i] = fmaf(p.T.b[i][1], d, r.f[e]);
            }
        }
    };

    float e1_acc = 0.f, e2_acc = 0.f, sq_acc = 0.f;
    int par = 0;

    // Finish node row `yr`: o[n] holds this thread's contributions to nodes x0..x0+E; node x0 also receives
    // the left neighbour's o[E] through LDS.
    auto emit_row = [&](const float (&o)[NW + 1], const float (&keep)[NW], int yr, bool owned_row) {
        xch[par][tid] = o[NW];
        __syncthreads();
        const float left = (tid > 0) ? xch[par][tid - 1] : 0.f;
        par ^= 1;
        if (owned_row && col_owner) {
            float v[NW];
#pragma unroll
            for (int n = 0; n < NW; ++n) {
                const float t = (o[n] + (n == 0 ? left : 0.f)) * keep[n];
                sq_acc = fmaf(t, t, sq_acc);                 // nodes beyond the domain receive no contribution: t == 0
                v[n] = t * p.out_scale;
            }
            if (sb.out) store_seg<NW, VEC>(sb.out, (unsigned)yr * (unsigned)p.nx, x0, p.nx, v);
        }
    };

    // one element layer between the rows held in L (lower) and U (upper)
    auto layer = [&](int ey, const RowState2D<NGP, E>& L, const RowState2D<NGP, E>& U) {
        const bool own_layer = ey >= ey_own;
        const float cnt = (own_layer && col_owner) ? 1.f : 0.f;
        float o[NW + 1], le1 = 0.f, le2 = 0.f;
#pragma unroll
        for (int n = 0; n <= NW; ++n) o[n] = 0.f;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (ex0 + e < p.nelx) {       // elements beyond the domain are skipped (and: scheduling fence, see header)
                float fg[NGP * NGP];
                if constexpr (FGP) {
                    const unsigned eo = (unsigned)ey * (unsigned)p.nelx + (unsigned)(ex0 + e);
#pragma unroll
                    for (int gi = 0; gi < NGP * NGP; ++gi) fg[gi] = fgp[eo + (unsigned)gi * eps];
                }
                float ct0[NGP], ct1[NGP], cdx0, cdx1, e1, e2;
                q1_layer_2d<NGP, FGP>(p.T, L.TU[e], U.TU[e], L.DX[e], U.DX[e], L.TN[e], U.TN[e], L.TF[e], U.TF[e], fg, ct0, ct1,
                                      cdx0, cdx1, e1, e2);
                le1 += e1;
                le2 += e2;
                // row ey is complete for this element: x-stage transpose of (layer below + this layer)
                float ssum = 0.f, bsum = cdx0 + CDX[e];
#pragma unroll
                for (int i = 0; i < NGP; ++i) {
                    const float t = ct0[i] + CT[e][i];
                    ssum += t;
                    bsum = fmaf(p.T.b[i][1], t, bsum);
                    CT[e][i] = ct1[i];
                }
                CDX[e] = cdx1;
                o[e + 1] += bsum;
                o[e] += ssum - bsum;
            }
        }
        e1_acc = fmaf(cnt, le1, e1_acc);
        e2_acc = fmaf(cnt, le2, e2_acc);
        emit_row(o, L.keep, ey, own_layer);
    };

    RowRaw2D<E> raw;
    row_issue(ey_begin, raw);
    row_stage(raw, SA);
    int ey = ey_begin;
#ifdef DN_PREFETCH
    row_issue(ey_begin + 1, raw);
    for (; ey < ey_end; ++ey) {
        row_stage(raw, SB);                // row ey+1: loads issued one layer ago
        row_issue(ey + 2, raw);            // prefetch: in flight while this layer is computed
        layer(ey, SA, SB);
        SA = SB;
    }
#else
    for (; ey < ey_end; ++ey) {
        row_issue(ey + 1, raw);
        row_stage(raw, SB);
        layer(ey, SA, SB);
        SA = SB;                           // the upper row becomes the lower row of the next layer
    }
#endif
    // the last strip also owns the top boundary row of the domain: only the layer below contributes
    if (ey_end == p.nely) {
        float o[NW + 1];
#pragma unroll
        for (int n = 0; n <= NW; ++n) o[n] = 0.f;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (ex0 + e < p.nelx) {
                float ssum = 0.f, bsum = CDX[e];
#pragma unroll
                for (int i = 0; i < NGP; ++i) { ssum += CT[e][i]; bsum = fmaf(p.T.b[i][1], CT[e][i], bsum); }
                o[e + 1] += bsum;
                o[e] += ssum - bsum;
            }
        }
        emit_row(o, SA.keep, p.ny - 1, true);
    }

    if (p.want_sums) finish_sums(p, e1_acc, e2_acc, sq_acc, tid, T, red, &last_flag);
}

// ---- dispatch over the compile-time flag set --------------------------------------------------------------
template <int NGP, int E, bool VEC, int FL>
static void launch_one(const PoissonParams& pp, const Geom2D& g, int batch, hipStream_t s) {
    hipLaunchKernelGGL((poisson2d_q1_kernel<NGP, E, VEC, FL>), dim3(g.chunks, g.strips, batch), dim3(g.T), 0, s, pp);
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
