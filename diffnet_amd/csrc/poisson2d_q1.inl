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
#include <cstdlib>

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
#ifndef DN_PRIO_ROT
#define DN_PRIO_ROT 3      // 0 = off (A/B switch); measured -7 % kernel time at the bench shape
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

    RowState2D<NGP, E> SA;
    float CT[E][NGP], CDX[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        CDX[e] = 0.f;
        SA.keep[e] = 1.f;
#pragma unroll
        for (int i = 0; i < NGP; ++i) { CT[e][i] = 0.f; SA.TN[e][i] = 1.f; SA.TF[e][i] = 0.f; }
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
#ifndef DN_ABLATE_XCH                      // timing experiment only: drop the neighbour hand-over
        xch[par][tid] = o[NW];
        // LDS-only barrier: __syncthreads() would also drain vmcnt, i.e. wait for the prefetched next row
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const float left = (tid > 0) ? xch[par][tid - 1] : 0.f;
        par ^= 1;
#else
        const float left = 0.f;
#endif
        if (owned_row && col_owner) {
            float v[NW];
#pragma unroll
            for (int n = 0; n < NW; ++n) {
                const float t = (o[n] + (n == 0 ? left : 0.f)) * keep[n];
                sq_acc = fmaf(t, t, sq_acc);                 // nodes beyond the domain receive no contribution: t == 0
                v[n] = t * p.out_scale;
            }
#ifndef DN_ABLATE_STORE
            if (sb.out) store_seg<NW, VEC>(sb.out, (unsigned)yr * (unsigned)p.nx, x0, p.nx, v);
#else
            if (sb.out && v[0] == 12345.678f) store_seg<NW, VEC>(sb.out, (unsigned)yr * (unsigned)p.nx, x0, p.nx, v);
#endif
        }
    };

    // One element layer between the staged lower row S (state, updated IN PLACE) and the freshly loaded upper row `r`:
    // the upper row's x-stage values live only while their element is processed, so a second row state (and the copy
    // between the two) is never materialised -- ~30 VGPRs less than staging the whole upper row first.
    auto layer = [&](int ey, RowState2D<NGP, E>& S, RowRaw2D<E>& r) {
        const bool own_layer = ey >= ey_own;
        const float cnt = (own_layer && col_owner) ? 1.f : 0.f;
        float keep_up[NW];
#pragma unroll
        for (int n = 0; n < NW; ++n) keep_up[n] = 1.f;
        if constexpr (BC_U8C) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (sb.mask[k] != nullptr) {
                    const float val = p.bc[k].value;
#pragma unroll
                    for (int n = 0; n <= NW; ++n) {
                        const bool set = n < NW ? ((r.m8[k][0] >> (8 * n)) & 0xffu) != 0u : r.m8[k][1] != 0u;
                        r.u[n] = set ? val : r.u[n];
                        if (n < NW) keep_up[n] = set ? 0.f : keep_up[n];
                    }
                }
            }
        } else if constexpr (BC_ANY) {
            bc_apply<NW>(p, sb, r.bc, r.u, keep_up);
        }
        float o[NW + 1], le1 = 0.f, le2 = 0.f;
#pragma unroll
        for (int n = 0; n <= NW; ++n) o[n] = 0.f;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (ex0 + e < p.nelx) {       // elements beyond the domain are skipped (and: scheduling fence, see header)
                // x-stage of the upper row for this element only
                float TU1[NGP], TN1[NGP], TF1[NGP];
                const float DX1 = r.u[e + 1] - r.u[e];
#pragma unroll
                for (int i = 0; i < NGP; ++i) { TU1[i] = fmaf(p.T.b[i][1], DX1, r.u[e]); TN1[i] = 1.f; TF1[i] = 0.f; }
                if constexpr (HAS_NU) {
                    const float d = r.n[e + 1] - r.n[e];
#pragma unroll
                    for (int i = 0; i < NGP; ++i) TN1[i] = fmaf(p.T.b[i][1], d, r.n[e]);
                }
                if constexpr (HAS_F) {
                    const float d = r.f[e + 1] - r.f[e];
#pragma unroll
                    for (int i = 0; i < NGP; ++i) TF1[i] = fmaf(p.T.b[i][1], d, r.f[e]);
                }
                float fg[NGP * NGP];
                if constexpr (FGP) {
                    const unsigned eo = (unsigned)ey * (unsigned)p.nelx + (unsigned)(ex0 + e);
#pragma unroll
                    for (int gi = 0; gi < NGP * NGP; ++gi) fg[gi] = fgp[eo + (unsigned)gi * eps];
                }
                float ct0[NGP], ct1[NGP], cdx0, cdx1, e1, e2;
#ifndef DN_ABLATE_COMPUTE
                q1_layer_2d<NGP, FGP>(p.T, S.TU[e], TU1, S.DX[e], DX1, S.TN[e], TN1, S.TF[e], TF1, fg, ct0, ct1, cdx0, cdx1, e1, e2);
#else                                      // timing experiment only: keep every load alive with trivial arithmetic
#pragma unroll
                for (int i = 0; i < NGP; ++i) { ct0[i] = S.TU[e][i] + TN1[i]; ct1[i] = TU1[i] + TF1[i]; }
                cdx0 = S.DX[e]; cdx1 = DX1; e1 = cdx0; e2 = cdx1;
#endif
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
                    S.TU[e][i] = TU1[i];       // the upper row becomes the lower row of the next layer
                    if constexpr (HAS_NU) S.TN[e][i] = TN1[i];
                    if constexpr (HAS_F) S.TF[e][i] = TF1[i];
                }
                CDX[e] = cdx1;
                S.DX[e] = DX1;
                o[e + 1] += bsum;
                o[e] += ssum - bsum;
            }
        }
        e1_acc = fmaf(cnt, le1, e1_acc);
        e2_acc = fmaf(cnt, le2, e2_acc);
        emit_row(o, S.keep, ey, own_layer);
#pragma unroll
        for (int n = 0; n < NW; ++n) S.keep[n] = keep_up[n];
    };

    RowRaw2D<E> raw;
    row_issue(ey_begin, raw);
    row_stage(raw, SA);
    int ey = ey_begin;
#ifdef DN_PREFETCH
#error "register prefetch was measured slower and has been removed; see profiles/README.md"
#else
    auto set_prio = [&](int e) {
#if DN_PRIO_ROT
        // rotate the wave priority with its progress: equal-priority waves are served oldest-first, so the four waves of
        // a SIMD finish one after the other and the tail runs at low occupancy; a progress-dependent priority makes
        // them advance at the same rate
        switch (((ey_end - e) >> 1) & 3) {          // s_setprio takes an immediate
            case 0: __builtin_amdgcn_s_setprio(0); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            default: __builtin_amdgcn_s_setprio(3); break;
        }
#endif
    };
    // (a two-layers-per-trip ping-pong of the row states removes the SA = SB copies but was measured slower: 131 vs 115
    // VGPRs drops the occupancy from 4 to 3 waves per SIMD)
    for (; ey < ey_end; ++ey) {
        set_prio(ey);
#ifndef DN_ABLATE_MEM                      // timing experiment only: reuse the first row's data
        row_issue(ey + 1, raw);
#endif
        layer(ey, SA, raw);
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

// =============================================================================================
// Raw-carry form of the same kernel (selected by default, DN_2D_STAGED_STATE selects the form above).
// The carried lower row is kept as RAW nodal values (E + 1 per field) instead of its x-stage (NGP per element and
// field): 5 instead of 10 carried registers per element at E = 4, NGP = 3.  Its x-stage is recomputed where the element
// is processed (one subtraction + NGP FMAs per field -- what the state copies of the staged form cost anyway).  Two rows
// are processed per loop trip with the two raw rows and the two cotangent sets swapping roles, so nothing is copied.
// =============================================================================================
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
    const int chunk = blockIdx.x, strip = blockIdx.y, b = blockIdx.z;
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

// =============================================================================================
// LDS-DMA variant (E = 4, aligned rows, Dirichlet masks absent or uint8 + constant values).
// Same arithmetic as poisson2d_q1_kernel; the difference is how node rows reach the registers: every wave streams
// its 256-node row segments with `global_load_lds` (LDS-DMA: no VGPR destination) into a private two-slot ring,
// two rows ahead of the layer being computed, and reads them back with ds_read_b128.  This (a) keeps ~2 rows per
// wave in flight without prefetch registers, (b) removes the three extra per-row loads of the node shared with the
// next lane (it is the next lane's first float in LDS), (c) leaves the loaded-latency (~4 us under load) two full
// layer rounds to hide in.  Waits are counted (`s_waitcnt vmcnt(NG)` leaves the younger row in flight) and the
// hand-over barrier is LDS-only, as a `__syncthreads()` would drain the DMA (cdna_hip_programming.md section 5,
// "Pipelining across barriers").  All LDS is one dynamic array (same section, trap 4a).
// =============================================================================================
template <int NGP, int FL>
__global__ void __launch_bounds__(256, DN_Q1_2D_WAVES) poisson2d_q1_dma_kernel(const PoissonParams p) {
    constexpr int E = 4, NW = 4;
    constexpr bool HAS_NU = (FL & FL_NU) != 0, HAS_F = (FL & FL_F) != 0;
    constexpr bool BC_U8C = (FL & FL_BC_U8C) != 0;
    constexpr int NF = 1 + (HAS_NU ? 1 : 0) + (HAS_F ? 1 : 0);
    constexpr int NM = BC_U8C ? 2 : 0;
    constexpr int FSLOT = 1024 + 256, MSLOT = 256 + 256;         // bytes per field / mask per row
    constexpr int SLOT = NF * FSLOT + NM * MSLOT;
    constexpr int NG = 2 * (NF + NM);                             // LDS-DMA instructions per row
    static_assert((FL & (FL_FGP | FL_BC)) == 0, "LDS-DMA kernel: nodal forcing and uint8/constant Dirichlet only");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int T = blockDim.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwave = T >> 6;
    const int chunk = blockIdx.x, strip = blockIdx.y, b = blockIdx.z;
    const int q = chunk * (T - 1) + tid;
    const int ex0 = q * E, x0 = ex0;
    const int xw = (chunk * (T - 1) + wave * 64) * E;             // first node of this wave
    const bool col_owner = !(chunk > 0 && tid == 0);
    const int64_t nps = (int64_t)p.nx * p.ny;
    const SampleBases sb = sample_bases(p, b, nps);
    const int R = p.rows_per_strip;
    const int ey_own = strip * R;
    const int ey_begin = ey_own > 0 ? ey_own - 1 : 0;
    const int ey_end = min(ey_own + R, p.nely);
    // LDS carve-up: [per-wave rings][xch 2 x T floats][red 8 doubles][flag]
    unsigned char* ring = lds + wave * 2 * SLOT;
    float* xch = reinterpret_cast<float*>(lds + nwave * 2 * SLOT);
    double* red = reinterpret_cast<double*>(lds + nwave * 2 * SLOT + 2 * 256 * 4);
    int* last_flag = reinterpret_cast<int*>(lds + nwave * 2 * SLOT + 2 * 256 * 4 + 64);

    using gptr = const __attribute__((address_space(1))) void*;
    using lptr = __attribute__((address_space(3))) void*;
    const float* fbase[3] = {sb.u, HAS_NU ? sb.nu : sb.f, sb.f};
    const uint8_t* mbase[2] = {reinterpret_cast<const uint8_t*>(sb.mask[0]),
                               reinterpret_cast<const uint8_t*>(sb.mask[1] ? sb.mask[1] : sb.mask[0])};
    const float mval[2] = {p.bc[0].value, sb.mask[1] ? p.bc[1].value : p.bc[0].value};   // absent 2nd condition: repeat the 1st

    const unsigned xl = (unsigned)min(x0, p.nx - NW);                       // this lane's 4 nodes (clamped, aligned)
    const unsigned xe = (unsigned)min(xw + 256 + lane, p.nx - 1);           // the node after the wave's segment (+ spare lanes)
    const unsigned xe4 = (unsigned)min(xw + 256 + 4 * lane, p.nx - NW);     // same for byte masks: LDS-DMA moves whole aligned dwords
    auto issue_row = [&](int yr) {
        const unsigned rowoff = (unsigned)min(yr, p.ny - 1) * (unsigned)p.nx;
        unsigned char* slot = ring + (yr & 1) * SLOT;
#pragma unroll
        for (int k = 0; k < NF; ++k) {
            __builtin_amdgcn_global_load_lds((gptr)(fbase[k] + (rowoff + xl)), (lptr)(slot + k * FSLOT), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr)(fbase[k] + (rowoff + xe)), (lptr)(slot + k * FSLOT + 1024), 4, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < NM; ++k) {
            __builtin_amdgcn_global_load_lds((gptr)(mbase[k] + (rowoff + xl)), (lptr)(slot + NF * FSLOT + k * MSLOT), 4, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr)(mbase[k] + (rowoff + xe4)), (lptr)(slot + NF * FSLOT + k * MSLOT + 256), 4, 0, 0);
        }
    };

    RowState2D<NGP, E> SA, SB;
    float CT[E][NGP], CDX[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        CDX[e] = 0.f;
        SA.keep[e] = SB.keep[e] = 1.f;
#pragma unroll
        for (int i = 0; i < NGP; ++i) { CT[e][i] = 0.f; SA.TN[e][i] = SB.TN[e][i] = 1.f; SA.TF[e][i] = SB.TF[e][i] = 0.f; }
    }

    // read a landed row out of the ring, apply Dirichlet conditions and the x-stage.
    // The ring is read with inline-asm ds_read: for a compiler-visible LDS read hipcc (ROCm 7.2) conservatively drains
    // every LDS-DMA in flight (`s_waitcnt vmcnt(0)`), which would serialise the two-row prefetch; the asm reads are
    // ordered by the counted vmcnt wait above them and retired by the explicit lgkmcnt(0) below.
    typedef float f4 __attribute__((ext_vector_type(4)));
    const unsigned ring_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)ring;
    auto read_stage = [&](int yr, RowState2D<NGP, E>& S) {
        const unsigned a16 = ring_lds + (unsigned)(yr & 1) * SLOT + (unsigned)lane * 16u;
        const unsigned a4 = ring_lds + (unsigned)(yr & 1) * SLOT + (unsigned)lane * 4u;
        f4 v[3];
        float nx1[3];
        unsigned mw[2] = {0u, 0u}, mn[2] = {0u, 0u};
#pragma unroll
        for (int k = 0; k < NF; ++k) {
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[k]) : "v"(a16), "n"(k * FSLOT));
            asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(nx1[k]) : "v"(a16), "n"(k * FSLOT + 16));
        }
#pragma unroll
        for (int k = 0; k < NM; ++k) {
            asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(mw[k]) : "v"(a4), "n"(NF * FSLOT + k * MSLOT));
            asm volatile("ds_read_u8 %0, %1 offset:%2" : "=v"(mn[k]) : "v"(a4), "n"(NF * FSLOT + k * MSLOT + 4));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float r[3][NW + 1];
#pragma unroll
        for (int k = 0; k < NF; ++k) {
            asm volatile("" : "+v"(v[k]), "+v"(nx1[k]));                  // consumers stay behind the wait
            r[k][0] = v[k].x; r[k][1] = v[k].y; r[k][2] = v[k].z; r[k][3] = v[k].w;
            r[k][4] = nx1[k];
        }
#pragma unroll
        for (int k = 0; k < NM; ++k) asm volatile("" : "+v"(mw[k]), "+v"(mn[k]));
        if constexpr (BC_U8C) {
#pragma unroll
            for (int n = 0; n < NW; ++n) S.keep[n] = 1.f;
#pragma unroll
            for (int k = 0; k < NM; ++k) {
                const uint32_t w = mw[k], wn = mn[k];
#pragma unroll
                for (int n = 0; n <= NW; ++n) {
                    const bool set = n < NW ? ((w >> (8 * n)) & 0xffu) != 0u : wn != 0u;
                    r[0][n] = set ? mval[k] : r[0][n];
                    if (n < NW) S.keep[n] = set ? 0.f : S.keep[n];
                }
            }
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
            S.DX[e] = r[0][e + 1] - r[0][e];
#pragma unroll
            for (int i = 0; i < NGP; ++i) S.TU[e][i] = fmaf(p.T.b[i][1], S.DX[e], r[0][e]);
            if constexpr (HAS_NU) {
                const float d = r[1][e + 1] - r[1][e];
#pragma unroll
                for (int i = 0; i < NGP; ++i) S.TN[e][i] = fmaf(p.T.b[i][1], d, r[1][e]);
            }
            if constexpr (HAS_F) {
                constexpr int kf = HAS_NU ? 2 : 1;
                const float d = r[kf][e + 1] - r[kf][e];
#pragma unroll
                for (int i = 0; i < NGP; ++i) S.TF[e][i] = fmaf(p.T.b[i][1], d, r[kf][e]);
            }
        }
    };

    float e1_acc = 0.f, e2_acc = 0.f, sq_acc = 0.f;
    int par = 0;

    auto emit_row = [&](const float (&o)[NW + 1], const float (&keep)[NW], int yr, bool owned_row) {
        xch[par * 256 + tid] = o[NW];
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // LDS-only: the DMA stays in flight
        const float left = (tid > 0) ? xch[par * 256 + tid - 1] : 0.f;
        par ^= 1;
        if (owned_row && col_owner) {
            float v[NW];
#pragma unroll
            for (int n = 0; n < NW; ++n) {
                const float t = (o[n] + (n == 0 ? left : 0.f)) * keep[n];
                sq_acc = fmaf(t, t, sq_acc);
                v[n] = t * p.out_scale;
            }
            if (sb.out) store_seg<NW, true>(sb.out, (unsigned)yr * (unsigned)p.nx, x0, p.nx, v);
        }
    };

    auto layer = [&](int ey, const RowState2D<NGP, E>& L, const RowState2D<NGP, E>& U) {
        const bool own_layer = ey >= ey_own;
        const float cnt = (own_layer && col_owner) ? 1.f : 0.f;
        float o[NW + 1], le1 = 0.f, le2 = 0.f;
#pragma unroll
        for (int n = 0; n <= NW; ++n) o[n] = 0.f;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (ex0 + e < p.nelx) {
                float ct0[NGP], ct1[NGP], cdx0, cdx1, e1, e2;
#ifndef DN_ABLATE_COMPUTE
                q1_layer_2d<NGP, false>(p.T, L.TU[e], U.TU[e], L.DX[e], U.DX[e], L.TN[e], U.TN[e], L.TF[e], U.TF[e], nullptr, ct0,
                                        ct1, cdx0, cdx1, e1, e2);
#else
#pragma unroll
                for (int i = 0; i < NGP; ++i) { ct0[i] = L.TU[e][i] + U.TN[e][i]; ct1[i] = U.TU[e][i] + U.TF[e][i]; }
                cdx0 = L.DX[e]; cdx1 = U.DX[e]; e1 = cdx0; e2 = cdx1;
#endif
                le1 += e1;
                le2 += e2;
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

    issue_row(ey_begin);
    issue_row(ey_begin + 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NG) : "memory");           // row ey_begin landed, ey_begin+1 in flight
    read_stage(ey_begin, SA);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // ring reads done before the slot is refilled
    issue_row(ey_begin + 2);
    for (int ey = ey_begin; ey < ey_end; ++ey) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NG) : "memory");       // row ey+1 landed; row ey+2 may still fly
        read_stage(ey + 1, SB);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        issue_row(ey + 3);
        layer(ey, SA, SB);
        SA = SB;
    }
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // no DMA may outlive the workgroup's LDS
    if (p.want_sums) finish_sums(p, e1_acc, e2_acc, sq_acc, tid, T, red, last_flag);
}

template <int NGP, int FL>
static void launch_dma(const PoissonParams& pp, const Geom2D& g, int batch, hipStream_t s) {
    constexpr int NF = 1 + ((FL & FL_NU) ? 1 : 0) + ((FL & FL_F) ? 1 : 0), NM = (FL & FL_BC_U8C) ? 2 : 0;
    const size_t lds = (size_t)(g.T / 64) * 2 * (NF * 1280 + NM * 512) + 2 * 256 * 4 + 64 + 16;
    hipLaunchKernelGGL((poisson2d_q1_dma_kernel<NGP, FL>), dim3(g.chunks, g.strips, batch), dim3(g.T), lds, s, pp);
}

// ---- dispatch over the compile-time flag set --------------------------------------------------------------
template <int NGP, int E, bool VEC, int FL>
static void launch_one(const PoissonParams& pp, const Geom2D& g, int batch, hipStream_t s) {
#ifdef DN_2D_STAGED_STATE
    hipLaunchKernelGGL((poisson2d_q1_kernel<NGP, E, VEC, FL>), dim3(g.chunks, g.strips, batch), dim3(g.T), 0, s, pp);
#else
    hipLaunchKernelGGL((poisson2d_q1_raw_kernel<NGP, E, VEC, FL>), dim3(g.chunks, g.strips, batch), dim3(g.T), 0, s, pp);
#endif
}

template <int NGP, int E, bool VEC, int FLF>   // FLF: nu / f flags already fixed
static void launch_bc(const PoissonParams& pp, const Geom2D& g, int batch, hipStream_t s) {
    const bool any = pp.bc[0].mask || pp.bc[1].mask;
    bool u8c = any && pp.bc[0].mask != nullptr;
    for (int k = 0; k < 2; ++k)
        if (pp.bc[k].mask && (!pp.bc[k].mask_is_u8 || pp.bc[k].field)) u8c = false;
    // The LDS-DMA variant is functionally complete and parity-tested (DN_USE_DMA=1), but measured 5-15 % SLOWER than the
    // register path on MI355X at the bench shape (profiles/README.md): the kernel is not load-latency bound, so the
    // two-row-deep ring buys nothing and costs LDS round trips.  It stays opt-in.
    static const bool use_dma = getenv("DN_USE_DMA") != nullptr;
    if constexpr (E == 4 && VEC && (FLF & FL_FGP) == 0) {
        if (use_dma && (!any || u8c)) {
            if (!any) launch_dma<NGP, FLF>(pp, g, batch, s);
            else launch_dma<NGP, FLF | FL_BC_U8C>(pp, g, batch, s);
            return;
        }
    }
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
