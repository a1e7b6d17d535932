// Fused FSDT (Mindlin) plate residuals, ASSEMBLED-STENCIL form (round 4; the element form is fsdt.hip).
//
// Same operator as fsdt.hip -- examples/elasticity/single_instance/e1_plate_bending_fsdt.py:128-232 of the reference: Dirichlet
// substitution, nine Gauss-point evaluations, constitutive law, three weak-form residuals, assembly, Dirichlet rows, Frobenius sums --
// on a different factorisation.  The constitutive coefficients are constants and the mesh is uniform, so every term of the residuals is
// a tensor product of two 1-D element matrices of the quadrature rule applied to one nodal field:
//     M[a][b] = sum_g w_g N_a N_b     K[a][b] = sum_g w_g N_a' N_b'     C[a][b] = sum_g w_g N_a' N_b     (C^T: test value, trial derivative)
//     R1 = A55 (C_x (x) M_y) phi_x + A55 (K_x (x) M_y) w + A44 (M_x (x) C_y) phi_y + A44 (M_x (x) K_y) w - q l_x (x) l_y
//     R2 = D11 (K_x (x) M_y) phi_x + D12 (C_x (x) C_y^T) phi_y + D66 (M_x (x) K_y) phi_x + D66 (C_x^T (x) C_y) phi_y + A55 (M_x (x) M_y) phi_x + A55 (C_x^T (x) M_y) w
//     R3 = D66 (C_x (x) C_y^T) phi_x + D66 (K_x (x) M_y) phi_y + D12 (C_x^T (x) C_y) phi_x + D22 (M_x (x) K_y) phi_y + A44 (M_x (x) M_y) phi_y + A44 (M_x (x) C_y^T) w
// (exact for ANY rule: the quadrature of a tensor-product integrand with constant coefficients factorises; the matrices are built on the
// host from the mesh's own tables).  Assembled over the elements the x factor is a banded operator along a node row: applied at the
// thread's OWN nodes (the vertex node takes the left element's last row and the right element's first row, interior nodes one row),
// 11 products per node row; the y factor is applied by scattering a row's products into the node rows it couples to (a window of 2 P + 1
// accumulator rows marches with the strip).  Against the element form (sum-factorised Gauss-point loop, ~920 VALU instructions per Q2
// element) this is ~11 x 4.5 x 2 multiply-adds per NODE (4 nodes per Q2 element), no per-element result to hand to a neighbour thread:
// the only exchange is the left neighbour's node values (ds_bpermute), so a wave is self-contained -- no LDS slot, no barrier, and a
// workgroup is just a bundle of independent one-wave chunks of 62 element columns: lanes 0 and 63 are ghosts that only supply their
// neighbours' halo (lane 0 of the first chunk owns the mesh's first column), so no node is loaded by two lanes of a wave.
//
// Launch: grid = (ceil(chunks / waves per workgroup), strips of R element rows, B).  A strip recomputes the layer under its first row
// (as the element form does); bitwise repeatable, no atomics on the data path.
//
// Measured on MI355X (profiles/r4_fsdt_stencil.txt; 1025^2 nodes, Q2, 3 x 3 points, fp32 mask, in-kernel sums): one sample 27.6 -> 21.3 us, eight
// samples 95.9 -> 59.4 us against the element form; ~480 VALU instructions (139 packed) per element layer, 117-125 VGPRs.  With the arithmetic removed
// (-DDN_ST_ABL_MATH) the eight-sample launch takes 52 us: the kernel sits on its access pattern; the in-kernel sums cost 4.5-6.6 us per launch, which
// the loss + gradient pair avoids by deferring them to the second launch (dn_fsdt_args.defer_sums / den_workspace, fsdt_common.h).
#include <algorithm>
#include <cstdlib>

#include "fsdt_common.h"

namespace dn {

constexpr unsigned FSDT_MATS_MAGIC = 0x46534454u;       // "FSDT"
struct FsdtMats {
    unsigned magic;        // FSDT_MATS_MAGIC: what the kernel must find where it computes the matrices to lie in the kernel-argument segment (st_fresh)
    float x[3][4][4];      // op 0: M, 1: K, 2: C; x axis: weights gpw, derivatives scaled by 2 / hx
    float y[3][4][4];      // y axis: weights gpw * wscale, derivatives scaled by 2 / hy
    float lx[4], ly[4];    // sum_g w_g N_a (ly carries wscale)
};

enum { ST_M = 0, ST_K = 1, ST_C = 2 };

#ifndef DN_ST_VEC2
#define DN_ST_VEC2 1              // (Q2) the two own nodes of a row as ONE 8-byte access (4-byte aligned on rows of an odd number of nodes: the hardware takes it)
#endif                            // instead of two 4-byte ones.  Equal where the arrays come out of the 256 MB cache (single launches re-run on the same buffers:
                                  // 55.1 vs 55.4 us at eight samples), 11-15 % faster where they come from HBM -- the loss + gradient pair at eight samples:
                                  // 130 -> 116 us with fp32 masks, 127 -> 109 us with uint8 masks (profiles/r4_fsdt_stencil.txt sections 4 and 9)
// The thread's own nodes of a row.  NW == 2 with DN_ST_VEC2: one access of two elements at min(x0, nx - 2); the closing column (x0 == nx - 1) takes the second half.
template <int NW, typename T>
__device__ __forceinline__ void st_load_own(const T* __restrict__ base, unsigned rowoff, int x0, int nx, T (&dst)[NW]) {
#if DN_ST_VEC2
    if constexpr (NW == 2) {
        const unsigned xl = (unsigned)min(x0, nx - 2);
        const bool last = x0 > nx - 2;
        if constexpr (sizeof(T) == 4) {
            const float2 v = ld_at<float2>(base, rowoff + xl);
            const float a = last ? v.y : v.x;
            dst[0] = reinterpret_cast<const T&>(a);
            dst[1] = reinterpret_cast<const T&>(v.y);
        } else {
            const uint16_t w = ld_at<uint16_t>(base, rowoff + xl);
            dst[0] = (T)(last ? (w >> 8) : (w & 0xffu));
            dst[1] = (T)(w >> 8);
        }
        return;
    }
#endif
    load_own<NW, false>(base, rowoff, x0, nx, dst);
}
template <int NW>
__device__ __forceinline__ void st_store_own(float* __restrict__ base, unsigned rowoff, int x0, int nx, const float (&src)[NW]) {
#if DN_ST_VEC2
    if constexpr (NW == 2) {
        if (x0 + 1 < nx) st_at<float2>(base, rowoff + (unsigned)x0, make_float2(src[0], src[1]));
        else if (x0 < nx) st_at<float>(base, rowoff + (unsigned)x0, src[0]);
        return;
    }
#endif
    store_seg<NW, false>(base, rowoff, x0, nx, src);
}

// The matrices are read from the kernel-argument segment through a pointer the compiler cannot see through (st_fresh), once per phase of a
// layer: 54 matrix entries + the coefficients + the launch's pointers do not fit the 100 SGPRs, and what does not fit is spilled into VGPR
// lanes and read back with v_readlane (~33 cycles of the SIMD each: 150 of them per layer in the first packed build).  A phase (x factor of the
// layer's rows; y factor) re-reads its <= 21 entries with scalar loads that hit the scalar cache; M and K are symmetric: the upper triangle.
typedef const FsdtMats __attribute__((address_space(4)))* st_mats_ptr;
__device__ __forceinline__ st_mats_ptr st_fresh(st_mats_ptr m) {
    asm volatile("" : "+s"(m));
    return m;
}
template <int OP, bool TR>
__device__ __forceinline__ float st_mx(st_mats_ptr m, int a, int b) {
    if constexpr (OP == ST_C) return TR ? m->x[OP][b][a] : m->x[OP][a][b];
    return a <= b ? m->x[OP][a][b] : m->x[OP][b][a];
}
template <int OP, bool TR>
__device__ __forceinline__ float st_my(st_mats_ptr m, int a, int b) {
    if constexpr (OP == ST_C) return TR ? m->y[OP][b][a] : m->y[OP][a][b];
    return a <= b ? m->y[OP][a][b] : m->y[OP][b][a];
}

// Value types: float, or v2f = (phi_x, phi_y) / two neighbouring own nodes in the halves of a 64-bit register pair (v_pk_fma_f32 / v_pk_mul_f32: one
// issue slot for two multiply-adds; wave-uniform matrix entries broadcast into both halves through op_sel) -- poisson_elem.h
typedef float st_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float st_fma(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ st_v2f st_fma(float a, st_v2f b, st_v2f c) { return __builtin_elementwise_fma((st_v2f)(a), b, c); }
__device__ __forceinline__ float st_mul(float a, float b) { return a * b; }
__device__ __forceinline__ st_v2f st_mul(float a, st_v2f b) { return (st_v2f)(a) * b; }

// x factor at the thread's own nodes: o[0] = left element's row P against (left nodes, own node 0) + right element's row 0; o[n] = row n
template <int P, int OP, bool TR, typename V>
__device__ __forceinline__ void st_xop(st_mats_ptr m, const V (&uL)[P + 1], const V (&uR)[P + 1], V (&o)[P]) {
    V s = st_mul(st_mx<OP, TR>(m, P, 0), uL[0]);
#pragma unroll
    for (int b = 1; b <= P; ++b) s = st_fma(st_mx<OP, TR>(m, P, b), uL[b], s);
#pragma unroll
    for (int b = 0; b <= P; ++b) s = st_fma(st_mx<OP, TR>(m, 0, b), uR[b], s);
    o[0] = s;
#pragma unroll
    for (int n = 1; n < P; ++n) {
        V t = st_mul(st_mx<OP, TR>(m, n, 0), uR[0]);
#pragma unroll
        for (int b = 1; b <= P; ++b) t = st_fma(st_mx<OP, TR>(m, n, b), uR[b], t);
        o[n] = t;
    }
}

// y factor: the products of the node row at local index JIN of an element layer go to that layer's rows, window rows BASE .. BASE + P;
// the own nodes two at a time in packed registers
template <int P, int OP, bool TR, int BASE, int JIN>
__device__ __forceinline__ void st_yscatter(st_mats_ptr m, const float (&g)[P], float (&acc)[2 * P + 1][P]) {
#pragma unroll
    for (int jo = 0; jo <= P; ++jo) {
        const float t = st_my<OP, TR>(m, jo, JIN);
#pragma unroll
        for (int n = 0; n + 1 < P; n += 2) {
            const st_v2f r = st_fma(t, (st_v2f){g[n], g[n + 1]}, (st_v2f){acc[BASE + jo][n], acc[BASE + jo][n + 1]});
            acc[BASE + jo][n] = r.x;
            acc[BASE + jo][n + 1] = r.y;
        }
        if constexpr (P % 2 == 1) acc[BASE + jo][P - 1] = fmaf(t, g[P - 1], acc[BASE + jo][P - 1]);
    }
}

// The eleven (y operator, residual) groups of one node row, after the x factor and the constitutive coefficients
template <int P>
struct StGroups {
    float r1m[P], r1c[P], r1k[P];
    float r2m[P], r2ct[P], r2k[P], r2c[P];
    float r3ct[P], r3m[P], r3c[P], r3k[P];
};

// cw: w, cxy: (phi_x, phi_y) at the own nodes + the node shared with the right neighbour (after scaling and the Dirichlet substitution)
template <int P>
__device__ __forceinline__ void st_stage(const FsdtParams& p, st_mats_ptr m, const float (&cw)[P + 1], const st_v2f (&cxy)[P + 1], float lf, float okf,
                                         StGroups<P>& G) {
#ifdef DN_ST_ABL_MATH             // measurement build (tools/variant_build.sh): the access pattern without the arithmetic
#pragma unroll
    for (int n = 0; n < P; ++n) {
        const float t = cw[n] + cxy[n].x + cxy[n].y + cw[P] * lf + okf;
        G.r1m[n] = G.r1c[n] = G.r1k[n] = G.r2m[n] = G.r2ct[n] = G.r2k[n] = G.r2c[n] = G.r3ct[n] = G.r3m[n] = G.r3c[n] = G.r3k[n] = t;
    }
    return;
#endif
    float wL[P + 1], wR[P + 1];
    st_v2f xyL[P + 1], xyR[P + 1];
#pragma unroll
    for (int b = 0; b < P; ++b) {            // the left neighbour's own nodes = the left element's first P nodes
        float t0 = __shfl_up(cw[b], 1, 64), t1 = __shfl_up(cxy[b].x, 1, 64), t2 = __shfl_up(cxy[b].y, 1, 64);
        asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2));         // (the exchange stays where every lane takes part)
        wL[b] = lf * t0;
        xyL[b] = st_mul(lf, (st_v2f){t1, t2});
    }
    wL[P] = lf * cw[0];
    xyL[P] = st_mul(lf, cxy[0]);
#pragma unroll
    for (int b = 0; b <= P; ++b) {
        wR[b] = okf * cw[b];
        xyR[b] = st_mul(okf, cxy[b]);
    }
    float wK[P], wM[P], wCT[P];
    st_v2f C[P], K[P], M[P], CT[P];          // .x: of phi_x, .y: of phi_y
    st_xop<P, ST_K, false>(m, wL, wR, wK);
    st_xop<P, ST_M, false>(m, wL, wR, wM);
    st_xop<P, ST_C, true>(m, wL, wR, wCT);
    st_xop<P, ST_C, false>(m, xyL, xyR, C);
    st_xop<P, ST_K, false>(m, xyL, xyR, K);
    st_xop<P, ST_M, false>(m, xyL, xyR, M);
    st_xop<P, ST_C, true>(m, xyL, xyR, CT);
#pragma unroll
    for (int n = 0; n < P; ++n) {
        const float a44wM = p.A44 * wM[n], a44yM = p.A44 * M[n].y;
        G.r1m[n] = p.A55 * (C[n].x + wK[n]);
        G.r1c[n] = a44yM;
        G.r1k[n] = a44wM;
        G.r2m[n] = fmaf(p.D11, K[n].x, p.A55 * (M[n].x + wCT[n]));
        G.r2ct[n] = p.D12 * C[n].y;
        G.r2k[n] = p.D66 * M[n].x;
        G.r2c[n] = p.D66 * CT[n].y;
        G.r3ct[n] = fmaf(p.D66, C[n].x, a44wM);
        G.r3m[n] = fmaf(p.D66, K[n].y, a44yM);
        G.r3c[n] = p.D12 * CT[n].x;
        G.r3k[n] = p.D22 * M[n].y;
    }
}

template <int P, int BASE, int JIN>
__device__ __forceinline__ void st_scatter(st_mats_ptr m, const StGroups<P>& G, float (&acc)[3][2 * P + 1][P]) {
#ifdef DN_ST_ABL_MATH
#pragma unroll
    for (int n = 0; n < P; ++n) { acc[0][BASE][n] += G.r1m[n]; acc[1][BASE + 1][n] += G.r2m[n]; acc[2][BASE][n] -= G.r3m[n]; }
    return;
#endif
    st_yscatter<P, ST_M, false, BASE, JIN>(m, G.r1m, acc[0]);
    st_yscatter<P, ST_C, false, BASE, JIN>(m, G.r1c, acc[0]);
    st_yscatter<P, ST_K, false, BASE, JIN>(m, G.r1k, acc[0]);
    st_yscatter<P, ST_M, false, BASE, JIN>(m, G.r2m, acc[1]);
    st_yscatter<P, ST_C, true, BASE, JIN>(m, G.r2ct, acc[1]);
    st_yscatter<P, ST_K, false, BASE, JIN>(m, G.r2k, acc[1]);
    st_yscatter<P, ST_C, false, BASE, JIN>(m, G.r2c, acc[1]);
    st_yscatter<P, ST_C, true, BASE, JIN>(m, G.r3ct, acc[2]);
    st_yscatter<P, ST_M, false, BASE, JIN>(m, G.r3m, acc[2]);
    st_yscatter<P, ST_C, false, BASE, JIN>(m, G.r3c, acc[2]);
    st_yscatter<P, ST_K, false, BASE, JIN>(m, G.r3k, acc[2]);
}

// MK: Dirichlet mask kind (0 none, 1 uint8, 2 fp32 compared with 0.5); BCF: some boundary value is a field.  Compile-time for the reason
// given in fsdt.hip: no load may sit inside a wave-uniform branch.
template <int P, int MK, bool BCF>
__global__ void __launch_bounds__(256) fsdt2d_st_kernel(const FsdtParams p, const FsdtMats mats_by_value, const int nchunks) {
    constexpr int NB = P + 1, NW = P, NWIN = 2 * P + 1;
    static_assert(sizeof(FsdtParams) % 8 == 0 && alignof(FsdtMats) == 4, "kernel-argument layout: the matrices follow the parameters");
    const st_mats_ptr km = (st_mats_ptr)((const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + sizeof(FsdtParams));
    // the layout assumption, checked on every launch: anything but the magic word there turns every output and sum of the launch into NaN
    const float layout_poison = km->magic == FSDT_MATS_MAGIC ? 0.f : __builtin_nanf("");
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int lane = (int)threadIdx.x & 63;
    const int chunk = (int)blockIdx.x * ((int)blockDim.x >> 6) + wave;      // one wave = one chunk of 62 element columns + two ghost lanes
    const int b = blockIdx.z;
    const int R = p.rows_per_strip;
    const int ey_own = (int)blockIdx.y * R;
    const int ey_begin = ey_own > 0 ? ey_own - 1 : 0;                       // the layer under the strip's first node row is recomputed
    const int ey_end = min(ey_own + R, p.nely);
    const int q = chunk * 62 + lane;
    const int x0 = q * P;
    const bool col_owner = lane != 63 && !(chunk > 0 && lane == 0);       // lanes 0 and 63 are ghosts: they supply their neighbours' halo
    const float okf = q < p.nelx ? 1.f : 0.f;      // the closing thread column (and lanes right of the mesh) have no element to their right
    const float lf = q > 0 ? 1.f : 0.f;            // the first thread column has none to its left
    const int64_t nps = (int64_t)p.nx * p.ny;
    const int ymax = p.ny - 1;

    __shared__ double red[16], den_bc[3];
    __shared__ int last_flag;
    float sq[3] = {0.f, 0.f, 0.f};
    float den3[3] = {0.f, 0.f, 0.f};
    if (p.den_part) den_from_partials(p, (int)threadIdx.x, (int)blockDim.x, red, den_bc, den3);      // consumer of a deferring launch (fsdt_common.h)

    if (chunk < nchunks) {
        const float* fb[3];
        const float* bcf[3];
        bool has_bcf[3];
        float* ob[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            fb[k] = p.fld[k] + (int64_t)b * nps;
            has_bcf[k] = p.bcf[k] != nullptr;
            bcf[k] = has_bcf[k] ? p.bcf[k] + (p.bcf_batched[k] ? (int64_t)b * nps : 0) : fb[k];
            ob[k] = p.out[k] ? p.out[k] + (int64_t)b * nps : nullptr;
        }
        const int64_t mo = p.mask_batched ? (int64_t)b * nps : 0;
        const uint8_t* m8 = reinterpret_cast<const uint8_t*>(p.mask) + (MK == 1 ? mo : 0);
        const float* mf = reinterpret_cast<const float*>(p.mask) + (MK == 2 ? mo : 0);

        float fscale[3] = {1.f, 1.f, 1.f};
        if (p.in_scale) { fscale[0] = p.in_scale[0]; fscale[1] = p.in_scale[1]; fscale[2] = p.in_scale[2]; }
        if (p.in_num) {               // cotangent of the norms over the norms (the VJP of ||R_k||), torch's convention at ||R_k|| == 0: zero
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float den = p.den_part ? den3[k] : p.in_den[k];
                fscale[k] = den > 0.f ? p.in_num[k] / den : (den == den ? 0.f : den);          // (a NaN norm -- stale deferred partials -- stays NaN)
            }
        }

        struct RawRow {              // the thread's OWN nodes only: the node it shares with the right neighbour comes from that lane
            float v[3][NW], bf[3][NW], mfl[NW];
            uint8_t mb[NW];
        };
        auto row_issue = [&](int yr, RawRow& w) {
            const unsigned rowoff = (unsigned)min(yr, ymax) * (unsigned)p.nx;
#pragma unroll
            for (int k = 0; k < 3; ++k) st_load_own<NW>(fb[k], rowoff, x0, p.nx, w.v[k]);
            if constexpr (MK == 1) st_load_own<NW>(m8, rowoff, x0, p.nx, w.mb);
            if constexpr (MK == 2) st_load_own<NW>(mf, rowoff, x0, p.nx, w.mfl);
            if constexpr (BCF && MK != 0) {
#pragma unroll
                for (int k = 0; k < 3; ++k) st_load_own<NW>(bcf[k], rowoff, x0, p.nx, w.bf[k]);
            }
        };
        // landed row: input scaling, Dirichlet nodes (mask >= 0.5) take the boundary values
        auto row_consume = [&](const RawRow& w, float (&cw)[NB], st_v2f (&cxy)[NB], unsigned& bits_out) {
            unsigned bits = 0u;
            if constexpr (MK == 1) {
#pragma unroll
                for (int n = 0; n < NW; ++n) bits |= (w.mb[n] != 0) ? (1u << n) : 0u;
            }
            if constexpr (MK == 2) {
#pragma unroll
                for (int n = 0; n < NW; ++n) bits |= (w.mfl[n] >= 0.5f) ? (1u << n) : 0u;
            }
            bits_out = bits;
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int n = 0; n < NW; ++n) {
                    float v = w.v[k][n] * fscale[k];
                    if constexpr (MK != 0) {
                        float bv = p.bcv[k];
                        if constexpr (BCF) bv = has_bcf[k] ? w.bf[k][n] : bv;
                        v = (bits & (1u << n)) ? bv : v;
                    }
                    if (k == 0) cw[n] = v;
                    else if (k == 1) cxy[n].x = v;
                    else cxy[n].y = v;
                }
            // the node shared with the right neighbour: that lane's first own node, after ITS scaling and substitution (lane 63 is a ghost)
            float t0 = __shfl_down(cw[0], 1, 64), t1 = __shfl_down(cxy[0].x, 1, 64), t2 = __shfl_down(cxy[0].y, 1, 64);
            asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2));
            cw[NW] = t0;
            cxy[NW] = (st_v2f){t1, t2};
        };

        // window of accumulator rows: [0 .. P] the current element layer's node rows, [P + 1 .. 2 P] the rest of the next layer's
        float acc[3][NWIN][P];
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int j = 0; j < NWIN; ++j)
#pragma unroll
                for (int n = 0; n < P; ++n) acc[k][j][n] = 0.f;
        unsigned fixed[NB];                           // Dirichlet bits of the window's rows 0 .. P (own nodes)
        float kv[BCF ? NB : 1][3][P];                 // BCF: their boundary values (= the substituted node values)

        // the load vector's x factor at the own nodes, times -q
        float cq[P];
        cq[0] = -p.q * fmaf(lf, km->lx[P], okf * km->lx[0]);
#pragma unroll
        for (int n = 1; n < P; ++n) cq[n] = -p.q * km->lx[n];

        // finished node rows wait here until flush_rows() stores them (after the next rows have been requested: fsdt.hip)
        float pend[P][3][NW];
        unsigned pend_off[P];
        bool pend_st[P];
#pragma unroll
        for (int r = 0; r < P; ++r) pend_st[r] = false;
        auto flush_rows = [&]() {
#pragma unroll
            for (int r = 0; r < P; ++r) {
                if (pend_st[r]) {
#pragma unroll
                    for (int k = 0; k < 3; ++k)
                        if (ob[k]) st_store_own<NW>(ob[k], pend_off[r], x0, p.nx, pend[r][k]);
                }
                pend_st[r] = false;
            }
        };
        // window row jo is complete: Dirichlet rows of the residual carry the boundary values (e1_plate_bending_fsdt.py:222-228)
        auto emit_row = [&](int jo, int slot, int yr, bool owned_row) {
            const bool st = owned_row && col_owner;
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int n = 0; n < NW; ++n) {
                    float v = acc[k][jo][n] + layout_poison;
                    if constexpr (MK != 0) {
                        float bv = p.bcv[k];
                        if constexpr (BCF) bv = has_bcf[k] ? kv[jo][k][n] : bv;
                        v = (fixed[jo] & (1u << n)) ? bv : v;
                    }
                    sq[k] = (st && x0 + n < p.nx) ? fmaf(v, v, sq[k]) : sq[k];
                    pend[slot][k][n] = v;
                }
            pend_off[slot] = (unsigned)yr * (unsigned)p.nx;
            pend_st[slot] = st;
        };
        auto keep_row = [&](int j, const float (&cw)[NB], const st_v2f (&cxy)[NB], unsigned bits) {
            fixed[j] = bits;
            if constexpr (BCF) {
#pragma unroll
                for (int n = 0; n < P; ++n) { kv[j][0][n] = cw[n]; kv[j][1][n] = cxy[n].x; kv[j][2][n] = cxy[n].y; }
            }
        };

        RawRow W[P];
        {
            RawRow w0;
            row_issue(ey_begin * P, w0);
#pragma unroll
            for (int r = 1; r <= P; ++r) row_issue(ey_begin * P + r, W[r - 1]);       // all P + 1 rows of the first layer in flight together
            float cw[NB];
            st_v2f cxy[NB];
            unsigned bits;
            row_consume(w0, cw, cxy, bits);
            keep_row(0, cw, cxy, bits);
            StGroups<P> G;
            st_stage<P>(p, st_fresh(km), cw, cxy, lf, okf, G);
            st_scatter<P, 0, 0>(st_fresh(km), G, acc);          // the strip's first node row as row 0 of the first layer
        }
        for (int ey = ey_begin; ey < ey_end; ++ey) {
            float cwR[P][NB];
            st_v2f cxyR[P][NB];
            unsigned bitsR[P];
#pragma unroll
            for (int r = 0; r < P; ++r) row_consume(W[r], cwR[r], cxyR[r], bitsR[r]);
#pragma unroll
            for (int r = 1; r <= P; ++r) row_issue((ey + 1) * P + r, W[r - 1]);      // rows beyond the mesh re-read the last one (unused)
            flush_rows();
            if constexpr (P <= 2) {
                // x phase: the layer's P new node rows
                StGroups<P> G[P];
                {
                    const st_mats_ptr mx = st_fresh(km);
#pragma unroll
                    for (int j = 1; j <= P; ++j) {
                        keep_row(j, cwR[j - 1], cxyR[j - 1], bitsR[j - 1]);
                        st_stage<P>(p, mx, cwR[j - 1], cxyR[j - 1], lf, okf, G[j - 1]);
                    }
                }
                // y phase
                {
                    const st_mats_ptr my = st_fresh(km);
#pragma unroll
                    for (int jo = 0; jo <= P; ++jo)
#pragma unroll
                        for (int n = 0; n < P; ++n) acc[0][jo][n] = fmaf(my->ly[jo], cq[n], acc[0][jo][n]);      // - q l_x (x) l_y of this layer
                    st_scatter<P, 0, 1>(my, G[0], acc);
                    if constexpr (P >= 2) st_scatter<P, 0, 2>(my, G[1], acc);
                    if (ey + 1 < p.nely) st_scatter<P, P, 0>(my, G[P - 1], acc);            // the layer's top row is row 0 of the layer above
                }
            } else {
                // Q3: one node row at a time (three rows' groups at once are 99 registers: the build then spills into the accumulation registers)
                {
                    const st_mats_ptr my = st_fresh(km);
#pragma unroll
                    for (int jo = 0; jo <= P; ++jo)
#pragma unroll
                        for (int n = 0; n < P; ++n) acc[0][jo][n] = fmaf(my->ly[jo], cq[n], acc[0][jo][n]);
                }
#pragma unroll
                for (int j = 1; j <= P; ++j) {
                    keep_row(j, cwR[j - 1], cxyR[j - 1], bitsR[j - 1]);
                    StGroups<P> G;
                    st_stage<P>(p, st_fresh(km), cwR[j - 1], cxyR[j - 1], lf, okf, G);
                    const st_mats_ptr my = st_fresh(km);
                    if (j == 1) st_scatter<P, 0, 1>(my, G, acc);
                    if (j == 2) st_scatter<P, 0, 2>(my, G, acc);
                    if constexpr (P >= 3) {
                        if (j == 3) st_scatter<P, 0, 3>(my, G, acc);
                    }
                    if (j == P && ey + 1 < p.nely) st_scatter<P, P, 0>(my, G, acc);
                }
            }
            const bool own_layer = ey >= ey_own;
#pragma unroll
            for (int jo = 0; jo < P; ++jo) emit_row(jo, jo, ey * P + jo, own_layer);
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int n = 0; n < P; ++n) {
#pragma unroll
                    for (int j = 0; j <= P; ++j) acc[k][j][n] = acc[k][j + P][n];
#pragma unroll
                    for (int j = P + 1; j < NWIN; ++j) acc[k][j][n] = 0.f;
                }
            fixed[0] = fixed[P];
            if constexpr (BCF) {
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int n = 0; n < P; ++n) kv[0][k][n] = kv[P][k][n];
            }
        }
        flush_rows();
        if (ey_end == p.nely) {
            emit_row(0, 0, p.ny - 1, true);
            flush_rows();
        }
    }
    if (p.want_sums) {
        if (p.defer_sums) store_partials3(p, sq, (int)threadIdx.x, (int)blockDim.x, red);
        else finish_sums3(p, sq, (int)threadIdx.x, (int)blockDim.x, red, &last_flag);
    }
}

struct FsdtStGeom { int wpb, chunks, gx, R, strips; };

static inline int st_ceil_div(int a, int b) { return (a + b - 1) / b; }

static FsdtStGeom fsdt_st_plan(const dn_mesh* m) {
    FsdtStGeom g;
    const int P = m->degree;
    const int nelx = (m->nx - 1) / P, nely = (m->ny - 1) / P;
    g.chunks = nelx <= 62 ? 1 : st_ceil_div(nelx, 62);       // 62 owner lanes per wave (the first chunk owns 63 thread columns: 0 .. 62)
    // waves per workgroup: the bundle with the fewest idle waves in the last workgroup of a chunk row (ties: the larger)
    g.wpb = 1;
    int best_pad = 1 << 30;
    for (int w = 4; w >= 1; --w) {
        const int pad = st_ceil_div(g.chunks, w) * w - g.chunks;
        if (pad < best_pad) { best_pad = pad; g.wpb = w; }
    }
    // strip height: >= 2304 waves (2.25 per SIMD; 4096 once a row of strips has >= 64 waves) where the mesh has them, at the price of one recomputed
    // layer per strip; 1025^2 Q2 measured (gpurun_out/t2_times_st.txt, t5_, t6_): B = 1: R = 2 / 4 / 8 -> 21.3 / 22.5 / 24.9 us, B = 2: 27.8 / 26.0 / 30.8,
    // B = 4: R = 4 / 8 / 16 -> 46.3 / 37.7 / 48.1, B = 8: R = 8 / 16 -> 59.4 / 61.3
    const long long per_strip = (long long)g.chunks * m->batch;
    const long long want = per_strip >= 64 ? 4096 : 2304;
    int R = 32;
    while (R > 2 && per_strip * st_ceil_div(nely, R) < want) R /= 2;
    const char* e = config(CFG_PLAN_FSDT);      // "T,R": T = 64 x waves per workgroup (tuning experiments only)
    int T, RR;
    if (e && sscanf(e, "%d,%d", &T, &RR) == 2 && T >= 64 && T <= 256 && RR >= 1) { g.wpb = T / 64; R = RR; }
    if (R > nely) R = nely;
    g.R = R < 1 ? 1 : R;
    g.strips = st_ceil_div(nely, g.R);
    g.gx = st_ceil_div(g.chunks, g.wpb);
    return g;
}

int64_t fsdt_st_workgroups(const dn_mesh* m) {
    const FsdtStGeom g = fsdt_st_plan(m);
    return (int64_t)g.gx * g.strips * m->batch;
}

template <int P>
static int fsdt_st_launch_p(const FsdtParams& pp, const FsdtMats& mm, const FsdtStGeom& g, int batch, hipStream_t s) {
    const int mk = !pp.mask ? 0 : (pp.mask_is_u8 ? 1 : 2);
    const bool bcf = mk != 0 && (pp.bcf[0] || pp.bcf[1] || pp.bcf[2]);
    dim3 grid(g.gx, g.strips, batch), block(64 * g.wpb);
    switch (mk * 2 + (bcf ? 1 : 0)) {
        case 0: case 1: hipLaunchKernelGGL((fsdt2d_st_kernel<P, 0, false>), grid, block, 0, s, pp, mm, g.chunks); return 0;
        case 2: hipLaunchKernelGGL((fsdt2d_st_kernel<P, 1, false>), grid, block, 0, s, pp, mm, g.chunks); return 0;
        case 3: hipLaunchKernelGGL((fsdt2d_st_kernel<P, 1, true>), grid, block, 0, s, pp, mm, g.chunks); return 0;
        case 4: hipLaunchKernelGGL((fsdt2d_st_kernel<P, 2, false>), grid, block, 0, s, pp, mm, g.chunks); return 0;
        default: hipLaunchKernelGGL((fsdt2d_st_kernel<P, 2, true>), grid, block, 0, s, pp, mm, g.chunks); return 0;
    }
}

// pp: as filled by dn_fsdt_apply (rows_per_strip is set here); the 1-D matrices are formed from the mesh's tables in double precision
int fsdt_st_launch(const dn_mesh* m, float wscale, FsdtParams& pp, hipStream_t s) {
    const FsdtStGeom g = fsdt_st_plan(m);
    pp.rows_per_strip = g.R;
    FsdtMats mm;
    mm.magic = FSDT_MATS_MAGIC;
    const int nb = m->degree + 1;
    for (int a = 0; a < 4; ++a) {
        double la = 0.0;
        for (int gp = 0; gp < m->ngp; ++gp) la += (a < nb) ? (double)m->gpw[gp] * m->basis[gp][a] : 0.0;
        mm.lx[a] = (float)la;
        mm.ly[a] = (float)(la * wscale);
        for (int b = 0; b < 4; ++b) {
            double M = 0.0, K = 0.0, C = 0.0;
            if (a < nb && b < nb)
                for (int gp = 0; gp < m->ngp; ++gp) {
                    const double w = m->gpw[gp], na = m->basis[gp][a], nbv = m->basis[gp][b], da = m->dbasis[gp][a], db = m->dbasis[gp][b];
                    M += w * na * nbv;
                    K += w * da * db;
                    C += w * da * nbv;
                }
            const double sx = m->scale[0], sy = m->scale[1];
            mm.x[ST_M][a][b] = (float)M;
            mm.x[ST_K][a][b] = (float)(K * sx * sx);
            mm.x[ST_C][a][b] = (float)(C * sx);
            mm.y[ST_M][a][b] = (float)(M * wscale);
            mm.y[ST_K][a][b] = (float)(K * sy * sy * wscale);
            mm.y[ST_C][a][b] = (float)(C * sy * wscale);
        }
    }
    switch (m->degree) {
        case 1: return fsdt_st_launch_p<1>(pp, mm, g, m->batch, s);
        case 2: return fsdt_st_launch_p<2>(pp, mm, g, m->batch, s);
        default: return fsdt_st_launch_p<3>(pp, mm, g, m->batch, s);
    }
}

}  // namespace dn
