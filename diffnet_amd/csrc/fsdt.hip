// Fused first-order shear-deformation (Mindlin) plate residuals on structured Q_P meshes: Dirichlet substitution,
// the nine Gauss-point evaluations of (w, phi_x, phi_y), the constitutive combinations, the three weak-form
// residuals, their element->node assembly, the Dirichlet rows of the result and the three Frobenius sums in ONE pass.
//
// Replaces examples/elasticity/single_instance/e1_plate_bending_fsdt.py:128-232 of the reference (9 gauss_pt_evaluation
// calls = 9 * ngp^2 convolutions, ~40 elementwise ops on (B, nbf, ngp^2, nel) tensors, 3 assemblies); SURVEY.md 8(a)
// row a14, BASELINE.json configs[4] (512 x 512, Q2, 3 x 3 Gauss points).
//
//   Q_x  = A55 (phi_x + w,x)          Q_y  = A44 (phi_y + w,y)                     (A44, A55 carry K_s)
//   M_xx = D11 phi_x,x + D12 phi_y,y  M_yy = D12 phi_x,x + D22 phi_y,y   M_xy = D66 (phi_x,y + phi_y,x)
//   R1_a = sum_g JxW ( N_a,x Q_x  + N_a,y Q_y  - N_a q   )
//   R2_a = sum_g JxW ( N_a,x M_xx + N_a,y M_xy + N_a Q_x )
//   R3_a = sum_g JxW ( N_a,x M_xy + N_a,y M_yy + N_a Q_y )
// The residual is the gradient of the plate energy, so its Jacobian is symmetric: the backward pass is the same kernel
// applied to the masked cotangents with q = 0 and zero Dirichlet values (diffnet_amd/elasticity.py).
//
// Same mapping as the generic fused Poisson kernel (poisson_fused.hip): a thread owns one element column of a strip
// and marches over element rows; node values and partially assembled outputs of the current layer stay in
// registers; the contribution to the node column shared with the right neighbour goes through a double-buffered LDS
// slot; strip / chunk seams are closed by recomputing one layer (no atomics, bitwise repeatable).
#include <algorithm>
#include <cstdlib>

#include "fsdt_common.h"

namespace dn {

// LDS words of the chained strips' hand-overs as inline asm (a `volatile` LDS access makes the compiler drain every outstanding load: see
// poisson2d_q1_cf.hip)
__device__ __forceinline__ unsigned fs_lds_addr(const void* p) { return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p; }
__device__ __forceinline__ void fs_lds_st(unsigned a, float v) { asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(v) : "memory"); }
__device__ __forceinline__ void fs_lds_st(unsigned a, unsigned v) { asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(v) : "memory"); }
__device__ __forceinline__ float fs_lds_ld(unsigned a) {
    float d;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(d) : "v"(a) : "memory");
    return d;
}
__device__ __forceinline__ unsigned fs_lds_ld_u(unsigned a) {
    unsigned d;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(d) : "v"(a) : "memory");
    return d;
}
constexpr unsigned FS_SPIN_MAX = 1u << 20;     // bound of every LDS flag poll: a producer that never arrives must not hang the GPU; reaching it poisons the wave's
                                               // outputs and sums with NaN and sets the workspace's sticky error word (dn_workspace_status) -- see poisson2d_q1_cf.hip
constexpr int FS_CH_MAXW = 12;                 // sub-strips (waves) per chained workgroup: 12 waves = 3 per SIMD at <= 168 VGPRs

// One element: nodal values F[k][jb][ib] of the three fields; its nodal residual contributions are ADDED to g[k][jb][ib]
// (the caller's partially assembled rows: no separate per-element result).
// Sum-factorised one x-Gauss point at a time: x-stage of the three fields for that point (value / x-derivative per node
// row), the NGP y-points with the constitutive law and the y-transpose, then the x-transpose of that point straight into
// g -- the live set is one point's stage values and cotangents (36 registers at Q2) instead of all points' (108).
// (Round 3, measured and removed -- profiles/r3_fsdt_packed_fields.txt: phi_x and phi_y as the halves of packed fp32 registers, 851 -> 700 VALU
// instructions per element, but 166 -> 188 VGPRs plus 84 bytes of scratch per thread in the marching kernel: 27.6 -> 38.9 us at B = 1, 95 -> 156 at B = 8.)
// MID (Q2 with the symmetric 3-point rule, checked on the host): the middle Gauss point sits on the middle node, where the basis is
// (0, 1, 0) and its derivative (-d, 0, +d).  A third of all 1-D contractions then are a copy or one difference instead of three
// fused multiply-adds: 171 of the element's 954 instructions (fmaf(0, x, a) cannot be folded by the compiler -- x may be a NaN --, so
// the zeros are spelled out here; same numbers as the generic form to rounding: d (F2 - F0) has one rounding less than d F2 + (-d) F0).
template <int P, int NGP, bool MID>
__device__ __forceinline__ void fsdt_elem(const FsdtParams& p, const float (&F)[3][P + 1][P + 1], float (&g)[3][P + 1][P + 1]) {
    constexpr int NB = P + 1;
    static_assert(!MID || (P == 2 && NGP == 3), "MID: Q2 elements, 3-point rule");
    const float dxm = p.dx[1][NB - 1], dym = p.dy[1][NB - 1];       // derivative of the last basis function at the middle point
#pragma unroll
    for (int ig = 0; ig < NGP; ++ig) {
        const bool xm = MID && ig == 1;
        float tv[3][NB], td[3][NB], rv[3][NB], rd[3][NB];
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int jb = 0; jb < NB; ++jb) {
                float a = 0.f, d = 0.f;
                if (xm) {
                    a = F[k][jb][1];
                    d = dxm * (F[k][jb][2] - F[k][jb][0]);
                } else {
#pragma unroll
                    for (int ib = 0; ib < NB; ++ib) {
                        a = fmaf(p.b[ig][ib], F[k][jb][ib], a);
                        d = fmaf(p.dx[ig][ib], F[k][jb][ib], d);
                    }
                }
                tv[k][jb] = a; td[k][jb] = d; rv[k][jb] = 0.f; rd[k][jb] = 0.f;
            }
#pragma unroll
        for (int jg = 0; jg < NGP; ++jg) {
            const bool ym = MID && jg == 1;
            float val[3], fx[3], fy[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float v = 0.f, x = 0.f, y = 0.f;
                if (ym) {
                    v = tv[k][1];
                    x = td[k][1];
                    y = dym * (tv[k][2] - tv[k][0]);
                } else {
#pragma unroll
                    for (int jb = 0; jb < NB; ++jb) {
                        v = fmaf(p.b[jg][jb], tv[k][jb], v);
                        x = fmaf(p.b[jg][jb], td[k][jb], x);
                        y = fmaf(p.dy[jg][jb], tv[k][jb], y);
                    }
                }
                val[k] = v; fx[k] = x; fy[k] = y;
            }
            const float W = p.w2[jg][ig];
            const float Qx = W * (p.A55 * (val[1] + fx[0])), Qy = W * (p.A44 * (val[2] + fy[0]));
            const float Mxx = W * fmaf(p.D11, fx[1], p.D12 * fy[2]);
            const float Myy = W * fmaf(p.D12, fx[1], p.D22 * fy[2]);
            const float Mxy = W * (p.D66 * (fy[1] + fx[2]));
            const float cv[3] = {-p.q * W, Qx, Qy}, cx[3] = {Qx, Mxx, Mxy}, cy[3] = {Qy, Mxy, Myy};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (ym) {
                    const float t = dym * cy[k];
                    rv[k][0] -= t;
                    rv[k][1] += cv[k];
                    rv[k][2] += t;
                    rd[k][1] += cx[k];
                } else {
#pragma unroll
                    for (int jb = 0; jb < NB; ++jb) {
                        rv[k][jb] = fmaf(p.b[jg][jb], cv[k], rv[k][jb]);
                        rv[k][jb] = fmaf(p.dy[jg][jb], cy[k], rv[k][jb]);
                        rd[k][jb] = fmaf(p.b[jg][jb], cx[k], rd[k][jb]);
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int jb = 0; jb < NB; ++jb) {
                if (xm) {
                    const float t = dxm * rd[k][jb];
                    g[k][jb][0] -= t;
                    g[k][jb][1] += rv[k][jb];
                    g[k][jb][2] += t;
                } else {
#pragma unroll
                    for (int ib = 0; ib < NB; ++ib) {
                        g[k][jb][ib] = fmaf(p.b[ig][ib], rv[k][jb], g[k][jb][ib]);
                        g[k][jb][ib] = fmaf(p.dx[ig][ib], rd[k][jb], g[k][jb][ib]);
                    }
                }
            }
    }
}

// grid = (chunks_x, strips_y, B), block = T threads; one element column per thread.
// MK: Dirichlet mask kind (0 none, 1 uint8, 2 fp32 compared with 0.5); BCF: some boundary value is a field (then all three
// slots are loaded, an absent one re-reads its own field and is ignored).  Both are compile-time so that NO load sits inside a
// wave-uniform branch: the compiler waits vmcnt(0) where such a branch joins, which serialised the three field loads, the
// mask load and the boundary-field loads of every row (three memory latencies per row instead of one).
// The P new node rows of layer k + 1 are requested before the arithmetic of layer k (software pipeline: `W`), and the
// finished rows of layer k are stored after that request (a store issued first would be younger than the loads the next
// consumer waits for; sitting in a divergent branch it would turn that wait into vmcnt(0)).
// CH (chained strips): the workgroup holds W = blockDim.x / 64 sub-strips of ONE wave each (64 element columns), neighbouring strips of
// one column of chunks.  A sub-strip does not recompute the layer under its first node row: it publishes that row (values after the
// Dirichlet substitution) in LDS for the last layer of the sub-strip below, which answers with that layer's contributions to the row.
// At one sample the element rows have to be cut into strips of 2 to fill the chip -- un-chained that is one recomputed layer per two
// (+50 % arithmetic in a kernel bound by its 918 instructions per element); chained it is one per workgroup.  Inside a one-wave
// sub-strip the hand-over to the right neighbour is a lane shuffle: no LDS slot, no barrier.
template <int P, int NGP, int MK, bool BCF, bool CH, bool MID>
__global__ void __launch_bounds__(CH ? 64 * FS_CH_MAXW : 256) fsdt2d_kernel(const FsdtParams p) {
    constexpr int NB = P + 1;
    constexpr int NW = P;                  // nodes owned per thread per node row
    const int T = CH ? 64 : (int)blockDim.x;
    const int sub = CH ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) : 0;
    const int nsub = CH ? (int)blockDim.x >> 6 : 1;
    const int tid = CH ? (int)threadIdx.x & 63 : (int)threadIdx.x;
    const int chunk = blockIdx.x, b = blockIdx.z;
    const int R = p.rows_per_strip;
    // Rows of this (sub-)strip.  Un-chained: strips of R element rows, each recomputes the layer under its first row.  Chained: the
    // workgroup covers nsub * R - 1 rows; sub-strip 0 -- the one that recomputes the seam layer under the workgroup -- owns R - 1 of
    // them, the others R, so that EVERY wave of the workgroup marches R layers (a workgroup is as slow as its slowest wave, and a wave
    // left alone on its SIMD issues at half the rate: with equal strips the chained launch was slower than the un-chained one)
    const int rows_wg = nsub * R - 1;
    const int ey_own = CH ? (int)blockIdx.y * rows_wg + (sub == 0 ? 0 : R - 1 + (sub - 1) * R) : (int)blockIdx.y * R;
    const int own_rows = (CH && sub == 0) ? R - 1 : R;
    const bool active = !CH || sub == 0 || ey_own < p.nely;
    const bool chain_dn = CH && sub > 0, chain_up = CH && sub + 1 < nsub && ey_own + own_rows < p.nely;
    const int q = chunk * (T - 1) + tid;   // chunks overlap by one thread column
    const int ex0 = q, x0 = ex0 * P;
    const bool col_owner = !(chunk > 0 && tid == 0);
    const int64_t nps = (int64_t)p.nx * p.ny;
    const int ey_begin = (ey_own > 0 && !chain_dn) ? ey_own - 1 : ey_own;     // chain_dn: no seam layer
    const int ey_end = min(ey_own + own_rows, p.nely);
    const int ymax = chain_up ? max(ey_end * P - 1, 0) : p.ny - 1;     // chain_up: the top node row comes from the sub-strip above, not from HBM
    const float okf = (ex0 < p.nelx) ? 1.f : 0.f;      // threads right of the mesh compute on clamped data, scaled by 0

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

    __shared__ double den_red[3 * 12], den_bc[3];
    float den3[3] = {0.f, 0.f, 0.f};
    if (p.den_part) den_from_partials(p, (int)threadIdx.x, (int)blockDim.x, den_red, den_bc, den3);      // consumer of a deferring launch (fsdt_common.h)
    float fscale[3] = {1.f, 1.f, 1.f};
    if (p.in_scale) { fscale[0] = p.in_scale[0]; fscale[1] = p.in_scale[1]; fscale[2] = p.in_scale[2]; }
    if (p.in_num) {               // cotangent of the norms over the norms (the VJP of ||R_k||), torch's convention at ||R_k|| == 0: zero
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float den = p.den_part ? den3[k] : p.in_den[k];
            fscale[k] = den > 0.f ? p.in_num[k] / den : (den == den ? 0.f : den);          // (a NaN norm -- stale deferred partials -- stays NaN)
        }
    }

    __shared__ float xch[CH ? 1 : 2][3][P][CH ? 1 : 256];
    __shared__ double red[16];
    __shared__ int last_flag;
    // CH: per seam (sub-strip s | s + 1) SEAM_WORDS x 64 floats + 2 flags, in dynamic LDS: [0 .. 3 NB) the upper strip's first node row,
    // [3 NB] its Dirichlet bits, [3 NB + 1 .. 6 NB + 1) the lower strip's contributions to that row, [6 NB + 1 .. 9 NB + 1) the upper
    // strip's own (parked) contributions
    constexpr int SEAM_WORDS = 9 * NB + 1;
    extern __shared__ float fs_dyn[];
    float* seam_base = fs_dyn;                                             // [nsub - 1][SEAM_WORDS][64]
    unsigned* flag_base = reinterpret_cast<unsigned*>(fs_dyn + (size_t)(nsub > 1 ? nsub - 1 : 0) * SEAM_WORDS * 64);       // [nsub - 1][2]
    if constexpr (CH) {
        if ((int)threadIdx.x < 2 * (nsub - 1)) flag_base[threadIdx.x] = 0u;
        __syncthreads();
    }
    float spin_poison = 0.f;          // NaN once a poll has run into its bound: everything the wave writes afterwards, its sums included, is NaN
    auto spin_until = [&](const unsigned* flag) {
        const unsigned fa = fs_lds_addr(flag);
        const unsigned lim = p.spin_limit > 0 ? (unsigned)p.spin_limit : FS_SPIN_MAX;
        unsigned n = 0;
        for (; n < lim; ++n) {
            if (__builtin_amdgcn_readfirstlane((int)fs_lds_ld_u(fa)) != 0) break;
            __builtin_amdgcn_s_sleep(1);
        }
        if (n == lim) spin_poison = __builtin_nanf("");
    };

    float cu[3][NB][NW + 1], acc[3][NB][NW + 1];
    unsigned fixed[NB];
#pragma unroll
    for (int r = 0; r < NB; ++r) {
        fixed[r] = 0u;
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int n = 0; n <= NW; ++n) acc[k][r][n] = 0.f;
    }

    struct RawRow {
        float v[3][NW + 1], bf[3][NW + 1], mfl[NW + 1];
        uint8_t mb[NW + 1];
    };
    auto row_issue = [&](int yr, RawRow& w) {
        const unsigned rowoff = (unsigned)min(yr, ymax) * (unsigned)p.nx;
#pragma unroll
        for (int k = 0; k < 3; ++k) load_seg<NW, false>(fb[k], rowoff, x0, p.nx, w.v[k]);
        if constexpr (MK == 1) load_seg<NW, false>(m8, rowoff, x0, p.nx, w.mb);
        if constexpr (MK == 2) load_seg<NW, false>(mf, rowoff, x0, p.nx, w.mfl);
        if constexpr (BCF && MK != 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) load_seg<NW, false>(bcf[k], rowoff, x0, p.nx, w.bf[k]);
        }
    };
    // landed row -> slot r: input scaling, Dirichlet nodes (mask >= 0.5) take the boundary values
    auto row_consume = [&](const RawRow& w, int r) {
        unsigned bits = 0u;
        if constexpr (MK == 1) {
#pragma unroll
            for (int n = 0; n <= NW; ++n) bits |= (w.mb[n] != 0) ? (1u << n) : 0u;
        }
        if constexpr (MK == 2) {
#pragma unroll
            for (int n = 0; n <= NW; ++n) bits |= (w.mfl[n] >= 0.5f) ? (1u << n) : 0u;
        }
        fixed[r] = bits;
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int n = 0; n <= NW; ++n) {
                float v = w.v[k][n] * fscale[k];
                if constexpr (MK != 0) {
                    float bv = p.bcv[k];
                    if constexpr (BCF) bv = has_bcf[k] ? w.bf[k][n] : bv;
                    v = (bits & (1u << n)) ? bv : v;
                }
                cu[k][r][n] = v;
            }
    };

    float sq[3] = {0.f, 0.f, 0.f};
    int par = 0;

    // finished node rows wait here until flush_rows() stores them (see the header)
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
                    if (ob[k]) store_seg<NW, false>(ob[k], pend_off[r], x0, p.nx, pend[r][k]);
            }
            pend_st[r] = false;
        }
    };

    // Emit node row yr from acc[.][r] (+ the left neighbour's hand-over for n == 0); Dirichlet rows of the residual carry
    // the boundary values (e1_plate_bending_fsdt.py:222-228), which cu holds at those nodes.
    auto emit_row = [&](int r, int slot, int yr, bool owned_row) {
        float lefts[3];
        if constexpr (CH) {               // one wave per sub-strip: the neighbour is the lane to the left
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float t = __shfl_up(acc[k][r][NW], 1, 64);
                asm volatile("" : "+v"(t));          // keep the exchange out of the select below (every lane must take part)
                lefts[k] = tid > 0 ? t : 0.f;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) xch[par][k][r % P][tid] = acc[k][r][NW];
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // LDS-only barrier (loads stay in flight)
#pragma unroll
            for (int k = 0; k < 3; ++k) lefts[k] = (tid > 0) ? xch[par][k][r % P][tid - 1] : 0.f;
        }
        const bool st = owned_row && col_owner;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float left = lefts[k];
#pragma unroll
            for (int n = 0; n < NW; ++n) {
                float v = acc[k][r][n] + (n == 0 ? left : 0.f);
                v = (fixed[r] & (1u << n)) ? cu[k][r][n] : v;
                if constexpr (CH) v += spin_poison;
                sq[k] = (st && x0 + n < p.nx) ? fmaf(v, v, sq[k]) : sq[k];
                pend[slot][k][n] = v;
            }
        }
        pend_off[slot] = (unsigned)yr * (unsigned)p.nx;
        pend_st[slot] = st;
    };

    if (active) {                     // (a chained workgroup's unused sub-strips only join the final reduction)
    float* seam_dn = seam_base + (size_t)(sub > 0 ? sub - 1 : 0) * SEAM_WORDS * 64;       // seam with the sub-strip below / above
    float* seam_up = seam_base + (size_t)sub * SEAM_WORDS * 64;
    RawRow W[P];
    {
        RawRow w0;
        row_issue(ey_begin * P, w0);
#pragma unroll
        for (int r = 1; r <= P; ++r) row_issue(ey_begin * P + r, W[r - 1]);       // all P + 1 rows of the first layer in flight together
        row_consume(w0, 0);
    }
    if constexpr (CH) {
        if (chain_dn) {               // publish the strip's first node row for the last layer of the strip below
            const unsigned s0 = fs_lds_addr(seam_dn + tid);
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int n = 0; n <= NW; ++n) fs_lds_st(s0 + (k * NB + n) * 256u, cu[k][0][n]);
            fs_lds_st(s0 + (3 * NB) * 256u, fixed[0]);
            if (tid == 0) fs_lds_st(fs_lds_addr(flag_base + 2 * (sub - 1)), 1u);       // LDS executes a wave's accesses in order
        }
    }
    for (int ey = ey_begin; ey < ey_end; ++ey) {
#pragma unroll
        for (int r = 1; r <= P; ++r) row_consume(W[r - 1], r);
#pragma unroll
        for (int r = 1; r <= P; ++r) row_issue((ey + 1) * P + r, W[r - 1]);      // rows beyond the mesh re-read the last one (unused)
        flush_rows();
        if constexpr (CH) {
            if (chain_up && ey == ey_end - 1) {       // the strip's last layer: its top node row is the first row of the strip above
                spin_until(flag_base + 2 * sub);
                const unsigned s0 = fs_lds_addr(seam_up + tid);
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int n = 0; n <= NW; ++n) cu[k][P][n] = fs_lds_ld(s0 + (k * NB + n) * 256u);
                fixed[P] = fs_lds_ld_u(s0 + (3 * NB) * 256u);
            }
        }
        const bool own_layer = ey >= ey_own;
        {
            static_assert(NW + 1 == NB, "one element per thread: the row state is the element's node block");
            float g[3][NB][NB];
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int jb = 0; jb < NB; ++jb)
#pragma unroll
                    for (int ib = 0; ib < NB; ++ib) g[k][jb][ib] = 0.f;
            fsdt_elem<P, NGP, MID>(p, cu, g);
#ifdef DN_FSDT_TWICE                  // diagnostic (tools/variant_build.sh): the element arithmetic a second time -- is the launch bound by it?
            {
                float cu2[3][NB][NB];
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int jb = 0; jb < NB; ++jb)
#pragma unroll
                        for (int ib = 0; ib < NB; ++ib) { cu2[k][jb][ib] = cu[k][jb][ib] + g[k][jb][ib]; asm volatile("" : "+v"(cu2[k][jb][ib])); }
                float g2[3][NB][NB] = {};
                fsdt_elem<P, NGP, MID>(p, cu2, g2);
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int jb = 0; jb < NB; ++jb)
#pragma unroll
                        for (int ib = 0; ib < NB; ++ib) g[k][jb][ib] = fmaf(1e-30f, g2[k][jb][ib], g[k][jb][ib]);
            }
#endif
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int jb = 0; jb < NB; ++jb)
#pragma unroll
                    for (int ib = 0; ib < NB; ++ib) acc[k][jb][ib] = fmaf(okf, g[k][jb][ib], acc[k][jb][ib]);
        }
        if (CH && chain_dn && ey == ey_own) {
            // the strip's first node row still lacks the contributions of the layer below it (the strip below computes them at its very
            // end): park this half, the node values and the Dirichlet bits in LDS and finish the row after the march
            const unsigned s0 = fs_lds_addr(seam_dn + tid);
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int n = 0; n <= NW; ++n) {
                    fs_lds_st(s0 + (6 * NB + 1 + k * NB + n) * 256u, acc[k][0][n]);      // (node values and bits: as published at the start)
                }
#pragma unroll
            for (int r = 1; r < P; ++r) emit_row(r, r, ey * P + r, own_layer);
        } else {
#pragma unroll
            for (int r = 0; r < P; ++r) emit_row(r, r, ey * P + r, own_layer);
        }
        par ^= 1;
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int n = 0; n <= NW; ++n) {
                cu[k][0][n] = cu[k][P][n];
                acc[k][0][n] = acc[k][P][n];
#pragma unroll
                for (int r = 1; r <= P; ++r) acc[k][r][n] = 0.f;
            }
        fixed[0] = fixed[P];
    }
    flush_rows();
    if constexpr (CH) {
        if (chain_up) {               // this strip's contributions to the first row of the strip above
            const unsigned s0 = fs_lds_addr(seam_up + tid);
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int n = 0; n <= NW; ++n) fs_lds_st(s0 + (3 * NB + 1 + k * NB + n) * 256u, acc[k][0][n]);
            if (tid == 0) fs_lds_st(fs_lds_addr(flag_base + 2 * sub + 1), 1u);
        }
    }
    if (ey_end == p.nely) {
        emit_row(0, 0, p.ny - 1, true);
        flush_rows();
    }
    if constexpr (CH) {
        if (chain_dn) {               // finish the strip's first node row: parked half + the carry of the strip below
            spin_until(flag_base + 2 * (sub - 1) + 1);
            const unsigned s0 = fs_lds_addr(seam_dn + tid);
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int n = 0; n <= NW; ++n) {
                    acc[k][0][n] = fs_lds_ld(s0 + (6 * NB + 1 + k * NB + n) * 256u) + fs_lds_ld(s0 + (3 * NB + 1 + k * NB + n) * 256u);
                    cu[k][0][n] = fs_lds_ld(s0 + (k * NB + n) * 256u);
                }
            fixed[0] = fs_lds_ld_u(s0 + (3 * NB) * 256u);
            emit_row(0, 0, ey_own * P, true);
            flush_rows();
        }
    }
    }                                 // active

    if constexpr (CH) {               // a hand-over poll ran into its bound: NaN sums + the sticky error word of the workspace (never silent)
#pragma unroll
        for (int k = 0; k < 3; ++k) sq[k] += spin_poison;
        if (spin_poison != spin_poison && p.counter != nullptr && (threadIdx.x & 63u) == 0u) atomicOr(p.counter + 8, 1u);      // DN_WS_ERRWORD
    }
    if (p.want_sums) {
        if (p.defer_sums) store_partials3(p, sq, (int)threadIdx.x, (int)blockDim.x, red);
        else finish_sums3(p, sq, (int)threadIdx.x, (int)blockDim.x, red, &last_flag);
    }
}

static inline int fs_ceil_div(int a, int b) { return (a + b - 1) / b; }
static constexpr int64_t FSDT_WS_HEADER = 64 * (1 + 64);

struct FsdtGeom { int T, chunks, R, strips, W; };     // W > 1: chained launch, W one-wave sub-strips per workgroup (T == 64); strips = workgroups per column of chunks

// Cost model of a launch plan, in wave-layers on the busiest SIMD (the kernel is bound by its arithmetic: ~918 VALU instructions per
// element layer at Q2 / 3 x 3; a SIMD issues them at ~2.4 cycles each once two waves share it, ~4.7 with a single wave): workgroups
// are dealt over 256 CUs, a CU holds 12 waves (<= 168 VGPRs), its busiest SIMD a quarter of the CU's wave-layers.
static double fsdt_cost(long long nwg, int wg_waves, double wg_wave_layers) {
    const long long per_cu = (nwg + 255) / 256;
    const long long resident = std::min<long long>(std::max(1, 12 / wg_waves), per_cu);
    const double waves_per_simd = (double)(resident * wg_waves) / 4.0;
    return (double)per_cu * wg_wave_layers / 4.0 * (waves_per_simd < 2.0 ? 2.0 / std::max(1.0, waves_per_simd) : 1.0);
}

static FsdtGeom fsdt_plan(const dn_mesh* m, bool allow_chain = true) {
    FsdtGeom g;
    const int P = m->degree;
    const int Q = (m->nx - 1) / P + 1;          // logical thread columns (one per element + the closing column)
    const int nely = (m->ny - 1) / P;
    double best = -1.0;
    g.T = 64; g.chunks = 1; g.W = 1;
    for (int T = 64; T <= 256; T += 64) {
        const int chunks = Q <= T ? 1 : fs_ceil_div(Q - 1, T - 1);
        const double util = (double)Q / ((double)chunks * T);
        // at equal utilisation wider workgroups win (fewer recomputed chunk-seam columns, fewer workgroup dispatches:
        // 1025^2 Q2 B = 8: 285 us with 192 threads x 8 rows, 324 us with 64 x 16 -- profiles/r1_configs.txt)
        const double score = util + 0.0003 * T;
        if (score > best) { best = score; g.T = T; g.chunks = chunks; }
    }
    // strip height: enough WAVES for ~4 per SIMD (the element is a long dependent chain), at the price of one recomputed
    // layer per strip
    const long long per_strip = (long long)g.chunks * m->batch * (g.T / 64);
    int R = 32;
    while (R > 4 && per_strip * fs_ceil_div(nely, R) < 4096) R /= 2;
    // fewer than 2 waves per SIMD even at 4 rows (1025^2 Q2, one sample: 1152 waves): 2-row strips -- a SIMD with a single wave issues
    // a VALU instruction every ~4.7 cycles, with two every ~2.4, which outweighs the second recomputed layer (32.7 -> 28.9 us;
    // from 2 samples on 4 rows win: profiles/r2_fsdt_plans_steady.txt)
    if (R == 4 && nely >= 2 && per_strip * fs_ceil_div(nely, 4) < 2048) R = 2;
    if (R > nely) R = nely;
    g.R = R < 1 ? 1 : R;
    g.strips = fs_ceil_div(nely, g.R);
    // Chained alternative (fsdt2d_kernel<.., CH>, "PLAN_FSDT" "64,R,W"): W one-wave sub-strips per workgroup recompute ONE seam layer per
    // workgroup instead of one per strip -- a quarter less arithmetic at one sample.  Measured (profiles/r3_fsdt_plans.txt), it is not
    // faster: at B = 1 the launch is bound by the latency chain of a wave's 2-3 layers, not by the arithmetic (28.9 us either way), at
    // B = 8 it won 5 % before the middle-point form of the element and loses 20 % with it (the chained instantiation is capped at 168
    // VGPRs by its 12-wave workgroups and spills).  Only chain lengths that fill the SIMDs evenly (4, 10, 12) are usable at all: a
    // workgroup's waves go to the SIMDs in turn, and a second workgroup is not placed when one SIMD would exceed its register file.
    // Kept as a plan option, not chosen by the library.
    (void)fsdt_cost;
    const char* e = config(CFG_PLAN_FSDT);      // "T,R[,W]" (tuning experiments only; W >= 2: chained, T is then 64)
    int T, RR, W = 1;
    if (e && sscanf(e, "%d,%d,%d", &T, &RR, &W) >= 2 && T >= 64 && T <= 256 && RR >= 1) {
        g.W = (W >= 2 && W <= FS_CH_MAXW && allow_chain) ? W : 1;
        if (g.W > 1) T = 64;
        g.T = T; g.R = RR > nely ? nely : RR;
        g.chunks = Q <= T ? 1 : fs_ceil_div(Q - 1, T - 1);
        g.strips = g.W > 1 ? fs_ceil_div(nely, g.W * g.R - 1) : fs_ceil_div(nely, g.R);
    }
    return g;
}

static int fsdt_validate(const dn_mesh* m) {
    if (!m || m->nsd != 2) return DN_E_BADARG;
    if (m->degree < 1 || m->degree > 3 || m->ngp < 2 || m->ngp > 4) return DN_E_UNSUPPORTED;
    if (m->batch < 1 || m->batch > 65535 || m->nx < 2 || m->ny < 2) return DN_E_BADARG;
    if ((m->nx - 1) % m->degree || (m->ny - 1) % m->degree) return DN_E_BADARG;
    if ((int64_t)m->nx * m->ny >= (1ll << 30)) return DN_E_UNSUPPORTED;
    return 0;
}

template <int P, int NGP, bool MID>
static int fsdt_launch_mk(const FsdtParams& pp, const FsdtGeom& g, int batch, hipStream_t s) {
    const int mk = !pp.mask ? 0 : (pp.mask_is_u8 ? 1 : 2);
    const bool bcf = mk != 0 && (pp.bcf[0] || pp.bcf[1] || pp.bcf[2]);
    if (g.W > 1) {                    // chained sub-strips: one wave each, seams through dynamic LDS
        if (bcf || g.T != 64 || g.W > FS_CH_MAXW) return DN_E_BADARG;
        dim3 grid(g.chunks, g.strips, batch), block(64 * g.W);
        const size_t lds = (size_t)(g.W - 1) * ((9 * (P + 1) + 1) * 64 * sizeof(float) + 2 * sizeof(unsigned));
        switch (mk) {
            case 0: hipLaunchKernelGGL((fsdt2d_kernel<P, NGP, 0, false, true, MID>), grid, block, lds, s, pp); return 0;
            case 1: hipLaunchKernelGGL((fsdt2d_kernel<P, NGP, 1, false, true, MID>), grid, block, lds, s, pp); return 0;
            default: hipLaunchKernelGGL((fsdt2d_kernel<P, NGP, 2, false, true, MID>), grid, block, lds, s, pp); return 0;
        }
    }
    dim3 grid(g.chunks, g.strips, batch), block(g.T);
    switch (mk * 2 + (bcf ? 1 : 0)) {
        case 0: case 1: hipLaunchKernelGGL((fsdt2d_kernel<P, NGP, 0, false, false, MID>), grid, block, 0, s, pp); return 0;
        case 2: hipLaunchKernelGGL((fsdt2d_kernel<P, NGP, 1, false, false, MID>), grid, block, 0, s, pp); return 0;
        case 3: hipLaunchKernelGGL((fsdt2d_kernel<P, NGP, 1, true, false, MID>), grid, block, 0, s, pp); return 0;
        case 4: hipLaunchKernelGGL((fsdt2d_kernel<P, NGP, 2, false, false, MID>), grid, block, 0, s, pp); return 0;
        default: hipLaunchKernelGGL((fsdt2d_kernel<P, NGP, 2, true, false, MID>), grid, block, 0, s, pp); return 0;
    }
}

template <int P>
static int fsdt_launch(const FsdtParams& pp, const FsdtGeom& g, int ngp, int batch, bool mid, hipStream_t s) {
    switch (ngp) {
        case 2: return fsdt_launch_mk<P, 2, false>(pp, g, batch, s);
        case 3:
            if constexpr (P == 2) {
                if (mid) return fsdt_launch_mk<P, 3, true>(pp, g, batch, s);
            }
            return fsdt_launch_mk<P, 3, false>(pp, g, batch, s);
        case 4: return fsdt_launch_mk<P, 4, false>(pp, g, batch, s);
        default: return DN_E_UNSUPPORTED;
    }
}

}  // namespace dn

using namespace dn;

extern "C" int64_t dn_fsdt_workspace_bytes(const dn_mesh* m) {
    if (fsdt_validate(m) != 0) return DN_E_BADARG;
    const FsdtGeom g = fsdt_plan(m, false), gc = fsdt_plan(m, true);         // either plan fits (un-chained has the most workgroups per strip)
    const int64_t n1 = (int64_t)g.chunks * g.strips * m->batch, n2 = (int64_t)gc.chunks * gc.strips * m->batch;
    return FSDT_WS_HEADER + (int64_t)(3 * sizeof(double)) * std::max(std::max(n1, n2), fsdt_st_workgroups(m));
}

extern "C" int dn_fsdt_apply(const dn_mesh* m, const dn_fsdt_args* a, void* stream) {
    int rc = fsdt_validate(m);
    if (rc) return rc;
    if (!a || !a->w || !a->phi_x || !a->phi_y) return DN_E_BADARG;
    if (!a->out[0] && !a->out[1] && !a->out[2] && !a->sumsq && !a->norms && !a->defer_sums) return DN_E_BADARG;
    // in_num goes with in_den or, round 4, with den_workspace (the norms formed from a deferring launch's partials), not with both, not with in_scale
    if (a->in_num ? ((a->in_den != nullptr) == (a->den_workspace != nullptr) || a->in_scale != nullptr) : (a->in_den != nullptr || a->den_workspace != nullptr))
        return DN_E_BADARG;
    if (a->defer_sums && a->den_workspace) return DN_E_BADARG;          // (sumsq / norms of a consuming call are the PRODUCER's)
    const bool want_red = a->defer_sums || (!a->den_workspace && (a->sumsq || a->norms));
    const bool any_bcf = a->bc_mask && (a->bc_field[0] || a->bc_field[1] || a->bc_field[2]);
    const FsdtGeom g = fsdt_plan(m, !any_bcf);
    // round 4: the assembled-stencil form (fsdt_st.hip) is the default; dn_config_set("FSDT_FORM", "elem"), "FSDT_GENERIC" or a chained
    // launch plan ("PLAN_FSDT" "64,R,W") keep the element form of rounds 1-3
    const char* form = config(CFG_FSDT_FORM);
    // Q3 stays on the element form unless "FSDT_FORM" = "stencil" asks for the other: measured slower there (766^2 Q3: B = 8 69.0 vs 55.2 us, B = 4 47.1 vs
    // 36.7 -- 226-244 registers, one node row per phase; profiles/r4_fsdt_stencil.txt section 7), and its instantiation with Dirichlet value FIELDS needs 314
    // registers, spills into the accumulation registers and came out wrong on the GPU (never run: section 6).  Q1 and Q2 are faster in the stencil form at
    // every size measured (1024^2 Q1 B = 8: 66 vs 109 us).
    const bool ask_st = form && form[0] == 's';
    const bool stencil = !(form && form[0] == 'e') && config(CFG_FSDT_GENERIC) == nullptr && g.W == 1 && (m->degree <= 2 || (ask_st && !any_bcf));
    const int64_t nwg = stencil ? fsdt_st_workgroups(m) : (int64_t)g.chunks * g.strips * m->batch;
    if (want_red && (!a->workspace || a->workspace_bytes < FSDT_WS_HEADER + (int64_t)(3 * sizeof(double)) * nwg)) return DN_E_WORKSPACE;

    FsdtParams pp;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            pp.b[i][j] = m->basis[i][j];
            pp.dx[i][j] = m->dbasis[i][j] * m->scale[0];
            pp.dy[i][j] = m->dbasis[i][j] * m->scale[1];
            pp.w2[i][j] = m->gpw[i] * (m->gpw[j] * a->wscale);
        }
    pp.D11 = a->D11; pp.D12 = a->D12; pp.D22 = a->D22; pp.D66 = a->D66; pp.A44 = a->A44; pp.A55 = a->A55; pp.q = a->q;
    pp.fld[0] = a->w; pp.fld[1] = a->phi_x; pp.fld[2] = a->phi_y;
    pp.in_scale = a->in_scale;
    pp.in_num = a->in_num; pp.in_den = a->in_den; pp.norms = a->norms;
    pp.mask = a->bc_mask; pp.mask_is_u8 = a->mask_is_u8; pp.mask_batched = a->mask_batched;
    for (int k = 0; k < 3; ++k) {
        pp.bcf[k] = a->bc_field[k]; pp.bcf_batched[k] = a->bc_field_batched[k]; pp.bcv[k] = a->bc_value[k];
        pp.out[k] = a->out[k];
    }
    pp.counter = reinterpret_cast<unsigned*>(a->workspace);
    pp.part = a->workspace ? reinterpret_cast<double*>(reinterpret_cast<char*>(a->workspace) + FSDT_WS_HEADER) : nullptr;
    pp.sumsq = a->sumsq;
    pp.nx = m->nx; pp.ny = m->ny;
    pp.nelx = (m->nx - 1) / m->degree; pp.nely = (m->ny - 1) / m->degree;
    pp.rows_per_strip = g.R;
    pp.want_sums = want_red ? 1 : 0;
    pp.defer_sums = a->defer_sums;
    pp.den_ticket = a->den_ticket;
    pp.den_counter = reinterpret_cast<const unsigned*>(a->den_workspace);
    pp.den_part = a->den_workspace ? reinterpret_cast<const double*>(reinterpret_cast<const char*>(a->den_workspace) + FSDT_WS_HEADER) : nullptr;
    pp.spin_limit = config(CFG_HANDOVER_SPIN_LIMIT) ? std::atoi(config(CFG_HANDOVER_SPIN_LIMIT)) : 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (stencil) {
        rc = fsdt_st_launch(m, a->wscale, pp, s);
        if (rc) return rc;
        DN_LAUNCH_CHECK();
        return 0;
    }
    switch (m->degree) {
        case 1: rc = fsdt_launch<1>(pp, g, m->ngp, m->batch, false, s); break;
        case 2: {
            // the symmetric 3-point rule puts its middle point on the middle node of a Q2 element: basis (0, 1, 0), derivative (-d, 0, d)
            const bool mid = m->ngp == 3 && m->basis[1][0] == 0.f && m->basis[1][2] == 0.f && m->basis[1][1] == 1.f && m->dbasis[1][1] == 0.f &&
                             m->dbasis[1][0] == -m->dbasis[1][2] && config(CFG_FSDT_GENERIC) == nullptr;
            rc = fsdt_launch<2>(pp, g, m->ngp, m->batch, mid, s);
            break;
        }
        default: rc = fsdt_launch<3>(pp, g, m->ngp, m->batch, false, s); break;
    }
    if (rc) return rc;
    DN_LAUNCH_CHECK();
    return 0;
}
