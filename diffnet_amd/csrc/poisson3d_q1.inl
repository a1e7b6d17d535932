// 3-D Q1 fused Poisson kernel, fully sum-factorised marching form.  Included by
// poisson3d_q1_g{2,3,4}.hip with DN_NGP defined.
//
// grid = (chunks_x * tiles_y, strips_z, B), block = (TX, TY).  Thread (tx, ty) owns E consecutive elements of
// element row ey and marches over element planes.  Carried per element: the in-plane stage of the lower node plane
// (VU/VX/VY/VN/VF at the in-plane Gauss points) and the cotangents of that plane's stage values from the layer
// below.  Per layer: load the two node rows of the new plane, x- and y-stage them, the O(NGP^2) layer arithmetic
// (q1_layer_3d), then ONE in-plane transpose (y then x) per finished plane; contributions to nodes shared with the
// neighbouring threads (right, up, up-right) go through a double-buffered LDS slot, one LDS-only barrier per layer.
#include <cstdlib>

#include "poisson_common.h"

namespace dn {

enum : int { FL3_NU = 1, FL3_F = 2, FL3_FGP = 4, FL3_BC = 8, FL3_BC_U8C = 16, FL3_BC_ONE = 32, FL3_E1G = 64, FL3_BC_F32 = 128, FL3_LOAD = 256, FL3_BOX = 512 };   // FL3_BOX (two-element node-owner form): at least one condition is given as faces of the domain box (DN_MASK_BOX): no array, no load   // FL3_LOAD (with FL3_F, two-element node-owner form): `f` holds the ASSEMBLED load vector b_a = sum_e sum_g W_g N_a f_g (dn_poisson_args.f_is_load) -- one FMA per node instead of the element's forcing arithmetic   // FL3_BC_F32 (node-owner form, with FL3_BC_U8C): the constant-value masks are fp32 images (> 0.5), the reference's format   // FL3_BC_U8C: uint8 masks with constant values only; FL3_BC_ONE (node-owner form): exactly one of them

template <int NGP, int E>
struct PlaneState3D {
    float VU[E][NGP][NGP], VX[E][NGP], VY[E][NGP], VN[E][NGP][NGP], VF[E][NGP][NGP];
    float keep[E];
};

// pure 1-D weight of Gauss point g (1 at compile time for the unit-weight rule)
#define T_W(T, g, UW) ((UW) ? 1.f : (T).w[g])

#ifndef DN_Q1W_WAVES_T16
#define DN_Q1W_WAVES_T16 5        // 16-wide one-element form with the next plane's loads in flight: <= 96 VGPRs
#endif
#ifndef DN_Q1W_WAVES_E1
#define DN_Q1W_WAVES_E1 6         // one element per thread at 2 x 2 x 2 points: <= 80 VGPRs (6 waves per SIMD)
#endif
#ifndef DN_Q1W_WAVES
#define DN_Q1W_WAVES 4            // second-generation kernel: <= 128 VGPRs (4 waves per SIMD)
#endif
#ifndef DN_Q1_3D_WAVES
#define DN_Q1_3D_WAVES 2
#endif
#ifndef DN_PRIO_ROT3D
#define DN_PRIO_ROT3D 0
#endif

// =============================================================================================================
// Second-generation kernel (the round-1 kernel it replaced -- poisson3d_q1m_kernel, ~1/3 more VALU instructions per element -- is no longer
// in the tree; its numbers stay in profiles/r1_*): same mapping, hand-over and reductions.
//   * coefficient planes are staged WEIGHTED (w_j w_i nu, w_j w_i f) together with their in-plane sums A[j], B[i], once per
//     plane; the layer arithmetic (q1_layer_3d_w) works on raw differences and applies 1/h^2, alpha and the user scale
//     once per element; carried cotangents are folded into the layer's own fused multiply-adds;
//   * UW (exact 2-point rule, all weights 1): every weight multiplication vanishes at compile time;
//   * every global access is SGPR base + 32-bit byte offset (ld_at / st_at): no 64-bit VALU address arithmetic.
// =============================================================================================================
template <int NGP, int E>
struct PlaneW {
    float VU[E][NGP][NGP], VX[E][NGP], VY[E][NGP];
    float VN[E][NGP][NGP];                            // weighted nu at the in-plane Gauss points
    float VF[E][NGP][NGP];                            // weighted f
    float keep[E];
};

// in-plane stage of u for one element: nodal values r0[e], r0[e+1] (row ey) and r1[e], r1[e+1] (row ey + 1).
// V = float, or v2f: two elements at once in the halves of packed-fp32 registers (v_pk_fma_f32 ...)
template <int NGP, typename V = float>
__device__ __forceinline__ void stage_u3(const ElemTab& T, V a0, V a1, V b0, V b1, V (&VU)[NGP][NGP], V (&VX)[NGP], V (&VY)[NGP]) {
    const V dx0 = a1 - a0, dx1 = b1 - b0, ddx = dx1 - dx0;
#pragma unroll
    for (int j = 0; j < NGP; ++j) VX[j] = vfma(T.b[j][1], ddx, dx0);
#pragma unroll
    for (int i = 0; i < NGP; ++i) {
        const V t0 = vfma(T.b[i][1], dx0, a0);
        VY[i] = vfma(T.b[i][1], dx1, b0) - t0;
#pragma unroll
        for (int j = 0; j < NGP; ++j) VU[j][i] = vfma(T.b[j][1], VY[i], t0);
    }
}

// weighted in-plane stage of a coefficient field: V[j][i] = w_j w_i * c(gp j, i)
template <int NGP, bool UW, typename V = float>
__device__ __forceinline__ void stage_w3(const ElemTab& T, V a0, V a1, V b0, V b1, V (&W)[NGP][NGP]) {
    const V dx0 = a1 - a0, dx1 = b1 - b0;
#pragma unroll
    for (int i = 0; i < NGP; ++i) {
        V t0, t1;
        if constexpr (UW) {
            t0 = vfma(T.b[i][1], dx0, a0);
            t1 = vfma(T.b[i][1], dx1, b0);
        } else {
            t0 = vfma(T.wb[i], dx0, T.w[i] * a0);
            t1 = vfma(T.wb[i], dx1, T.w[i] * b0);
        }
        const V dy = t1 - t0;
#pragma unroll
        for (int j = 0; j < NGP; ++j) W[j][i] = UW ? vfma(T.b[j][1], dy, t0) : vfma(T.wb[j], dy, T.w[j] * t0);
    }
}

// T16: one element per thread in tiles exactly 16 threads wide.  A thread row is then a DPP row, so the hand-over to the right
// neighbour is a `row_shr:1` move (zero fill at the tile's left edge) instead of an LDS slot, the up-right hand-over folds into
// the up slot, and a thread's two nodes per row come from ONE (4-byte aligned) dwordx2 / ushort load.
struct __attribute__((packed, aligned(4))) F2U { float a, b; };
struct __attribute__((packed, aligned(1))) B2U { uint8_t a, b; };

// lane l <- lane l - 1 inside each row of 16 lanes, 0 for the first lane.  NOT a DPP move: on gfx950 every DPP / SDWA / v_readlane
// instruction costs a SIMD ~33 cycles (about 14 plain VALU instructions) as soon as two or more waves share it, while a
// ds_bpermute_b32 / ds_swizzle_b32 costs ~3 (tools/micro/valu_mem.hip, profiles/r2_valu_mem.txt).  `from` is the byte address of the
// source lane ((lane - 1) & 63) * 4, `nf` is 0 for the first lane of a row and 1 elsewhere.
#ifndef DN_T16_DPP
#define DN_T16_DPP 0
#endif
#ifndef DN_T16_PAIRED
#define DN_T16_PAIRED 1
#endif
__device__ __forceinline__ float lane_from_left(float v, int from, float nf) {
#if DN_T16_DPP
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
#else
    return nf * __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(from, __builtin_bit_cast(int, v)));
#endif
}

#if defined(DN_STAMP3D) && DN_NGP == 2
// Diagnostic build only (tools/stamp3d.py): per-wave cycle budget of the T16 loop, accumulated in scalar registers with s_memtime
// and written once at the end of the kernel by every 61st workgroup.  Phases per layer: A = wait for the prefetched plane + stage
// it, B = request the next plane + deferred store, C = layer arithmetic, D = hand-over write + barrier wait, E = hand-over read +
// finish the node value.
__device__ unsigned long long dn_stamp_buf[8192 * 8];
extern "C" int dn_debug_stamps(void* dst, size_t bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(dn_stamp_buf), bytes); }
#define DN_STAMP(var) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t__ = __builtin_amdgcn_s_memtime(); var += t__ - stamp_last; stamp_last = t__; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define DN_STAMP(var) do { } while (0)
#endif

template <int NGP, int E, bool VEC, int FL, bool UW, bool T16>
__global__ void __launch_bounds__(256, NGP == 4 ? 2 : (T16 ? (NGP == 2 ? ((FL & (FL3_BC | FL3_FGP)) ? 4 : DN_Q1W_WAVES_T16) : 3) : ((E == 1 && NGP == 2) ? ((FL & (FL3_BC | FL3_FGP)) ? 4 : DN_Q1W_WAVES_E1) : DN_Q1W_WAVES))) poisson3d_q1w_kernel(const PoissonParams p, const int chunks_x, const int tiles_y,
                                                                            const int strips_z) {
    static_assert(!T16 || E == 1, "T16 is the one-element-per-thread form");
    constexpr int NW = E;
    constexpr int G = NGP * NGP * NGP;
    constexpr bool HAS_NU = (FL & FL3_NU) != 0, HAS_F = (FL & FL3_F) != 0, FGP = (FL & FL3_FGP) != 0;
    constexpr bool BC_U8C = (FL & FL3_BC_U8C) != 0, BC_ANY = (FL & (FL3_BC | FL3_BC_U8C)) != 0;
    const int TX = blockDim.x, TY = blockDim.y;
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int tid = ty * TX + tx;
    // 1-D grid with an XCD-aware decode (cdna_hip_programming.md T1): workgroups are dealt round-robin to the 8 XCDs, each with its
    // own L2.  In-plane neighbours share node rows and z-neighbours one plane, so every XCD gets a contiguous range of the logical
    // order (chunk fastest, then tile, strip, sample): the halo of a tile is then read from HBM once instead of once per XCD.  The
    // remap is a bijection for any grid size.
    unsigned lid = blockIdx.x;
    {
        const unsigned nwg = gridDim.x, xcd = lid & 7u, idx = lid >> 3, base = nwg >> 3, rem = nwg & 7u;
        lid = xcd * base + min(xcd, rem) + idx;
    }
    const int chunk = (int)(lid % (unsigned)chunks_x);
    lid /= (unsigned)chunks_x;
    const int tile = (int)(lid % (unsigned)tiles_y);
    lid /= (unsigned)tiles_y;
    const int strip = selected_strip(p, (int)(lid % (unsigned)strips_z)), b = (int)(lid / (unsigned)strips_z);
    const int q = chunk * (TX - 1) + tx;
    const int ex0 = q * E, x0 = ex0;
    const int ey = tile * (TY - 1) + ty;
    const bool owner = !(chunk > 0 && tx == 0) && !(tile > 0 && ty == 0);
    const unsigned npl = (unsigned)(p.nx * p.ny);
    const int64_t nps = (int64_t)npl * p.nz;
    const unsigned epl = (unsigned)(p.nelx * p.nely);
    const unsigned eps = epl * (unsigned)p.nelz;
    const SampleBases sb = sample_bases(p, b, nps);
    const float* fgp = FGP ? p.fgp + (p.f_batched ? (int64_t)b * eps * G : 0) : nullptr;
    const int R = p.rows_per_strip;
    const int ez_own = strip * R;
    const int ez_begin = ez_own > 0 ? ez_own - 1 : 0;
    const int ez_end = min(ez_own + R, p.nelz);
    const bool row_ok = ey < p.nely;
    const bool noderow_ok = ey < p.ny;
    // Elements beyond the mesh (ragged right / upper edge of the last chunk / tile) are computed like any other on clamped,
    // finite node values and their results multiplied by 0: no per-element branches, no exec-masked regions in the loop.
    float okf[E];
#pragma unroll
    for (int e = 0; e < E; ++e) okf[e] = (row_ok && (ex0 + e < p.nelx)) ? 1.f : 0.f;

#if defined(DN_STAMP3D) && DN_NGP == 2
    unsigned long long stamp_A = 0, stamp_B = 0, stamp_C = 0, stamp_D = 0, stamp_E = 0, stamp_n = 0, stamp_last = 0;
    const unsigned long long stamp_t0 = __builtin_amdgcn_s_memtime();
#endif
    __shared__ float xch[2][T16 ? 1 : NW + 2][256];
    __shared__ double red[2 * (256 / 64)];      // block_sum2: two sums per wave
    __shared__ int last_flag;

    PlaneW<NGP, E> SA, SB;
    float cU[E][NGP][NGP], cX[E][NGP], cY[E][NGP];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        SA.keep[e] = SB.keep[e] = 1.f;
#pragma unroll
        for (int j = 0; j < NGP; ++j) {
            cX[e][j] = cY[e][j] = 0.f;
#pragma unroll
            for (int i = 0; i < NGP; ++i) {
                cU[e][j][i] = 0.f;
                SA.VN[e][j][i] = SB.VN[e][j][i] = T_W(p.T, j, UW) * T_W(p.T, i, UW);    // nu absent: the constant field 1
                SA.VF[e][j][i] = SB.VF[e][j][i] = 0.f;
            }
        }
    }

    // byte offsets of this thread's two node rows inside a plane; the plane offset is added per plane (one VALU add each)
    const int y0c = min(ey, p.ny - 1), y1c = min(ey + 1, p.ny - 1);
    const unsigned row0 = (unsigned)y0c * (unsigned)p.nx, row1 = (unsigned)y1c * (unsigned)p.nx;
    const unsigned x0c = (unsigned)min(x0, p.nx - 2);         // T16: in-bounds start of the thread's node pair
    const bool pair_shifted = x0 == p.nx - 1;                  // T16: owner of the last node column (no element of its own)

    // Raw node values of one plane (this thread's two rows), as loaded: `plane_issue` only issues the loads, `plane_consume`
    // applies the Dirichlet conditions and stages.  PF (T16 form): the loads of plane k + 2 are issued BEFORE the arithmetic of
    // layer k and consumed after it, so a wave does not sit on its memory latency once per layer (rocprofv3: 64 % of the wave
    // cycles were s_waitcnt / barrier waits without it); 14 more live VGPRs at one element per thread.
    // uint8 masks with constant values: both mask slots are always loaded (see plane_issue)
    const bool has_mask[2] = {sb.mask[0] != nullptr, sb.mask[1] != nullptr};
    const uint8_t* mask8[2];
    mask8[0] = reinterpret_cast<const uint8_t*>(has_mask[0] ? sb.mask[0] : sb.mask[1]);
    mask8[1] = reinterpret_cast<const uint8_t*>(has_mask[1] ? sb.mask[1] : sb.mask[0]);
    struct RawPlane {
        float ru[2][NW + 1], rn[2][NW + 1], rf[2][NW + 1];
        BcRaw<NW> braw[2];
        uint8_t m8[2][2][NW + 1];
    };
    auto plane_issue = [&](int zreq, RawPlane& W) {
#ifdef DN_ABL_ZFIX3D                       // timing experiment only: every plane re-reads the strip's first plane (cache hits)
        const unsigned zoff = (unsigned)ez_begin * npl + (unsigned)(zreq & 1) * 4u * (unsigned)p.nx;
#else
        const unsigned zoff = (unsigned)min(zreq, p.nz - 1) * npl;
#endif
        const unsigned rowoff[2] = {zoff + row0, zoff + row1};
        float (&ru)[2][NW + 1] = W.ru;
        float (&rn)[2][NW + 1] = W.rn;
        float (&rf)[2][NW + 1] = W.rf;
        BcRaw<NW> (&braw)[2] = W.braw;
        uint8_t (&m8)[2][2][NW + 1] = W.m8;
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            if constexpr (T16) {
                // both nodes of the row in one load; threads right of the mesh read the last valid pair (their element is masked)
                const unsigned o2 = rowoff[jb] + x0c;
#ifdef DN_ABL_NX4                          // timing experiment only: DN_ABL_NX4 aligned dwordx4 loads per layer and thread, nothing else
                if (jb == 0) {
                    float4 a = ld_at<float4>(sb.u, o2 & ~3u);
                    if (DN_ABL_NX4 > 1) { const float4 t = ld_at<float4>(sb.nu, o2 & ~3u); a.y += t.x; a.z += t.y; }
                    if (DN_ABL_NX4 > 2) { const float4 t = ld_at<float4>(sb.f, o2 & ~3u); a.w += t.z; a.x += t.w; }
                    ru[0][0] = a.x; ru[0][1] = a.y; ru[1][0] = a.z; ru[1][1] = a.w;
                    rn[0][0] = a.y; rn[0][1] = a.z; rn[1][0] = a.w; rn[1][1] = a.x;
                    rf[0][0] = a.z; rf[0][1] = a.w; rf[1][0] = a.x; rf[1][1] = a.y;
                    if constexpr (BC_U8C) {
                        const uint8_t mz = (uint8_t)(a.x == 123.456f);
                        m8[0][0][0] = m8[0][0][1] = m8[0][1][0] = m8[0][1][1] = m8[1][0][0] = m8[1][0][1] = m8[1][1][0] = m8[1][1][1] = mz;
                    }
                }
                continue;
#endif
#ifdef DN_ABL_ROW0DWORD                    // timing experiment only: one aligned dword / byte per field for row 0
                if (jb == 0) {
                    const float a0 = ld_at<float>(sb.u, o2);
                    ru[jb][0] = a0; ru[jb][1] = a0 + 0.5f;
                    if constexpr (HAS_NU) { const float t = ld_at<float>(sb.nu, o2); rn[jb][0] = t; rn[jb][1] = t; }
                    if constexpr (HAS_F) { const float t = ld_at<float>(sb.f, o2); rf[jb][0] = t; rf[jb][1] = t; }
                    if constexpr (BC_U8C) {
#pragma unroll
                        for (int k = 0; k < 2; ++k) { const uint8_t t = ld_at<uint8_t>(mask8[k], o2); m8[jb][k][0] = t; m8[jb][k][1] = t; }
                    }
                    continue;
                }
#endif
#if defined(DN_ABL_LOADS3D) || defined(DN_ABL_ROW0ONLY) || defined(DN_ABL_ROW0DWORD)   // timing experiment only: no VMEM loads (all rows / row 1 only); values made up from the offset
#if !defined(DN_ABL_LOADS3D)
                if (jb == 1)
#endif
                {
                    const float t = __uint_as_float((o2 & 0xffffu) | 0x3f800000u);
                    ru[jb][0] = t; ru[jb][1] = t + 0.5f;
                    rn[jb][0] = t; rn[jb][1] = t; rf[jb][0] = t; rf[jb][1] = t;
                    if constexpr (BC_U8C) { m8[jb][0][0] = m8[jb][0][1] = m8[jb][1][0] = m8[jb][1][1] = (uint8_t)(o2 == 0xffffffffu); }
                    continue;
                }
#endif
#if DN_T16_PAIRED          // one 4-byte-aligned dwordx2 / ushort per node pair; two aligned dword loads instead (DN_T16_PAIRED=0) measured 165 -> 249 us at 256^3: the second load hits lines still in flight (profiles/r2_3d_bottleneck.md)
                const F2U a = ld_at<F2U>(sb.u, o2);
                ru[jb][0] = a.a; ru[jb][1] = a.b;
                if constexpr (HAS_NU) { const F2U t = ld_at<F2U>(sb.nu, o2); rn[jb][0] = t.a; rn[jb][1] = t.b; }
                if constexpr (HAS_F) { const F2U t = ld_at<F2U>(sb.f, o2); rf[jb][0] = t.a; rf[jb][1] = t.b; }
#else
                ru[jb][0] = ld_at<float>(sb.u, o2); ru[jb][1] = ld_at<float>(sb.u, o2 + 1u);
                if constexpr (HAS_NU) { rn[jb][0] = ld_at<float>(sb.nu, o2); rn[jb][1] = ld_at<float>(sb.nu, o2 + 1u); }
                if constexpr (HAS_F) { rf[jb][0] = ld_at<float>(sb.f, o2); rf[jb][1] = ld_at<float>(sb.f, o2 + 1u); }
#endif
                if constexpr (BC_U8C) {
                    // unconditional loads (an absent mask reads the other one and is ignored): a load inside a uniform branch makes
                    // the compiler wait vmcnt(0) at the end of the branch, which serialises every load of the plane behind it
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
#if DN_T16_PAIRED
                        const B2U t = ld_at<B2U>(mask8[k], o2);
                        m8[jb][k][0] = t.a; m8[jb][k][1] = t.b;
#else
                        m8[jb][k][0] = ld_at<uint8_t>(mask8[k], o2); m8[jb][k][1] = ld_at<uint8_t>(mask8[k], o2 + 1u);
#endif
                    }
                } else if constexpr (BC_ANY) {
                    bc_issue<NW, false>(p, sb, rowoff[jb], (int)x0c, braw[jb]);
                }
            } else {
                load_seg<NW, VEC>(sb.u, rowoff[jb], x0, p.nx, ru[jb]);
                if constexpr (HAS_NU) load_seg<NW, VEC>(sb.nu, rowoff[jb], x0, p.nx, rn[jb]);
                if constexpr (HAS_F) load_seg<NW, VEC>(sb.f, rowoff[jb], x0, p.nx, rf[jb]);
                if constexpr (BC_U8C) {
#pragma unroll
                    for (int k = 0; k < 2; ++k) load_seg<NW, VEC>(mask8[k], rowoff[jb], x0, p.nx, m8[jb][k]);
                } else if constexpr (BC_ANY) {
                    bc_issue<NW, VEC>(p, sb, rowoff[jb], x0, braw[jb]);
                }
            }
        }
    };
    auto plane_consume = [&](RawPlane& W, PlaneW<NGP, E>& S) {
        float (&ru)[2][NW + 1] = W.ru;
        float (&rn)[2][NW + 1] = W.rn;
        float (&rf)[2][NW + 1] = W.rf;
        BcRaw<NW> (&braw)[2] = W.braw;
        uint8_t (&m8)[2][2][NW + 1] = W.m8;
        float kall[NW + 1];                   // 0 on Dirichlet nodes of row ey (all NW + 1 loaded nodes)
        if constexpr (BC_U8C) {
#pragma unroll
            for (int n = 0; n <= NW; ++n) kall[n] = 1.f;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float val = p.bc[k].value;
#pragma unroll
                for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                    for (int n = 0; n <= NW; ++n) {
                        const bool set = has_mask[k] && m8[jb][k][n] != 0;
                        ru[jb][n] = set ? val : ru[jb][n];
                        if (jb == 0) kall[n] = set ? 0.f : kall[n];
                    }
            }
        } else if constexpr (BC_ANY) {
            float k1[NW + 1];
            bc_apply_all<NW>(p, sb, braw[0], ru[0], kall);
            bc_apply_all<NW>(p, sb, braw[1], ru[1], k1);
        } else {
#pragma unroll
            for (int n = 0; n <= NW; ++n) kall[n] = 1.f;
        }
        // T16: the thread of the last node column loads the pair (nx-2, nx-1); the node it owns is the second one
#pragma unroll
        for (int n = 0; n < NW; ++n) S.keep[n] = (T16 && pair_shifted) ? kall[n + 1] : kall[n];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            stage_u3<NGP>(p.T, ru[0][e], ru[0][e + 1], ru[1][e], ru[1][e + 1], S.VU[e], S.VX[e], S.VY[e]);
            if constexpr (HAS_NU) stage_w3<NGP, UW>(p.T, rn[0][e], rn[0][e + 1], rn[1][e], rn[1][e + 1], S.VN[e]);
            if constexpr (HAS_F) stage_w3<NGP, UW>(p.T, rf[0][e], rf[0][e + 1], rf[1][e], rf[1][e + 1], S.VF[e]);
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    auto plane_stage = [&](int zreq, PlaneW<NGP, E>& S) {
        RawPlane W;
        plane_issue(zreq, W);
        plane_consume(W, S);
    };

    float e1_acc = 0.f, e2_acc = 0.f, sq_acc = 0.f;
    int par = 0;

    // adjoint of stage_u3: cotangents of one plane's stage values -> contributions to the element's 2 x 2 nodes
    auto plane_transpose = [&](int e, const float (&tU)[NGP][NGP], const float (&tX)[NGP], const float (&tY)[NGP], float (&o)[2][NW + 1]) {
        float s0 = 0.f, t0 = 0.f, s1 = 0.f, t1 = 0.f;
#pragma unroll
        for (int i = 0; i < NGP; ++i) {
            float s = 0.f, t = 0.f;
#pragma unroll
            for (int j = 0; j < NGP; ++j) { s += tU[j][i]; t = fmaf(p.T.b[j][1], tU[j][i], t); }
            const float c1 = t + tY[i], c0 = s - c1;
            s0 += c0; t0 = fmaf(p.T.b[i][1], c0, t0);
            s1 += c1; t1 = fmaf(p.T.b[i][1], c1, t1);
        }
        float sX = 0.f, d1 = 0.f;
#pragma unroll
        for (int j = 0; j < NGP; ++j) { sX += tX[j]; d1 = fmaf(p.T.b[j][1], tX[j], d1); }
        const float g01 = t0 + (sX - d1), g11 = t1 + d1;
        o[0][e + 1] = fmaf(okf[e], g01, o[0][e + 1]); o[0][e] = fmaf(okf[e], s0 - g01, o[0][e]);
        o[1][e + 1] = fmaf(okf[e], g11, o[1][e + 1]); o[1][e] = fmaf(okf[e], s1 - g11, o[1][e]);
    };

    const unsigned out_row = (unsigned)ey * (unsigned)p.nx;
    const int from_left = (int)(((unsigned)tid - 1u) & 63u) << 2;      // T16 hand-over: source lane of lane_from_left
    const float nfirst = tx > 0 ? 1.f : 0.f;
    float pend_v = 0.f;                       // T16: output of the last emitted plane, not yet stored
    unsigned pend_off = 0u;
    bool pend_st = false;
    auto flush_store = [&]() {
#ifdef DN_ABL_STORE3D                      // timing experiment only
        if (pend_st && pend_v == 123.456f) st_at<float>(sb.out, pend_off, pend_v);
#else
        if (pend_st) st_at<float>(sb.out, pend_off, pend_v);
#endif
        pend_st = false;
    };
    auto emit_plane = [&](const float (&o)[2][NW + 1], const float (&keep)[NW], int z, bool owned_plane) {
        if constexpr (T16) {
            const float left = lane_from_left(o[0][1], from_left, nfirst);          // right-hand contribution of the thread to the left
#ifdef DN_ABL_XCH3D                        // timing experiment only: no LDS hand-over, no barrier
            float t = o[0][0] + left + o[1][0] + lane_from_left(o[1][1], from_left, nfirst);
#else
            DN_STAMP(stamp_C);
            xch[par][0][tid] = o[1][0] + lane_from_left(o[1][1], from_left, nfirst);        // up slot with the up-right part of the left thread folded in
#ifndef DN_ABLATE_BAR3D                    // timing experiment only: results are wrong without the barrier
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
            DN_STAMP(stamp_D);
            // the value is stored by flush_store(), AFTER the next plane has been consumed and the one after it requested: a
            // store issued here would be younger than those loads' consumer wait, which the compiler turns into vmcnt(0)
            float t = o[0][0] + left;
            if (ty > 0) t += xch[par][0][tid - 16];
#endif
            t *= keep[0];
            const bool st = owned_plane && owner && noderow_ok;
            sq_acc = st ? fmaf(t, t, sq_acc) : sq_acc;
            pend_v = t * p.out_scale;
            pend_off = (unsigned)z * npl + out_row + (unsigned)x0;
            pend_st = st && sb.out != nullptr && x0 < p.nx;
            par ^= 1;
            return;
        }
        xch[par][0][tid] = o[0][NW];
#pragma unroll
        for (int n = 0; n < NW; ++n) xch[par][(T16 ? 0 : 1 + n)][tid] = o[1][n];
        xch[par][T16 ? 0 : NW + 1][tid] = o[1][NW];
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (owned_plane && owner && noderow_ok) {
            float v[NW];
#pragma unroll
            for (int n = 0; n < NW; ++n) {
                float t = o[0][n];
                if (ty > 0) t += xch[par][T16 ? 0 : 1 + n][tid - TX];
                if (n == 0) {
                    if (tx > 0) t += xch[par][0][tid - 1];
                    if (tx > 0 && ty > 0) t += xch[par][T16 ? 0 : NW + 1][tid - TX - 1];
                }
                t *= keep[n];
                sq_acc = fmaf(t, t, sq_acc);
                v[n] = t * p.out_scale;
            }
            if (sb.out) store_seg<NW, VEC>(sb.out, (unsigned)z * npl + out_row, x0, p.nx, v);
        }
        par ^= 1;
    };

    auto layer = [&](int ez, const PlaneW<NGP, E>& L, const PlaneW<NGP, E>& U) {
        const bool own_layer = ez >= ez_own;
        const float cnt = (own_layer && owner) ? 1.f : 0.f;
        float o[2][NW + 1], le1 = 0.f, le2 = 0.f;
#pragma unroll
        for (int n = 0; n <= NW; ++n) o[0][n] = o[1][n] = 0.f;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            float fg[G];
            if constexpr (FGP) {
                const unsigned eo = (unsigned)ez * epl + (unsigned)min(ey, p.nely - 1) * (unsigned)p.nelx + (unsigned)min(ex0 + e, p.nelx - 1);
#pragma unroll
                for (int gi = 0; gi < G; ++gi) fg[gi] = ld_at<float>(fgp, eo + (unsigned)gi * eps);
            }
            float tU[NGP][NGP], tX[NGP], tY[NGP], e1, e2;
            q1_layer_3d_w<NGP, FGP, HAS_F, UW>(p.T, L.VU[e], U.VU[e], L.VX[e], U.VX[e], L.VY[e], U.VY[e], L.VN[e], U.VN[e], L.VF[e], U.VF[e],
                                               fg, cU[e], cX[e], cY[e], tU, tX, tY, e1, e2);
            // pin the element's energy sums here: left alone, the compiler sinks their fused multiply-adds to the end of the loop
            // body and keeps every Gauss-point factor alive until then (232 VGPRs instead of ~110)
            asm volatile("" : "+v"(e1), "+v"(e2));
            le1 = fmaf(okf[e], e1, le1);
            le2 = fmaf(okf[e], e2, le2);
            plane_transpose(e, tU, tX, tY, o);
            __builtin_amdgcn_sched_barrier(0);      // keep the element streams apart: interleaving them doubles the live set
        }
        e1_acc = fmaf(cnt, le1, e1_acc);
        e2_acc = fmaf(cnt, le2, e2_acc);
        emit_plane(o, L.keep, ez, own_layer);
    };

    plane_stage(ez_begin, SA);
    if constexpr (T16) {
        // software pipeline: plane k + 2 is in flight while layer k is computed (planes beyond the mesh re-read the last one)
        RawPlane W;
        plane_issue(ez_begin + 1, W);
        int ez = ez_begin;
#if defined(DN_STAMP3D) && DN_NGP == 2
        stamp_last = __builtin_amdgcn_s_memtime();
#endif
#pragma nounroll
        for (; ez + 1 < ez_end; ez += 2) {
            plane_consume(W, SB);
            DN_STAMP(stamp_A);
            plane_issue(ez + 2, W);
            flush_store();
            DN_STAMP(stamp_B);
            layer(ez, SA, SB);
            DN_STAMP(stamp_E);
            plane_consume(W, SA);
            DN_STAMP(stamp_A);
            plane_issue(ez + 3, W);
            flush_store();
            DN_STAMP(stamp_B);
            layer(ez + 1, SB, SA);
            DN_STAMP(stamp_E);
#if defined(DN_STAMP3D) && DN_NGP == 2
            stamp_n += 2;
#endif
        }
        if (ez < ez_end) {
            plane_consume(W, SB);
            flush_store();
            layer(ez, SA, SB);
            SA = SB;
        }
        flush_store();
    } else {
        // two layers per trip with the roles of the two plane states swapped: no state copy at the end of a layer
        int ez = ez_begin;
#pragma nounroll
        for (; ez + 1 < ez_end; ez += 2) {
            plane_stage(ez + 1, SB);
            layer(ez, SA, SB);
            plane_stage(ez + 2, SA);
            layer(ez + 1, SB, SA);
        }
        if (ez < ez_end) {
            plane_stage(ez + 1, SB);
            layer(ez, SA, SB);
            SA = SB;
        }
    }
    if (ez_end == p.nelz) {       // the last strip owns the top boundary plane: only the layer below contributes
        float o[2][NW + 1];
#pragma unroll
        for (int n = 0; n <= NW; ++n) o[0][n] = o[1][n] = 0.f;
#pragma unroll
        for (int e = 0; e < E; ++e) plane_transpose(e, cU[e], cX[e], cY[e], o);
        emit_plane(o, SA.keep, p.nz - 1, true);
        if constexpr (T16) flush_store();
    }

#if defined(DN_STAMP3D) && DN_NGP == 2
    if constexpr (T16) {
        if (blockIdx.x % 61u == 0u && (tid & 63) == 0) {
            const unsigned slot = (blockIdx.x / 61u) * 4u + (unsigned)(tid >> 6);
            if (slot < 8192u) {
                unsigned long long* d = dn_stamp_buf + slot * 8u;
                d[0] = stamp_A; d[1] = stamp_B; d[2] = stamp_C; d[3] = stamp_D; d[4] = stamp_E; d[5] = stamp_n;
                d[6] = stamp_t0; d[7] = __builtin_amdgcn_s_memtime();
            }
        }
    }
#endif
    if (p.want_sums) finish_sums(p, e1_acc, e2_acc, sq_acc, tid, TX * TY, red, &last_flag, (double)p.T.esc);
}

// =============================================================================================================
// Third form (default for nodal / absent forcing and no or uint8 constant-value conditions): EVERY NODE IS REQUESTED ONCE per
// workgroup and plane.  The T16 form above asks for every node four times (two rows x two columns of threads, misaligned
// dwordx2 / ushort pairs): 182 TA-cycles per wave and layer, which is what bounds it (profiles/r2_3d_bottleneck.md).  Here a thread
// loads only the node it owns -- aligned dword / byte loads, 4 rows x 16 lanes per wave-instruction --, applies the Dirichlet
// conditions to it once, and publishes {u', nu, f, keep} as one 16-byte record in an LDS tile of 17 x 17 nodes; the 33 halo nodes
// of the tile (column 16, row 16) are loaded 9 per wave by lanes 0..8 of the same unconditional instructions.  After the barrier
// the hand-over needs anyway, a thread reads the four records of its element with ds_read_b128.  One barrier per layer:
//     request plane k + 2  ->  gather + stage plane k + 1 from LDS  ->  layer k  ->  publish plane k + 2, hand-over  ->  barrier  ->  finish plane k
// =============================================================================================================
#ifndef DN_TAB_VGPR
#define DN_TAB_VGPR 0             // 1: 1-D tables in vector registers (103 VGPRs, 4 waves): measured equal (profiles/r2_prio3d.txt), off
#endif
#ifndef DN_Q1N_WAVES
#define DN_Q1N_WAVES 5            // 2 x 2 x 2 points: <= 96 VGPRs.  The 3- and 4-point rules get 3 / 2 waves per SIMD (<= 168 / 256 VGPRs): at 5 they spilled 100-500 bytes of scratch per thread
#endif
#ifndef DN_PRIO3D
#define DN_PRIO3D 0               // 1 / 2: static / rotating wave priorities per workgroup: measured equal (profiles/r2_prio3d.txt), off
#endif
template <int NGP, int FL, bool UW>
__global__ void __launch_bounds__(256, NGP == 2 ? DN_Q1N_WAVES : (NGP == 3 ? 3 : 2)) poisson3d_q1n_kernel(const PoissonParams p, const int chunks_x, const int tiles_y, const int strips_z) {
    constexpr bool HAS_NU = (FL & FL3_NU) != 0, HAS_F = (FL & FL3_F) != 0, BC_U8C = (FL & FL3_BC_U8C) != 0;
    constexpr int NMASK = !BC_U8C ? 0 : ((FL & FL3_BC_ONE) ? 1 : 2);
    constexpr bool MASK_F32 = (FL & FL3_BC_F32) != 0;
    // FL3_E1G: the stiffness part of the energy is not summed Gauss point by Gauss point but taken from the finished nodal values:
    // sum_a u_a out_a = alpha * sum W nu |grad u|^2 - beta * sum W f u  (u = sum_a u_a N_a), one FMA per node instead of 15 per element
    constexpr bool E1G = (FL & FL3_E1G) != 0;     // uint8 mask arrays read per node (compile-time: no load in a uniform branch)
    static_assert((FL & (FL3_FGP | FL3_BC)) == 0, "node-owner form: nodal forcing, uint8 constant-value conditions");
    constexpr int E = 1, NW = 1;
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int tid = ty * 16 + tx;
    unsigned lid = blockIdx.x;                    // XCD-aware decode, see poisson3d_q1w_kernel
    {
        const unsigned nwg = gridDim.x, xcd = lid & 7u, idx = lid >> 3, base = nwg >> 3, rem = nwg & 7u;
        lid = xcd * base + min(xcd, rem) + idx;
    }
    const int chunk = (int)(lid % (unsigned)chunks_x);
    lid /= (unsigned)chunks_x;
    const int tile = (int)(lid % (unsigned)tiles_y);
    lid /= (unsigned)tiles_y;
    const int strip = selected_strip(p, (int)(lid % (unsigned)strips_z)), b = (int)(lid / (unsigned)strips_z);
    const int nx0 = chunk * 15, ny0 = tile * 15;               // first node of the tile
    const int x0 = nx0 + tx, ey = ny0 + ty;                    // the thread's node == lower-left node of its element
    const bool owner = !(chunk > 0 && tx == 0) && !(tile > 0 && ty == 0);
    const unsigned npl = (unsigned)(p.nx * p.ny);
    const int64_t nps = (int64_t)npl * p.nz;
    const SampleBases sb = sample_bases(p, b, nps);
    const int R = p.rows_per_strip;
    const int ez_own = strip * R;
    const int ez_begin = ez_own > 0 ? ez_own - 1 : 0;
    const int ez_end = min(ez_own + R, p.nelz);
    const bool noderow_ok = ey < p.ny;
    const float okf = (ey < p.nely && x0 < p.nelx) ? 1.f : 0.f;         // elements beyond the mesh: computed on clamped data, scaled by 0

    // DN_TAB_VGPR: the 1-D tables the loop multiplies with in VECTOR registers.  In isolation a VALU instruction with an SGPR operand
    // does not pair with another wave's instruction on gfx950 (3.8 SIMD-cycles per instruction at 90 % SGPR-operand share against
    // 2.3 with none, tools/micro/valu_sgpr.hip, profiles/r2_valu_sgpr.txt) and half of this loop's instructions read a table
    // entry -- but this kernel's waves do not pair anyway (its layer period is the sum of its phases), so the switch measured equal.
    ElemTab TV = p.T;
#if DN_TAB_VGPR
#define DN_V(x) asm volatile("" : "+v"(x))
#pragma unroll
    for (int j = 0; j < NGP; ++j) {
        DN_V(TV.b[j][1]);
        if constexpr (!UW) { DN_V(TV.w[j]); DN_V(TV.wb[j]); }
    }
    DN_V(TV.m[1]); DN_V(TV.m01);
    if constexpr (HAS_F) { DN_V(TV.m[2]); DN_V(TV.m12); DN_V(TV.nbw); }
    DN_V(TV.kap[0]); DN_V(TV.kap[1]); DN_V(TV.kap[2]);
#undef DN_V
#endif
#if defined(DN_STAMP3D) && DN_NGP == 2
    unsigned long long stamp_A = 0, stamp_B = 0, stamp_C = 0, stamp_D = 0, stamp_E = 0, stamp_n = 0, stamp_last = 0;
    const unsigned long long stamp_t0 = __builtin_amdgcn_s_memtime(), stamp_rt0 = __builtin_amdgcn_s_memrealtime();   // shader clock / constant 100 MHz clock
#endif
    __shared__ float4 rec[2][17 * 17];            // [plane parity][node row * 17 + node column] = {u after Dirichlet, nu, f, keep}
    __shared__ float xch[2][256];
    __shared__ double red[2 * (256 / 64)];      // block_sum2: two sums per wave
    __shared__ int last_flag;

    // in-plane offsets (nodes, clamped into the mesh) of the node this thread owns and of the halo node it fetches
    const unsigned own_off = (unsigned)min(ey, p.ny - 1) * (unsigned)p.nx + (unsigned)min(x0, p.nx - 1);
    const int lane = tid & 63, wave = tid >> 6;
    // 33 halo nodes, 9 per wave; their u, nu and f come from ONE load instruction: lanes 0..8 fetch u, 9..17 nu, 18..26 f of the wave's nine nodes
    // through per-lane 64-bit addresses (round 3: a vector-memory instruction costs a wave ~64 cycles of issue whatever its active lanes); the
    // other lanes repeat lane 26.  Each lane writes its one component of the halo record.
    const int hgrp = min(lane / 9, 2), hsub = min(lane - 9 * hgrp, 8);
    const int hidx = min(wave * 9 + hsub, 32);
    const int hrow = hidx < 16 ? hidx : 16, hcol = hidx < 16 ? 16 : hidx - 16;
    const unsigned halo_off = (unsigned)min(ny0 + hrow, p.ny - 1) * (unsigned)p.nx + (unsigned)min(nx0 + hcol, p.nx - 1);
    const bool halo_lane = lane < 27 && wave * 9 + hsub < 33 && (hgrp == 0 || (hgrp == 1 ? HAS_NU : HAS_F));
    const int own_rec = ty * 17 + tx, halo_rec = hrow * 17 + hcol;
    const float* const halo_src = (hgrp == 1 && HAS_NU) ? sb.nu : ((hgrp == 2 && HAS_F) ? sb.f : sb.u);       // per lane

    const bool has_mask[2] = {sb.mask[0] != nullptr, sb.mask[1] != nullptr};
    const uint8_t* mask8[2];
    mask8[0] = reinterpret_cast<const uint8_t*>(has_mask[0] ? sb.mask[0] : sb.mask[1]);      // an absent condition re-reads the other one and is
    mask8[1] = reinterpret_cast<const uint8_t*>(has_mask[1] ? sb.mask[1] : sb.mask[0]);      // ignored: no load inside a wave-uniform branch
    const float* mask32[2] = {reinterpret_cast<const float*>(mask8[0]), reinterpret_cast<const float*>(mask8[1])};

    // with one condition, slot 0 of mask8 / bcval is the one that is present
    const float bcval[2] = {NMASK == 1 ? (has_mask[0] ? p.bc[0].value : p.bc[1].value) : p.bc[0].value, p.bc[1].value};
    struct RawNodes { float u, n, f, h; uint8_t m[2][2]; float mf[2][2]; };       // own node: u, n, f; halo node: h (u, nu or f by lane group); masks [0] own, [1] halo
    auto plane_request = [&](int zreq, RawNodes& W) {
        const unsigned zoff = (unsigned)min(zreq, p.nz - 1) * npl;
        const unsigned o[2] = {zoff + own_off, zoff + halo_off};
        W.u = ld_at<float>(sb.u, o[0]);
        if constexpr (HAS_NU) W.n = ld_at<float>(sb.nu, o[0]);
        if constexpr (HAS_F) W.f = ld_at<float>(sb.f, o[0]);
        W.h = halo_src[o[1]];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if constexpr (BC_U8C) {
#pragma unroll
                for (int k = 0; k < NMASK; ++k) {
                    if constexpr (MASK_F32) W.mf[h][k] = ld_at<float>(mask32[k], o[h]);
                    else W.m[h][k] = ld_at<uint8_t>(mask8[k], o[h]);
                }
            }
        }
    };
    auto plane_publish = [&](const RawNodes& W, int zpl) {
        float uu = W.u, keep = 1.f, hv = W.h;
        if constexpr (BC_U8C) {
#pragma unroll
            for (int k = 0; k < NMASK; ++k) {
                const bool set = (NMASK == 1 || has_mask[k]) && (MASK_F32 ? W.mf[0][k] > 0.5f : W.m[0][k] != 0);
                uu = set ? bcval[k] : uu;
                keep = set ? 0.f : keep;
                const bool seth = hgrp == 0 && (NMASK == 1 || has_mask[k]) && (MASK_F32 ? W.mf[1][k] > 0.5f : W.m[1][k] != 0);
                hv = seth ? bcval[k] : hv;
            }
        }
        rec[zpl & 1][own_rec] = make_float4(uu, HAS_NU ? W.n : 1.f, HAS_F ? W.f : 0.f, keep);
        // (the .w of a halo record is never read; absent nu / f keep the 1 / 0 the prologue put there)
        if (halo_lane) reinterpret_cast<float*>(&rec[zpl & 1][halo_rec])[hgrp] = hv;
    };
    float keep_lo = 1.f, keep_up = 1.f;           // keep of the own node in the lower / upper plane of the current layer
    float u_lo = 0.f, u_up = 0.f, ut_acc = 0.f;   // E1G: the own node's value after the Dirichlet conditions, sum of u * out over the owned nodes
    auto plane_gather = [&](int zpl, PlaneW<NGP, 1>& S, float& keep, float& uown) {
        const float4* t = &rec[zpl & 1][own_rec];
        const float4 a0 = t[0], a1 = t[1], b0 = t[17], b1 = t[18];
        keep = a0.w;
        uown = a0.x;
        stage_u3<NGP>(TV, a0.x, a1.x, b0.x, b1.x, S.VU[0], S.VX[0], S.VY[0]);
        if constexpr (HAS_NU) stage_w3<NGP, UW>(TV, a0.y, a1.y, b0.y, b1.y, S.VN[0]);
        if constexpr (HAS_F) stage_w3<NGP, UW>(TV, a0.z, a1.z, b0.z, b1.z, S.VF[0]);
        __builtin_amdgcn_sched_barrier(0);
    };

    PlaneW<NGP, 1> SA, SB;
    float cU[NGP][NGP], cX[NGP], cY[NGP];
#pragma unroll
    for (int j = 0; j < NGP; ++j) {
        cX[j] = cY[j] = 0.f;
#pragma unroll
        for (int i = 0; i < NGP; ++i) {
            cU[j][i] = 0.f;
            SA.VN[0][j][i] = SB.VN[0][j][i] = T_W(TV, j, UW) * T_W(TV, i, UW);    // nu absent: the constant field 1
            SA.VF[0][j][i] = SB.VF[0][j][i] = 0.f;
        }
    }

    float e1_acc = 0.f, e2_acc = 0.f, sq_acc = 0.f;
    int par = 0;

    // adjoint of stage_u3: cotangents of one plane's stage values -> contributions to the element's 2 x 2 nodes
    auto plane_transpose = [&](const float (&tU)[NGP][NGP], const float (&tX)[NGP], const float (&tY)[NGP], float (&o)[2][2]) {
        float s0 = 0.f, t0 = 0.f, s1 = 0.f, t1 = 0.f;
#pragma unroll
        for (int i = 0; i < NGP; ++i) {
            float sv = 0.f, t = 0.f;
#pragma unroll
            for (int j = 0; j < NGP; ++j) { sv += tU[j][i]; t = fmaf(TV.b[j][1], tU[j][i], t); }
            const float c1 = t + tY[i], c0 = sv - c1;
            s0 += c0; t0 = fmaf(TV.b[i][1], c0, t0);
            s1 += c1; t1 = fmaf(TV.b[i][1], c1, t1);
        }
        float sX = 0.f, d1 = 0.f;
#pragma unroll
        for (int j = 0; j < NGP; ++j) { sX += tX[j]; d1 = fmaf(TV.b[j][1], tX[j], d1); }
        const float g01 = t0 + (sX - d1), g11 = t1 + d1;
        o[0][1] = okf * g01; o[0][0] = okf * (s0 - g01);
        o[1][1] = okf * g11; o[1][0] = okf * (s1 - g11);
    };

    const unsigned out_row = (unsigned)ey * (unsigned)p.nx;
    const int from_left = (int)(((unsigned)tid - 1u) & 63u) << 2;
    const float nfirst = tx > 0 ? 1.f : 0.f;
    float pend_v = 0.f;
    unsigned pend_off = 0u;
    bool pend_st = false;
    auto flush_store = [&]() {
        if (pend_st) st_at<float>(sb.out, pend_off, pend_v);
        pend_st = false;
    };
    // hand the finished contributions over, (optionally) publish the plane requested at the top of the layer, ONE barrier, finish the node
    auto emit_plane = [&](const float (&o)[2][2], float keep, float uown, int z, bool owned_plane, const RawNodes* W, int zpub) {
        DN_STAMP(stamp_C);
        const float left = lane_from_left(o[0][1], from_left, nfirst);
        xch[par][tid] = o[1][0] + lane_from_left(o[1][1], from_left, nfirst);
        if (W != nullptr) plane_publish(*W, zpub);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        DN_STAMP(stamp_D);
        float t = o[0][0] + left;
        if (ty > 0) t += xch[par][tid - 16];
        const bool st = owned_plane && owner && noderow_ok;
        if constexpr (E1G) ut_acc = st ? fmaf(t, uown, ut_acc) : ut_acc;      // before the Dirichlet rows are zeroed
        t *= keep;
        sq_acc = st ? fmaf(t, t, sq_acc) : sq_acc;
        pend_v = t * p.out_scale;
        pend_off = (unsigned)z * npl + out_row + (unsigned)x0;
        pend_st = st && sb.out != nullptr && x0 < p.nx;
        par ^= 1;
    };
    auto layer = [&](int ez, const PlaneW<NGP, 1>& L, const PlaneW<NGP, 1>& U, float keep, float uown, const RawNodes* W) {
        const bool own_layer = ez >= ez_own;
        const float cnt = (own_layer && owner) ? 1.f : 0.f;
        float o[2][2], fg[1] = {0.f};
        float tU[NGP][NGP], tX[NGP], tY[NGP], e1, e2;
        q1_layer_3d_w<NGP, false, HAS_F, UW>(TV, L.VU[0], U.VU[0], L.VX[0], U.VX[0], L.VY[0], U.VY[0], L.VN[0], U.VN[0], L.VF[0], U.VF[0],
                                             fg, cU, cX, cY, tU, tX, tY, e1, e2);
        if constexpr (E1G) {
            asm volatile("" : "+v"(e2));
        } else {
            asm volatile("" : "+v"(e1), "+v"(e2));
            e1_acc = fmaf(cnt * okf, e1, e1_acc);
        }
        e2_acc = fmaf(cnt * okf, e2, e2_acc);
        plane_transpose(tU, tX, tY, o);
        __builtin_amdgcn_sched_barrier(0);
        emit_plane(o, keep, uown, ez, own_layer, W, ez + 2);
    };

    // prologue: planes ez_begin and ez_begin + 1 into LDS (both requested before the first is consumed: one memory latency and one barrier
    // instead of two -- a workgroup of a 128^3 launch marches only ~10 layers), the lower one staged
    if (lane < 9 && wave * 9 + lane < 33) {          // the constant components of the halo records (both parities): nu = 1, f = 0, keep = 1
        rec[0][halo_rec] = make_float4(0.f, 1.f, 0.f, 1.f);
        rec[1][halo_rec] = make_float4(0.f, 1.f, 0.f, 1.f);
    }
    __syncthreads();
    RawNodes W;
    {
        RawNodes W0;
        plane_request(ez_begin, W0);
        plane_request(ez_begin + 1, W);
        plane_publish(W0, ez_begin);
        plane_publish(W, ez_begin + 1);
    }
    __syncthreads();
    plane_gather(ez_begin, SA, keep_lo, u_lo);
    __syncthreads();              // every thread has read plane ez_begin before the first layer publishes plane ez_begin + 2 into its slot
    int ez = ez_begin;
    // Wave priorities against lock-step: workgroups that start together and do identical work fall into phase (all request, then all
    // gather, then all compute: the layer period becomes the SUM of the VALU, TA and LDS times instead of their maximum).  Different
    // priorities for the workgroups resident on a CU let one run ahead through its arithmetic while the others use the memory paths.
    const int prio_hash = (int)((blockIdx.x >> 8) & 3u);
    auto set_prio = [&](int step) {
#if DN_PRIO3D == 1
        (void)step;
        switch (prio_hash) {
            case 0: __builtin_amdgcn_s_setprio(0); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            default: __builtin_amdgcn_s_setprio(3); break;
        }
#elif DN_PRIO3D == 2
        switch ((prio_hash + (step >> 1)) & 3) {
            case 0: __builtin_amdgcn_s_setprio(0); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            default: __builtin_amdgcn_s_setprio(3); break;
        }
#else
        (void)step;
#endif
    };
    set_prio(0);
#if defined(DN_STAMP3D) && DN_NGP == 2
    stamp_last = __builtin_amdgcn_s_memtime();
#endif
#pragma nounroll
    for (; ez + 1 < ez_end; ez += 2) {
#if DN_PRIO3D == 2
        set_prio(ez - ez_begin);
#endif
        plane_request(ez + 2, W);                 // lands while this layer is computed; published before the layer's barrier
        flush_store();
        DN_STAMP(stamp_A);
        plane_gather(ez + 1, SB, keep_up, u_up);
        DN_STAMP(stamp_B);
        layer(ez, SA, SB, keep_lo, u_lo, &W);
        DN_STAMP(stamp_E);
        plane_request(ez + 3, W);
        flush_store();
        DN_STAMP(stamp_A);
        plane_gather(ez + 2, SA, keep_lo, u_lo);
        DN_STAMP(stamp_B);
        layer(ez + 1, SB, SA, keep_up, u_up, &W);
        DN_STAMP(stamp_E);
#if defined(DN_STAMP3D) && DN_NGP == 2
        stamp_n += 2;
#endif
    }
    bool odd = false;
    if (ez < ez_end) {
        flush_store();
        plane_gather(ez + 1, SB, keep_up, u_up);
        layer(ez, SA, SB, keep_lo, u_lo, nullptr);
        odd = true;
    }
    flush_store();
    if (ez_end == p.nelz) {       // the last strip owns the top boundary plane: only the layer below contributes
        float o[2][2];
        plane_transpose(cU, cX, cY, o);
        emit_plane(o, odd ? keep_up : keep_lo, odd ? u_up : u_lo, p.nz - 1, true, nullptr, 0);
        flush_store();
    }

#if defined(DN_STAMP3D) && DN_NGP == 2
    if (tid == 0) {                               // node-owner build: wave 0 of EVERY workgroup (per-CU timelines, tools/stamp3d.py)
        const unsigned slot = blockIdx.x;
        if (slot < 8192u) {
            unsigned long long* d = dn_stamp_buf + slot * 8u;
            // where the wave ran: HW_REG_HW_ID (4: wave / SIMD / CU / SH / SE ids) and HW_REG_XCC_ID (20), whole registers
            const unsigned long long hwid = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)), xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));
            d[0] = stamp_A; d[1] = stamp_B; d[2] = stamp_C; d[3] = stamp_D;
            d[4] = stamp_E | ((__builtin_amdgcn_s_memrealtime() - stamp_rt0) << 40);          // lifetime in 10-ns ticks in the upper bits
            d[5] = stamp_n | (hwid << 16) | ((xcc & 0xffull) << 48);
            d[6] = stamp_t0; d[7] = __builtin_amdgcn_s_memtime();
        }
    }
#endif
    if constexpr (E1G) e1_acc = (ut_acc / p.T.esc + p.T.beta * e2_acc) / p.T.alpha;       // per-thread share of sum W nu |grad u|^2 (the identity holds for the total)
    if (p.want_sums) finish_sums(p, e1_acc, e2_acc, sq_acc, tid, 256, red, &last_flag, (double)p.T.esc);
}

#if DN_NGP == 2
// =============================================================================================================
// Fourth form (round 3): the node-owner form with TWO ELEMENTS PER THREAD along x, for the exact 2-point rule.  The node-owner kernel above
// is bound by instructions, and a third of them are per THREAD, not per element: the requests (8 vector-memory instructions per layer), the
// LDS records (4 reads, 2 writes), the hand-over (2 lane exchanges, slot write / read), the barrier, the Dirichlet selects, the store.  Here a
// thread owns the nodes x = nx0 + 2 tx, + 1 of row ny0 + ty (one 8-byte load per field and plane, one 8-byte store) and the two elements to
// their right; the tile is 32 x 16 elements (31 x 15 of them not shared with a neighbouring tile).  Records live in two LDS arrays, even and
// odd node columns, so that the three records a thread reads per node row are conflict-free b128 reads (lane stride 16 bytes).  The halo -- node
// row 16 (33 nodes) and node column 32 (16 nodes) -- is loaded 13 nodes per wave.  Needs an even nx (8-byte aligned pairs).
// =============================================================================================================
#ifndef DN_Q1N2_WAVES
#define DN_Q1N2_WAVES 3
#endif
#ifndef DN_Q1N2_LOAD_WAVES
#define DN_Q1N2_LOAD_WAVES 3      // waves per SIMD asked of the load-vector instantiations (no staged forcing planes: 141-151 VGPRs with nu)
#endif
template <int FL>
__global__ void __launch_bounds__(256, (FL & FL3_LOAD) ? DN_Q1N2_LOAD_WAVES : DN_Q1N2_WAVES) poisson3d_q1n2_kernel(const PoissonParams p, const int chunks_x, const int tiles_y, const int strips_z) {
    constexpr int NGP = 2;
    constexpr bool UW = true;
    // LOADV: the forcing arrives as the assembled load vector (one value per node, used by the node's owner only: no staging, no element
    // arithmetic); F_ARR: the third array is read at all (it travels in the records' .z either way); HAS_F: nodal forcing, staged per element
    constexpr bool HAS_NU = (FL & FL3_NU) != 0, F_ARR = (FL & FL3_F) != 0, LOADV = F_ARR && (FL & FL3_LOAD) != 0, HAS_F = F_ARR && !LOADV;
    constexpr bool BC_U8C = (FL & FL3_BC_U8C) != 0, BOX = (FL & FL3_BOX) != 0;
    constexpr int NMASK = !BC_U8C ? 0 : ((FL & FL3_BC_ONE) ? 1 : 2);
    constexpr bool MASK_F32 = (FL & FL3_BC_F32) != 0;
    constexpr bool E1G = (FL & FL3_E1G) != 0;
    static_assert((FL & (FL3_FGP | FL3_BC)) == 0, "node-owner form: nodal forcing, constant-value conditions");
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int tid = ty * 16 + tx;
    unsigned lid = blockIdx.x;                    // XCD-aware decode, see poisson3d_q1w_kernel
    {
        const unsigned nwg = gridDim.x, xcd = lid & 7u, idx = lid >> 3, base = nwg >> 3, rem = nwg & 7u;
        lid = xcd * base + min(xcd, rem) + idx;
    }
    const int chunk = (int)(lid % (unsigned)chunks_x);
    lid /= (unsigned)chunks_x;
    const int tile = (int)(lid % (unsigned)tiles_y);
    lid /= (unsigned)tiles_y;
    const int strip = selected_strip(p, (int)(lid % (unsigned)strips_z)), b = (int)(lid / (unsigned)strips_z);
    const int nx0 = chunk * 30, ny0 = tile * 15;               // first node of the tile (chunks overlap by one thread column = two elements)
    const int x0 = nx0 + 2 * tx, ey = ny0 + ty;                // the thread's first node == lower-left node of its first element
    const bool owner = !(chunk > 0 && tx == 0) && !(tile > 0 && ty == 0);
    const unsigned npl = (unsigned)(p.nx * p.ny);
    const int64_t nps = (int64_t)npl * p.nz;
    const SampleBases sb = sample_bases(p, b, nps);
    const int R = p.rows_per_strip;
    const int ez_own = strip * R;
    const int ez_begin = ez_own > 0 ? ez_own - 1 : 0;
    const int ez_end = min(ez_own + R, p.nelz);
    const bool noderow_ok = ey < p.ny;
    const float okf[2] = {(ey < p.nely && x0 < p.nelx) ? 1.f : 0.f, (ey < p.nely && x0 + 1 < p.nelx) ? 1.f : 0.f};
    const ElemTab& TV = p.T;
#if defined(DN_STAMP3D)
    unsigned long long stamp_A = 0, stamp_B = 0, stamp_C = 0, stamp_D = 0, stamp_E = 0, stamp_n = 0, stamp_last = 0;
    const unsigned long long stamp_t0 = __builtin_amdgcn_s_memtime(), stamp_rt0 = __builtin_amdgcn_s_memrealtime();
#endif

    __shared__ float4 recE[2][17][17];            // [plane parity][node row][even node column / 2] = {u after Dirichlet, nu, f, keep}
    __shared__ float4 recO[2][17][16];
    __shared__ float2 xch[2][256];
    __shared__ double red[2 * (256 / 64)];
    __shared__ int last_flag;
    if (blockIdx.x == 0u) fold_prev_sums(p, tid, 256, red);       // dn_poisson_args.fold_prev: close the evaluation before this one

    // own pair (clamped into the mesh: nx is even, so a pair is inside or outside as a whole) and the halo node this thread fetches
    const unsigned own_off = (unsigned)min(ey, p.ny - 1) * (unsigned)p.nx + (unsigned)min(x0, p.nx - 2);
    const int lane = tid & 63, wave = tid >> 6;
    // 49 halo nodes, 13 per wave, and ONE load instruction for their u, nu and f: lanes 0..12 fetch u, 13..25 nu, 26..38 f of the wave's 13 nodes
    // through per-lane 64-bit addresses (a vector-memory instruction costs a wave ~64 cycles of issue whatever its active lanes: three
    // instructions for 13 lanes each were a third of the kernel's vector-memory issue time); the other lanes repeat lane 38
    const int hgrp = min(lane / 13, 2), hsub = min(lane - 13 * hgrp, 12);
    const int hidx = min(wave * 13 + hsub, 48);
    const int hrow = hidx < 33 ? 16 : hidx - 33, hcol = hidx < 33 ? hidx : 32;
    const unsigned halo_off = (unsigned)min(ny0 + hrow, p.ny - 1) * (unsigned)p.nx + (unsigned)min(nx0 + hcol, p.nx - 1);
    const bool halo_lane = lane < 39 && wave * 13 + hsub < 49 && (hgrp == 0 || (hgrp == 1 ? HAS_NU : HAS_F));       // (the load vector is needed at owned nodes only: no halo)
    float* const halo_rec0 = reinterpret_cast<float*>((hcol & 1) ? &recO[0][hrow][hcol >> 1] : &recE[0][hrow][hcol >> 1]) + hgrp;      // component .x / .y / .z
    const unsigned halo_par_stride = 4u * ((hcol & 1) ? 17u * 16u : 17u * 17u);                                                        // floats
    const float* const halo_src = (hgrp == 1 && HAS_NU) ? sb.nu : ((hgrp == 2 && HAS_F) ? sb.f : sb.u);                               // per lane

    const bool has_mask[2] = {sb.mask[0] != nullptr, sb.mask[1] != nullptr};
    const uint8_t* mask8[2];
    mask8[0] = reinterpret_cast<const uint8_t*>(has_mask[0] ? sb.mask[0] : sb.mask[1]);
    mask8[1] = reinterpret_cast<const uint8_t*>(has_mask[1] ? sb.mask[1] : sb.mask[0]);
    const float* mask32[2] = {reinterpret_cast<const float*>(mask8[0]), reinterpret_cast<const float*>(mask8[1])};
    // BOX: per condition the faces of the domain box it fixes (0: not a box condition).  In-plane part per node (this thread's pair, its halo
    // node), constant over the march; the two faces across the marched axis per plane (wave-uniform)
    const int bfaces[2] = {BOX && p.bc[0].kind == DN_MASK_BOX ? p.bc[0].box_faces : 0, BOX && p.bc[1].kind == DN_MASK_BOX ? p.bc[1].box_faces : 0};
    auto box_xy = [&](int k, int x, int y) {
        return ((bfaces[k] & DN_FACE_XLO) && x == 0) || ((bfaces[k] & DN_FACE_XHI) && x == p.nx - 1) || ((bfaces[k] & DN_FACE_YLO) && y == 0) ||
               ((bfaces[k] & DN_FACE_YHI) && y == p.ny - 1);
    };
    const bool bxy0[2] = {box_xy(0, x0, ey), box_xy(1, x0, ey)}, bxy1[2] = {box_xy(0, x0 + 1, ey), box_xy(1, x0 + 1, ey)};
    const bool bxyh[2] = {box_xy(0, nx0 + hcol, ny0 + hrow), box_xy(1, nx0 + hcol, ny0 + hrow)};

    struct RawNodes2 {
        float2 u, n, f;               // own pair
        float h;                      // halo node: u, nu or f by lane group
        uint16_t m[2];                // uint8 masks of the own pair (two bytes), per condition
        float2 mf[2];                 // fp32 masks of the own pair
        uint8_t hm[2];
        float hmf[2];
    };
    auto plane_request = [&](int zreq, RawNodes2& W) {
        const unsigned zoff = (unsigned)min(zreq, p.nz - 1) * npl;
        const unsigned oo = zoff + own_off, oh = zoff + halo_off;
        W.u = ld_at<float2>(sb.u, oo);
        if constexpr (HAS_NU) W.n = ld_at<float2>(sb.nu, oo);
        if constexpr (F_ARR) W.f = ld_at<float2>(sb.f, oo);
        W.h = halo_src[oh];
        if constexpr (BC_U8C) {
#pragma unroll
            for (int k = 0; k < NMASK; ++k) {
                if constexpr (MASK_F32) { W.mf[k] = ld_at<float2>(mask32[k], oo); W.hmf[k] = ld_at<float>(mask32[k], oh); }
                else { W.m[k] = ld_at<uint16_t>(mask8[k], oo); W.hm[k] = ld_at<uint8_t>(mask8[k], oh); }
            }
        }
    };
    // img[j]: the node is set in the j-th LOADED mask image (NMASK == 1: the one image is whichever condition has it); box[k]: by condition k's faces.
    // The conditions are applied in their order (the reference's torch.where lines, e.g. IBN_3D.py:119-122)
    auto record = [&](float uu, float nn, float ff, const bool (&img)[2], const bool (&box)[2]) {
        float keep = 1.f;
        if constexpr (BC_U8C || BOX) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                bool s = false;
                if constexpr (NMASK == 2) s = has_mask[k] && img[k];
                if constexpr (NMASK == 1) s = has_mask[k] && img[0];
                if constexpr (BOX) s = s || box[k];
                uu = s ? p.bc[k].value : uu;
                keep = s ? 0.f : keep;
            }
        }
        return make_float4(uu, HAS_NU ? nn : 1.f, F_ARR ? ff : 0.f, keep);
    };
    auto plane_publish = [&](const RawNodes2& W, int zpl) {
        bool s0[2] = {false, false}, s1[2] = {false, false}, sh[2] = {false, false};
        if constexpr (BC_U8C) {
#pragma unroll
            for (int k = 0; k < NMASK; ++k) {
                if constexpr (MASK_F32) { s0[k] = W.mf[k].x > 0.5f; s1[k] = W.mf[k].y > 0.5f; sh[k] = W.hmf[k] > 0.5f; }
                else { s0[k] = (W.m[k] & 0xffu) != 0; s1[k] = (W.m[k] >> 8) != 0; sh[k] = W.hm[k] != 0; }
            }
        }
        const int par = zpl & 1;
        bool b0[2] = {false, false}, b1[2] = {false, false}, bh[2] = {false, false};
        if constexpr (BOX) {
            const int zc = min(zpl, p.nz - 1);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const bool zf = ((bfaces[k] & DN_FACE_ZLO) && zc == 0) || ((bfaces[k] & DN_FACE_ZHI) && zc == p.nz - 1);
                b0[k] = bxy0[k] || zf; b1[k] = bxy1[k] || zf; bh[k] = bxyh[k] || zf;
            }
        }
        recE[par][ty][tx] = record(W.u.x, W.n.x, W.f.x, s0, b0);
        recO[par][ty][tx] = record(W.u.y, W.n.y, W.f.y, s1, b1);
        if (halo_lane) {
            float hv = W.h;
            if constexpr (BC_U8C || BOX) {
                if (hgrp == 0) hv = record(hv, 0.f, 0.f, sh, bh).x;
            }
            halo_rec0[par * halo_par_stride] = hv;       // (the .w of a halo record is never read; absent nu / f keep the 1 / 0 of the prologue)
        }
    };
    // From here on a value of type v2f holds the same quantity of the thread's two elements (.x: element at node column x0, .y: at x0 + 1) or
    // of its two nodes: the element arithmetic runs on packed fp32 instructions, one per pair.
    struct PlaneP { v2f VU[NGP][NGP], VX[NGP], VY[NGP], VN[NGP][NGP], VF[NGP][NGP]; };
    v2f ut_acc = 0.f;                             // E1G: sum of u * out over the owned nodes
    auto plane_gather = [&](int zpl, PlaneP& S) {
        const int par = zpl & 1;
        const float4 a0 = recE[par][ty][tx], a1 = recO[par][ty][tx], a2 = recE[par][ty][tx + 1];
        const float4 b0 = recE[par][ty + 1][tx], b1 = recO[par][ty + 1][tx], b2 = recE[par][ty + 1][tx + 1];
        stage_u3<NGP, v2f>(TV, v2f{a0.x, a1.x}, v2f{a1.x, a2.x}, v2f{b0.x, b1.x}, v2f{b1.x, b2.x}, S.VU, S.VX, S.VY);
        if constexpr (HAS_NU) stage_w3<NGP, UW, v2f>(TV, v2f{a0.y, a1.y}, v2f{a1.y, a2.y}, v2f{b0.y, b1.y}, v2f{b1.y, b2.y}, S.VN);
        if constexpr (HAS_F) stage_w3<NGP, UW, v2f>(TV, v2f{a0.z, a1.z}, v2f{a1.z, a2.z}, v2f{b0.z, b1.z}, v2f{b1.z, b2.z}, S.VF);
        __builtin_amdgcn_sched_barrier(0);
    };

    PlaneP SA, SB;
    v2f cU[NGP][NGP], cX[NGP], cY[NGP];
#pragma unroll
    for (int j = 0; j < NGP; ++j) {
        cX[j] = cY[j] = 0.f;
#pragma unroll
        for (int i = 0; i < NGP; ++i) {
            cU[j][i] = 0.f;
            SA.VN[j][i] = SB.VN[j][i] = 1.f;              // nu absent: the constant field 1 (unit weights)
            SA.VF[j][i] = SB.VF[j][i] = 0.f;
        }
    }

    v2f e1_acc2 = 0.f, e2_acc2 = 0.f, sq_acc2 = 0.f;
    const v2f okv = {okf[0], okf[1]};
    int par = 0;

    // adjoint of stage_u3: cotangents of a plane's stage values -> contributions to each element's 2 x 2 nodes
    auto plane_transpose = [&](const v2f (&tU)[NGP][NGP], const v2f (&tX)[NGP], const v2f (&tY)[NGP], v2f (&o)[2][2]) {
        v2f s0 = 0.f, t0 = 0.f, s1 = 0.f, t1 = 0.f;
#pragma unroll
        for (int i = 0; i < NGP; ++i) {
            v2f sv = 0.f, t = 0.f;
#pragma unroll
            for (int j = 0; j < NGP; ++j) { sv += tU[j][i]; t = vfma(TV.b[j][1], tU[j][i], t); }
            const v2f c1 = t + tY[i], c0 = sv - c1;
            s0 += c0; t0 = vfma(TV.b[i][1], c0, t0);
            s1 += c1; t1 = vfma(TV.b[i][1], c1, t1);
        }
        v2f sX = 0.f, d1 = 0.f;
#pragma unroll
        for (int j = 0; j < NGP; ++j) { sX += tX[j]; d1 = vfma(TV.b[j][1], tX[j], d1); }
        const v2f g01 = t0 + (sX - d1), g11 = t1 + d1;
        o[0][1] = okv * g01; o[0][0] = okv * (s0 - g01);
        o[1][1] = okv * g11; o[1][0] = okv * (s1 - g11);
    };

    const unsigned out_row = (unsigned)ey * (unsigned)p.nx;
    const int from_left = (int)(((unsigned)tid - 1u) & 63u) << 2;
    const float nfirst = tx > 0 ? 1.f : 0.f;
    float2 pend_v = make_float2(0.f, 0.f);
    unsigned pend_off = 0u;
    bool pend_st = false;
    auto flush_store = [&]() {
        if (pend_st) st_at<float2>(sb.out, pend_off, pend_v);
        pend_st = false;
    };
    // o[node row][node column of the element]: contributions of the thread's two elements (.x, .y) to their 2 x 2 nodes in the plane being
    // finished.  The thread's node columns: c0 = o[.][0].x (+ the left thread's o[.][1].y), c1 = o[.][1].x + o[.][0].y; o[.][1].y goes right.
    auto emit_plane = [&](const v2f (&o)[2][2], int z, bool owned_plane, const RawNodes2* W, int zpub) {
        DN_STAMP(stamp_C);
        // keep and (E1G) the value of the own node pair in the plane being finished: re-read from the thread's own records rather than carried in
        // eight registers through the layer (the kernel sits at its register cap).  The reads are issued BEFORE this call's publish overwrites the
        // records of the same parity (a wave's LDS accesses execute in order) and land under the barrier.
        const float4 own0 = recE[z & 1][ty][tx], own1 = recO[z & 1][ty][tx];
        const v2f keep = {own0.w, own1.w}, uown = {own0.x, own1.x};
        const v2f bown = {own0.z, own1.z};            // LOADV: the load vector at the own node pair
        const float left0 = lane_from_left(o[0][1].y, from_left, nfirst);
        xch[par][tid] = make_float2(o[1][0].x + lane_from_left(o[1][1].y, from_left, nfirst), o[1][1].x + o[1][0].y);
        if (W != nullptr) plane_publish(*W, zpub);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        DN_STAMP(stamp_D);
        v2f t = {o[0][0].x + left0, o[0][1].x + o[0][0].y};
        if (ty > 0) {
            const float2 up = xch[par][tid - 16];
            t += v2f{up.x, up.y};
        }
        const bool st = owned_plane && owner && noderow_ok;
        if constexpr (LOADV) {
            // out_a -= beta * wscale * b_a,  sum W f u = sum_a u_a b_a (u after the Dirichlet conditions): one FMA each per owned node.
            // Threads beyond the mesh hold a clamped duplicate of a valid pair: excluded.
            const bool inmesh = st && x0 < p.nx;
            t = inmesh ? vfma(TV.nbw, bown, t) : t;
            e2_acc2 = inmesh ? vfma(uown, bown, e2_acc2) : e2_acc2;
        }
        if constexpr (E1G) ut_acc = st ? vfma(t, uown, ut_acc) : ut_acc;      // before the Dirichlet rows are zeroed
        t *= keep;
        sq_acc2 = st ? vfma(t, t, sq_acc2) : sq_acc2;
        pend_v = make_float2(t.x * p.out_scale, t.y * p.out_scale);
        pend_off = (unsigned)z * npl + out_row + (unsigned)x0;
        pend_st = st && sb.out != nullptr && x0 < p.nx;
        par ^= 1;
    };
    auto layer = [&](int ez, const PlaneP& L, const PlaneP& U, const RawNodes2* W) {
        const bool own_layer = ez >= ez_own;
        const v2f cnt = ((own_layer && owner) ? 1.f : 0.f) * okv;
        v2f o[2][2], tU[NGP][NGP], tX[NGP], tY[NGP], e1, e2;
        float fg[1] = {0.f};
        q1_layer_3d_w<NGP, false, HAS_F, UW, v2f>(TV, L.VU, U.VU, L.VX, U.VX, L.VY, U.VY, L.VN, U.VN, L.VF, U.VF, fg, cU, cX, cY, tU, tX, tY, e1, e2);
        if constexpr (E1G) {
            asm volatile("" : "+v"(e2));
        } else {
            asm volatile("" : "+v"(e1), "+v"(e2));
            e1_acc2 = vfma(cnt, e1, e1_acc2);
        }
        e2_acc2 = vfma(cnt, e2, e2_acc2);
        plane_transpose(tU, tX, tY, o);
        __builtin_amdgcn_sched_barrier(0);
        emit_plane(o, ez, own_layer, W, ez + 2);
    };

    // prologue: planes ez_begin and ez_begin + 1 into LDS (requested together), the lower one staged
    if (lane < 13 && wave * 13 + lane < 49) {          // the constant components of the halo records (both parities): nu = 1, f = 0, keep = 1
        float4* const r0 = (hcol & 1) ? &recO[0][hrow][hcol >> 1] : &recE[0][hrow][hcol >> 1];
        const unsigned st4 = (hcol & 1) ? 17u * 16u : 17u * 17u;
        r0[0] = make_float4(0.f, 1.f, 0.f, 1.f);
        r0[st4] = make_float4(0.f, 1.f, 0.f, 1.f);
    }
    __syncthreads();
    RawNodes2 W;
    {
        RawNodes2 W0;
        plane_request(ez_begin, W0);
        plane_request(ez_begin + 1, W);
        plane_publish(W0, ez_begin);
        plane_publish(W, ez_begin + 1);
    }
    __syncthreads();
    plane_gather(ez_begin, SA);
    __syncthreads();
    int ez = ez_begin;
#if defined(DN_STAMP3D)
    stamp_last = __builtin_amdgcn_s_memtime();
#endif
#pragma nounroll
    for (; ez + 1 < ez_end; ez += 2) {
        plane_request(ez + 2, W);                 // lands while this layer is computed; published before the layer's barrier
        flush_store();
        DN_STAMP(stamp_A);
        plane_gather(ez + 1, SB);
        DN_STAMP(stamp_B);
        layer(ez, SA, SB, &W);
        DN_STAMP(stamp_E);
        plane_request(ez + 3, W);
        flush_store();
        DN_STAMP(stamp_A);
        plane_gather(ez + 2, SA);
        DN_STAMP(stamp_B);
        layer(ez + 1, SB, SA, &W);
        DN_STAMP(stamp_E);
#if defined(DN_STAMP3D)
        stamp_n += 2;
#endif
    }
    bool odd = false;
    if (ez < ez_end) {
        flush_store();
        plane_gather(ez + 1, SB);
        layer(ez, SA, SB, nullptr);
        odd = true;
    }
    flush_store();
    if (ez_end == p.nelz) {       // the last strip owns the top boundary plane: only the layer below contributes
        v2f o[2][2];
        plane_transpose(cU, cX, cY, o);
        emit_plane(o, p.nz - 1, true, nullptr, 0);
        flush_store();
    }
#if defined(DN_STAMP3D)
    if (tid == 0) {                               // wave 0 of every workgroup (tools/stamp3d.py)
        const unsigned slot = blockIdx.x;
        if (slot < 8192u) {
            unsigned long long* d = dn_stamp_buf + slot * 8u;
            const unsigned long long hwid = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)), xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));
            d[0] = stamp_A; d[1] = stamp_B; d[2] = stamp_C; d[3] = stamp_D;
            d[4] = stamp_E | ((__builtin_amdgcn_s_memrealtime() - stamp_rt0) << 40);
            d[5] = stamp_n | (hwid << 16) | ((xcc & 0xffull) << 48);
            d[6] = stamp_t0; d[7] = __builtin_amdgcn_s_memtime();
        }
    }
#endif
    float e1_acc = e1_acc2.x + e1_acc2.y;
    const float e2_acc = e2_acc2.x + e2_acc2.y, sq_acc = sq_acc2.x + sq_acc2.y;
    if constexpr (E1G) e1_acc = ((ut_acc.x + ut_acc.y) / p.T.esc + p.T.beta * e2_acc) / p.T.alpha;
    if (p.want_sums) finish_sums(p, e1_acc, e2_acc, sq_acc, tid, 256, red, &last_flag, (double)p.T.esc);
}

// diagnostic: resident workgroups per CU the runtime grants the default 3-D kernel (tools/occ3d.py prints it)
extern "C" int dn_debug_occupancy_q1n(void) {
    int n = -1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, poisson3d_q1n_kernel<2, FL3_NU | FL3_F | FL3_BC_U8C | FL3_BC_ONE, true>, 256, 0) != hipSuccess) return -1;
    return n;
}
#endif

// ---- dispatch ---------------------------------------------------------------------------------------------
template <int NGP, int E, bool VEC, int FL>
static void launch3_one(const PoissonParams& pp, const Geom3D& g, int batch, hipStream_t s) {
    const dim3 grid((unsigned)((long long)g.chunks * g.tiles * g.strips * batch)), block(g.TX, g.TY);
    bool unit = true;                                   // the exact 2-point rule: all weights 1
    for (int i = 0; i < NGP; ++i) unit = unit && pp.T.w[i] == 1.0f;
    if constexpr (E == 1 && (FL & (FL3_FGP | FL3_BC)) == 0) {
        if (g.TX == 16 && g.TY == 16 && pp.nx >= 2 && config(CFG_Q1_3D_T16) == nullptr) {      // node-owner form (every node requested once)
            constexpr int FL1 = (FL & FL3_BC_U8C) ? (FL | FL3_BC_ONE) : FL;
            const bool one = (FL & FL3_BC_U8C) && ((pp.bc[0].mask != nullptr) != (pp.bc[1].mask != nullptr));
            if constexpr (NGP == 2) {
                if (unit) {
                    // energy from the nodal values (FL3_E1G) whenever the launch has a stiffness part and sums are wanted
                    const bool e1g = pp.T.alpha != 0.f && pp.want_sums && config(CFG_Q1_3D_E1SUM) == nullptr;
                    if (e1g) {
                        if (one) hipLaunchKernelGGL((poisson3d_q1n_kernel<NGP, FL1 | FL3_E1G, true>), grid, block, 0, s, pp, g.chunks, g.tiles, g.strips);
                        else hipLaunchKernelGGL((poisson3d_q1n_kernel<NGP, FL | FL3_E1G, true>), grid, block, 0, s, pp, g.chunks, g.tiles, g.strips);
                        return;
                    }
                    if (one) hipLaunchKernelGGL((poisson3d_q1n_kernel<NGP, FL1, true>), grid, block, 0, s, pp, g.chunks, g.tiles, g.strips);
                    else hipLaunchKernelGGL((poisson3d_q1n_kernel<NGP, FL, true>), grid, block, 0, s, pp, g.chunks, g.tiles, g.strips);
                    return;
                }
            }
            if (one) hipLaunchKernelGGL((poisson3d_q1n_kernel<NGP, FL1, false>), grid, block, 0, s, pp, g.chunks, g.tiles, g.strips);
            else hipLaunchKernelGGL((poisson3d_q1n_kernel<NGP, FL, false>), grid, block, 0, s, pp, g.chunks, g.tiles, g.strips);
            return;
        }
    }
    if constexpr (E == 1) {
        if (g.TX == 16 && pp.nx >= 2) {                // 16-wide tiles, one element per thread: DPP hand-over, paired loads
            if constexpr (NGP == 2) {
                if (unit) {
                    hipLaunchKernelGGL((poisson3d_q1w_kernel<NGP, 1, false, FL, true, true>), grid, block, 0, s, pp, g.chunks, g.tiles, g.strips);
                    return;
                }
            }
            hipLaunchKernelGGL((poisson3d_q1w_kernel<NGP, 1, false, FL, false, true>), grid, block, 0, s, pp, g.chunks, g.tiles, g.strips);
            return;
        }
    }
    if constexpr (NGP == 2) {
        if (unit) {
            hipLaunchKernelGGL((poisson3d_q1w_kernel<NGP, E, VEC, FL, true, false>), grid, block, 0, s, pp, g.chunks, g.tiles, g.strips);
            return;
        }
    }
    hipLaunchKernelGGL((poisson3d_q1w_kernel<NGP, E, VEC, FL, false, false>), grid, block, 0, s, pp, g.chunks, g.tiles, g.strips);
}

// node-owner form with fp32 mask images (constant values); only instantiated for one element per thread and nodal / absent forcing
template <int NGP, int E, bool VEC, int FL>
static void launch3_f32n(const PoissonParams& pp, const Geom3D& g, int batch, hipStream_t s) {
    if constexpr (E == 1 && (FL & FL3_FGP) == 0) {
        const dim3 grid((unsigned)((long long)g.chunks * g.tiles * g.strips * batch)), block(16, 16);
        constexpr int F2 = FL | FL3_BC_U8C | FL3_BC_F32, F1 = F2 | FL3_BC_ONE;
        const bool one = (pp.bc[0].mask != nullptr) != (pp.bc[1].mask != nullptr);
        bool unit = NGP == 2;
        for (int i = 0; i < NGP; ++i) unit = unit && pp.T.w[i] == 1.0f;
        if constexpr (NGP == 2) {
            if (unit) {
                if (one) hipLaunchKernelGGL((poisson3d_q1n_kernel<NGP, F1, true>), grid, block, 0, s, pp, g.chunks, g.tiles, g.strips);
                else hipLaunchKernelGGL((poisson3d_q1n_kernel<NGP, F2, true>), grid, block, 0, s, pp, g.chunks, g.tiles, g.strips);
                return;
            }
        }
        if (one) hipLaunchKernelGGL((poisson3d_q1n_kernel<NGP, F1, false>), grid, block, 0, s, pp, g.chunks, g.tiles, g.strips);
        else hipLaunchKernelGGL((poisson3d_q1n_kernel<NGP, F2, false>), grid, block, 0, s, pp, g.chunks, g.tiles, g.strips);
    } else {
        launch3_one<NGP, E, VEC, FL | FL3_BC>(pp, g, batch, s);
    }
}

template <int NGP, int E, bool VEC>
static void launch3_flags(const PoissonParams& pp, const Geom3D& g, int batch, hipStream_t s) {
    const int f = pp.fgp ? 2 : (pp.f ? 1 : 0);
    const bool bc = pp.bc[0].mask || pp.bc[1].mask;
    bool u8c = bc, f32c = bc;                        // all conditions uint8 / all fp32 images, constant values
    for (int k = 0; k < 2; ++k) {
        if (pp.bc[k].mask && (!pp.bc[k].mask_is_u8 || pp.bc[k].field)) u8c = false;
        if (pp.bc[k].mask && (pp.bc[k].mask_is_u8 || pp.bc[k].field)) f32c = false;
    }
    // fp32 images with constant values (the reference's masks): the node-owner form reads them as one dword per node; everything else
    // with fp32 masks or value fields goes through the generic form of the T16 kernel
    const bool f32n = f32c && E == 1 && !pp.fgp && g.TX == 16 && g.TY == 16 && config(CFG_Q1_3D_T16) == nullptr;
#define DN_L3(FLAGS)                                                                   \
    (!bc ? launch3_one<NGP, E, VEC, (FLAGS)>(pp, g, batch, s)                          \
         : u8c ? launch3_one<NGP, E, VEC, (FLAGS) | FL3_BC_U8C>(pp, g, batch, s)       \
         : f32n ? launch3_f32n<NGP, E, VEC, (FLAGS)>(pp, g, batch, s)                  \
               : launch3_one<NGP, E, VEC, (FLAGS) | FL3_BC>(pp, g, batch, s))
    if (pp.nu) {
        if (f == 0) DN_L3(FL3_NU);
        else if (f == 1) DN_L3(FL3_NU | FL3_F);
        else DN_L3(FL3_NU | FL3_FGP);
    } else {
        if (f == 0) DN_L3(0);
        else if (f == 1) DN_L3(FL3_F);
        else DN_L3(FL3_FGP);
    }
#undef DN_L3
}

#if DN_NGP == 2
template <int FLB>
static void launch3_n2_bc(const PoissonParams& pp, const dim3& grid, const Geom3D& g, hipStream_t s) {
    const bool any = pp.bc[0].mask || pp.bc[1].mask;
    const bool one = (pp.bc[0].mask != nullptr) != (pp.bc[1].mask != nullptr);
    bool f32 = false;
    for (int k = 0; k < 2; ++k)
        if (pp.bc[k].mask && !pp.bc[k].mask_is_u8) f32 = true;
    const dim3 block(16, 16);
#define DN_N2(FLAGS) hipLaunchKernelGGL((poisson3d_q1n2_kernel<(FLAGS)>), grid, block, 0, s, pp, g.chunks, g.tiles, g.strips)
    if (pp.bc[0].kind == DN_MASK_BOX || pp.bc[1].kind == DN_MASK_BOX) {       // box faces (no array), alone or beside ONE mask image
        if (!any) DN_N2(FLB | FL3_BOX);
        else if (!f32) DN_N2(FLB | FL3_BOX | FL3_BC_U8C | FL3_BC_ONE);
        else DN_N2(FLB | FL3_BOX | FL3_BC_U8C | FL3_BC_F32 | FL3_BC_ONE);
        return;
    }
    if (!any) DN_N2(FLB);
    else if (!f32 && one) DN_N2(FLB | FL3_BC_U8C | FL3_BC_ONE);
    else if (!f32) DN_N2(FLB | FL3_BC_U8C);
    else if (one) DN_N2(FLB | FL3_BC_U8C | FL3_BC_F32 | FL3_BC_ONE);
    else DN_N2(FLB | FL3_BC_U8C | FL3_BC_F32);
#undef DN_N2
}

static int launch3_n2(const PoissonParams& pp, const Geom3D& g, int batch, hipStream_t s) {
    const dim3 grid((unsigned)((long long)g.chunks * g.tiles * g.strips * batch));
    // energy from the nodal values (FL3_E1G) whenever the launch has a stiffness part and sums are wanted
    const bool e1g = pp.T.alpha != 0.f && pp.want_sums && config(CFG_Q1_3D_E1SUM) == nullptr;
    const int sel = (pp.nu ? 1 : 0) | (pp.f ? 2 : 0) | (e1g ? 4 : 0);
    if (pp.f && pp.f_is_load) {                    // the forcing as the assembled load vector (dn_poisson_args.f_is_load)
        switch (sel) {
            case 2: launch3_n2_bc<FL3_F | FL3_LOAD>(pp, grid, g, s); break;
            case 3: launch3_n2_bc<FL3_NU | FL3_F | FL3_LOAD>(pp, grid, g, s); break;
            case 6: launch3_n2_bc<FL3_F | FL3_LOAD | FL3_E1G>(pp, grid, g, s); break;
            default: launch3_n2_bc<FL3_NU | FL3_F | FL3_LOAD | FL3_E1G>(pp, grid, g, s); break;
        }
        return 0;
    }
    switch (sel) {
        case 0: launch3_n2_bc<0>(pp, grid, g, s); break;
        case 1: launch3_n2_bc<FL3_NU>(pp, grid, g, s); break;
        case 2: launch3_n2_bc<FL3_F>(pp, grid, g, s); break;
        case 3: launch3_n2_bc<FL3_NU | FL3_F>(pp, grid, g, s); break;
        case 4: launch3_n2_bc<FL3_E1G>(pp, grid, g, s); break;
        case 5: launch3_n2_bc<FL3_NU | FL3_E1G>(pp, grid, g, s); break;
        case 6: launch3_n2_bc<FL3_F | FL3_E1G>(pp, grid, g, s); break;
        default: launch3_n2_bc<FL3_NU | FL3_F | FL3_E1G>(pp, grid, g, s); break;
    }
    return 0;
}
#endif

#define DN_CAT2(a, b) a##b
#define DN_CAT(a, b) DN_CAT2(a, b)
int DN_CAT(launch_poisson3d_q1_g, DN_NGP)(const PoissonParams& pp, const Geom3D& g, int batch, bool vec, hipStream_t s) {
    if (pp.f && pp.f_is_load && !(DN_NGP == 2 && g.E == 2 && g.TX == 16 && g.TY == 16)) return DN_E_UNSUPPORTED;      // load vectors: two-element node-owner form only
    if (g.E == 1) { launch3_flags<DN_NGP, 1, false>(pp, g, batch, s); return 0; }
#if DN_NGP == 2
    if (g.E == 2 && g.TX == 16 && g.TY == 16) {
        // round 4: the closed-form-in-z kernel (poisson3d_q1_cf.hip) wherever the launch has a stiffness part; dn_config_set("Q1_3D_N2") keeps the round-3 kernel
        if (poisson3d_q1_cf_ok(pp) && (!pp.want_sums || pp.T.alpha != 0.f)) return launch_poisson3d_q1_cf(pp, g, batch, s);
        return launch3_n2(pp, g, batch, s);
    }       // node-owner form, two elements per thread (dn_poisson_apply checked its preconditions)
#endif
    return DN_E_UNSUPPORTED;
}

}  // namespace dn
