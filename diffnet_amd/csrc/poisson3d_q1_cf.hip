// 3-D Q1 fused Poisson kernel, CLOSED FORM ALONG THE MARCHED AXIS (round 4).  Default for the exact 2-point rule on
// meshes with an even number of nodes per row, nodal / absent / pre-assembled forcing and constant-value Dirichlet conditions held as
// uint8 or fp32 images or as faces of the domain box -- configs[2] and configs[3] of BASELINE.json and the slab decomposition.
//
// What it evaluates, per sample:   out_a = wscale ( alpha sum_e sum_g nu_g gradN_a . grad u_g  -  beta sum_e sum_g N_a f_g ),   rows of fixed
// nodes zeroed, the two energy sums and sum out^2 -- the loss bodies IBN_3D.py:114-136 and solve_in_object_3d.py:75-102 (5-6 x 8 single-channel
// Conv3d + cat + elementwise + backward in the reference, DiffNetFEM.py:7-18) in one pass over the nodal fields.
//
// The memory side is the node-owner form of poisson3d_q1n2_kernel (poisson3d_q1.inl): a 16 x 16 workgroup marches a tile of 32 x 16 elements
// over R element layers; a thread owns two nodes of a node row (one 8-byte load per field and plane, one 8-byte store) and the two elements to
// their right; every node is requested once per workgroup and plane and published into LDS planes (one array per field, even and odd node
// columns apart), one LDS-only barrier per layer, contributions to nodes of neighbouring threads handed over through ds_bpermute (x) and an LDS
// slot (y).  The ARITHMETIC is new.  The round-3 kernel was bound by its instruction count (~260 VALU instructions per thread and layer on
// packed fp32, VALU busy 73 % of the launch, profiles/r3_stamp3d_n2.txt); this form needs ~180 (126 packed):
//   * in-plane stages in MONOMIAL form: with u(x, y) = u00 + dx0 x + dy0 y + xy x y on an element face, the x-derivative at the y-Gauss points,
//     the y-derivative at the x-Gauss points and the value at the four in-plane points cost 14 instructions, their adjoint 22 (was 15 + 33);
//   * the x- and y-flux terms are integrated along z IN CLOSED FORM: for nu and du/dx both linear in z the sum over the two z-Gauss points of
//     (1 - t) nu u_x and t nu u_x is tau [ (A + B)(a + b) + kappa A a ] and tau [ (A + B)(a + b) + kappa B b ]  (A, B / a, b: the values on the
//     lower / upper plane, tau = sum_k t_k^2 (1 - t_k), kappa = sum_k t_k^3 / tau - 1 -- moments of the reference's own rule, exact, no new
//     quadrature); the plane-local products A a are shared by the two layers of a plane: 6 instructions per term instead of 15;
//   * the forcing term is the mass matrix M_x (x) M_y (x) M_z applied to the nodal f (f trilinear per element, rule exact): the node owner
//     applies the 3-point stencil of M_z to the plane it publishes (f is requested one plane ahead), the elements stage that plane like nu
//     (12 instructions) and add it to the cotangent of the staged u (4) -- and sum_g f_g u_g is the sum over the in-plane points of the same
//     products (5);
//   * the stiffness part of the energy comes from the finished nodal values (sum_a u_a out_a), as in round 3.
// Results differ from the per-Gauss-point kernels by fp32 rounding only (tools/q1cf3d_proto.py: the same formulas in float64 against the CPU
// oracle; tests/test_gpu_q1cf3d.py: this kernel against the oracle and against the round-3 kernels).
// Measured (profiles/r4_3d_cf_times.txt): 256^3 121 -> 95 us, 128^3 26 -> 22 us.  With the arithmetic cut down the launch is bound by its ACCESS
// PATTERN (89.6 us with every LDS access, hand-over, barrier and all arithmetic removed: pair loads 44.5 + store 22 + mask bytes 7 + halo 16) and,
// right behind it, by the per-layer chain of publish / hand-over / finish; the DN_CF3_ABL_* switches below are the timing experiments that showed it
// (their results are wrong by construction; none is defined in the library build).
#include <cmath>

#include "poisson_common.h"

namespace dn {

enum : int { CF3_NU = 1, CF3_F = 2, CF3_LOAD = 4, CF3_IMG = 8, CF3_ONE = 16, CF3_F32 = 32, CF3_BOX = 64 };
// CF3_LOAD (with CF3_F): `f` holds the assembled load vector (dn_poisson_args.f_is_load); CF3_IMG: mask images, uint8 or (CF3_F32) fp32, two or
// (CF3_ONE) one of them; CF3_BOX: at least one condition is given as faces of the domain box

struct Cf3Consts {
    float t0, t1;                      // lerp weights of the two Gauss points
    float snu;                         // scale of the nu records: kap_x * tau
    float ry, rz;                      // kap_y / kap_x, kap_z / (kap_x tau)
    float rzt0, rzt1, ryt0, ryt1;      // rz t_k, ry t_k
    float kappa, kappa2;               // sigma / tau - 1 and twice that
    float nbz;                         // -beta wscale / rz: forcing into the cotangent of the staged u
    float nbw;                         // -beta wscale
    float c00, c01, c11;               // 1-D element mass matrix of the rule
    float czd[4];                      // middle coefficient of the z mass stencil: interior plane, bottom plane, top plane, both (a mesh of one layer)
    float inv_esc, beta, inv_alpha;
};

#ifndef DN_CF3_PF
#define DN_CF3_PF 1                   // planes in flight per thread (raw register sets): 1, or 2 (9 VGPRs more; measured equal: profiles/r4_3d_cf_times.txt)
#endif
#ifndef DN_Q1CF_WAVES
#define DN_Q1CF_WAVES 3               // waves per SIMD asked of the compiler (<= 168 VGPRs)
#endif

#if defined(DN_STAMP3D)
// Diagnostic build only (tools/stamp3d.py): per-wave cycle budget of the loop, accumulated in scalar registers with s_memtime and written once
// at the end of the kernel by wave 0 of every workgroup.  Phases per layer: A = request the next plane + deferred store, B = gather the upper
// plane from LDS + stage it, C = layer arithmetic + adjoint, D = hand-over + publish + barrier, E = hand-over read + finish the node values.
__device__ unsigned long long dn_stamp_buf[8192 * 8];
extern "C" int dn_debug_stamps(void* dst, size_t bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(dn_stamp_buf), bytes); }
#define DN_STAMP(var) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t__ = __builtin_amdgcn_s_memtime(); var += t__ - stamp_last; stamp_last = t__; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define DN_STAMP(var) do { } while (0)
#endif

// lane l <- lane l - 1 of the wave, 0 for the first lane of a row of 16.  ds_bpermute, not DPP: on gfx950 a DPP move costs a shared SIMD ~33
// cycles, a ds_bpermute ~3 (tools/micro/valu_mem.hip, profiles/r2_valu_mem.txt).
__device__ __forceinline__ float cf3_from_left(float v, int from, float nf) {
    return nf * __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(from, __builtin_bit_cast(int, v)));
}

// wave-uniform float select on the scalar unit: (a < b) ? x : y for wave-uniform integers a, b.  Inline asm: written in C++ (with floats or with
// their bit patterns) the compiler moves both values into vector registers, selects there and reads the result back (4 VALU instructions)
__device__ __forceinline__ float cf3_usel_lt(int a, int b, float x, float y) {
    int r;
    asm("s_cmp_lt_i32 %1, %2\n\ts_cselect_b32 %0, %3, %4"
        : "=s"(r)
        : "s"(__builtin_amdgcn_readfirstlane(a)), "s"(__builtin_amdgcn_readfirstlane(b)), "s"(__builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x))),
          "s"(__builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, y)))
        : "scc");
    return __builtin_bit_cast(float, r);
}

template <int FL>
__global__ void __launch_bounds__(256, DN_Q1CF_WAVES) poisson3d_q1_cf_kernel(const PoissonParams p, const Cf3Consts k, const int chunks_x, const int tiles_y, const int strips_z) {
    constexpr bool HAS_NU = (FL & CF3_NU) != 0, F_ARR = (FL & CF3_F) != 0, LOADV = F_ARR && (FL & CF3_LOAD) != 0, HAS_F = F_ARR && !LOADV;
    constexpr bool IMG = (FL & CF3_IMG) != 0, BOX = (FL & CF3_BOX) != 0;
    constexpr int NMASK = !IMG ? 0 : ((FL & CF3_ONE) ? 1 : 2);
    constexpr bool MASK_F32 = (FL & CF3_F32) != 0;
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int tid = ty * 16 + tx;
    // 1-D grid with an XCD-aware decode: workgroups are dealt round-robin to the 8 XCDs, each with its own L2; every XCD gets a contiguous range of
    // the logical order (chunk fastest, then tile, strip, sample), so a tile's halo is read from HBM once instead of once per XCD
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
    // elements beyond the mesh (ragged last chunk / tile) are computed on clamped, finite node values and their results multiplied by 0
    const v2f zero2 = {0.f, 0.f};

    const v2f okv = {(ey < p.nely && x0 < p.nelx) ? 1.f : 0.f, (ey < p.nely && x0 + 1 < p.nelx) ? 1.f : 0.f};
    const v2f okown = owner ? okv : zero2;         // elements this workgroup counts in the element sums
#if defined(DN_STAMP3D)
    unsigned long long stamp_A = 0, stamp_A1 = 0, stamp_B = 0, stamp_C = 0, stamp_D = 0, stamp_E = 0, stamp_n = 0, stamp_last = 0;
    const unsigned long long stamp_t0 = __builtin_amdgcn_s_memtime(), stamp_rt0 = __builtin_amdgcn_s_memrealtime();
#endif

    // The node planes in LDS: ONE block, two slots; per slot 17 node rows, per row the three fields one after the other, per field the EVEN node
    // columns (17 floats, padded to 18) followed by the ODD ones (16, padded to 18).  A thread's element pair needs the node pairs (x0, x0 + 1)
    // = (E[tx], O[tx]) and (x0 + 1, x0 + 2) = (O[tx], E[tx + 1]) of two node rows: each is ONE ds_read2_b32 whose two passes read consecutive floats
    // in consecutive lanes (no bank conflicts) and whose result is the register pair the packed instructions take -- the 16-byte node records of
    // round 3 cost 24 v_mov per layer to form those pairs, plain rows of nodes a two-way bank conflict on every second read.  Every access of a
    // thread is one of two per-thread byte offsets (its node column in a row, its pair slot) plus a compile-time constant: the plane slots alternate
    // with the layers of the two-layer loop body and are literals there; all of a gather's offsets lie within the reach of the read2 offset fields.
    constexpr int HALF = 18, LP = 2 * HALF, ROW = 3 * LP + 4, PLANE = 17 * ROW;      // (rows of 112 floats: consecutive thread rows land on the other 16 banks)
    constexpr int OFF_U = 0, OFF_N = LP, OFF_F = 2 * LP, OFF_K = 2 * PLANE, OFF_X = OFF_K + 4 * 512, LDS_FLOATS = OFF_X + 2 * 512;
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];      // u' | nu * snu | f' or the load vector, per row;  keep of the own pairs;  hand-over slots
    typedef __attribute__((address_space(3))) float* lds_fptr;
    lds_fptr const lds_own = (lds_fptr)(lds + ty * ROW + tx);           // E[tx] of the thread's node row (field 0, slot 0); O[tx] is HALF floats on
    char* const lds_pair = reinterpret_cast<char*>(lds) + tid * 8;      // the thread's float2 in the keep / hand-over arrays
    // (per slot: the base of the gather's inline-asm reads)
    lds_fptr const lds_own_b0 = lds_own, lds_own_b1 = lds_own + PLANE;
    auto lds_rd2 = [&](lds_fptr base, int off_a, int off_b) { return v2f{base[off_a], base[off_b]}; };          // -> ds_read2_b32
    auto lds_wr2 = [&](lds_fptr base, int off_a, int off_b, float a, float b) { base[off_a] = a; base[off_b] = b; };   // -> ds_write2_b32
    auto lds_ld2 = [&](const char* base, int float_off) {
        const float2 t = *reinterpret_cast<const float2*>(base + float_off * 4);
        return v2f{t.x, t.y};
    };
    auto lds_st2 = [&](char* base, int float_off, float a, float b) { *reinterpret_cast<float2*>(base + float_off * 4) = make_float2(a, b); };
    __shared__ double red[2 * (256 / 64)];
    __shared__ int last_flag;
    if (blockIdx.x == 0u) fold_prev_sums(p, tid, 256, red);       // dn_poisson_args.fold_prev: close the evaluation before this one

    // own pair (clamped into the mesh: nx is even, so a pair is inside or outside as a whole) and the halo node this thread fetches
    const unsigned own_off = (unsigned)min(ey, p.ny - 1) * (unsigned)p.nx + (unsigned)min(x0, p.nx - 2);
    const int lane = tid & 63, wave = tid >> 6;
    // 49 halo nodes (node row 16: 33 nodes, node column 32: 16 nodes), 13 per wave, ONE load instruction for their u, nu and f: lanes 0..12 fetch
    // u, 13..25 nu, 26..38 f of the wave's 13 nodes through per-lane 64-bit addresses; the other lanes repeat lane 38
    const int hgrp = min(lane / 13, 2), hsub = min(lane - 13 * hgrp, 12);
    const int hidx = min(wave * 13 + hsub, 48);
    const int hrow = hidx < 33 ? 16 : hidx - 33, hcol = hidx < 33 ? hidx : 32;
    const unsigned halo_off = (unsigned)min(ny0 + hrow, p.ny - 1) * (unsigned)p.nx + (unsigned)min(nx0 + hcol, p.nx - 1);
    const bool halo_lane = lane < 39 && wave * 13 + hsub < 49 && (hgrp == 0 || (hgrp == 1 ? HAS_NU : HAS_F));       // (a load vector is needed at owned nodes only)
    float* const halo_lds = lds + (hgrp == 0 ? OFF_U : (hgrp == 1 ? OFF_N : OFF_F)) + hrow * ROW + (hcol & 1) * HALF + (hcol >> 1);    // the lane's field, slot 0
    const float* const halo_src = (hgrp == 1 && HAS_NU) ? sb.nu : ((hgrp == 2 && HAS_F) ? sb.f : sb.u);                               // per lane
    const bool halo_is_f = HAS_F && hgrp == 2;
    const unsigned halo_fmask = halo_is_f ? 0xffffffffu : 0u;
    const float halo_mul = (HAS_NU && hgrp == 1) ? k.snu : 1.f;

    const bool has_mask[2] = {sb.mask[0] != nullptr, sb.mask[1] != nullptr};
    const uint8_t* mask8[2];
    mask8[0] = reinterpret_cast<const uint8_t*>(has_mask[0] ? sb.mask[0] : sb.mask[1]);      // an absent condition re-reads the other one and is
    mask8[1] = reinterpret_cast<const uint8_t*>(has_mask[1] ? sb.mask[1] : sb.mask[0]);      // ignored: no load inside a wave-uniform branch
    const float* mask32[2] = {reinterpret_cast<const float*>(mask8[0]), reinterpret_cast<const float*>(mask8[1])};
    // BOX: per condition the faces of the domain box it fixes (0: not a box condition); in-plane part per node, constant over the march
    const int bfaces[2] = {BOX && p.bc[0].kind == DN_MASK_BOX ? p.bc[0].box_faces : 0, BOX && p.bc[1].kind == DN_MASK_BOX ? p.bc[1].box_faces : 0};
    auto box_xy = [&](int kk, int x, int y) {
        return ((bfaces[kk] & DN_FACE_XLO) && x == 0) || ((bfaces[kk] & DN_FACE_XHI) && x == p.nx - 1) || ((bfaces[kk] & DN_FACE_YLO) && y == 0) ||
               ((bfaces[kk] & DN_FACE_YHI) && y == p.ny - 1);
    };
    const bool bxy0[2] = {box_xy(0, x0, ey), box_xy(1, x0, ey)}, bxy1[2] = {box_xy(0, x0 + 1, ey), box_xy(1, x0 + 1, ey)};
    const bool bxyh[2] = {box_xy(0, nx0 + hcol, ny0 + hrow), box_xy(1, nx0 + hcol, ny0 + hrow)};
    // box faces without a mask image: what the in-plane faces do to a node is the same on every plane -- "fixed", and to which value (the LAST condition
    // that names one of its faces), worked out once.  Planes on a z face of a condition (two per mesh at most; wave-uniform) take the general path.
    auto box_fix = [&](const bool (&bx)[2], float& val) {
        val = bx[1] ? p.bc[1].value : p.bc[0].value;
        return bx[0] || bx[1];
    };
    float bval0 = 0.f, bval1 = 0.f, bvalh = 0.f;
    const bool bfix0 = box_fix(bxy0, bval0), bfix1 = box_fix(bxy1, bval1), bfixh = box_fix(bxyh, bvalh);
    const float bkeep0 = bfix0 ? 0.f : 1.f, bkeep1 = bfix1 ? 0.f : 1.f;

    struct RawNodes {
        v2f u, n, f;                  // own pair (f: one plane AHEAD of u and nu when it is nodal forcing)
        float h;                      // halo node: u, nu or f by lane group
        uint16_t m[2];                // uint8 masks of the own pair (two bytes), per condition
        v2f mf[2];                    // fp32 masks of the own pair
        uint8_t hm[2];
        float hmf[2];
    };
    auto ld_pair = [&](const float* base, unsigned off) {
        const float2 t = ld_at<float2>(base, off);
        return v2f{t.x, t.y};
    };
    // z-offsets (nodes) of plane z and, for nodal forcing, of the plane after it -- clamped into the mesh
    auto plane_request = [&](int zreq, RawNodes& W) {
        const unsigned zoff = (unsigned)min(zreq, p.nz - 1) * npl;
        const unsigned zoff_f = HAS_F ? (unsigned)min(zreq + 1, p.nz - 1) * npl : zoff;
        const unsigned oo = zoff + own_off, oh = zoff + halo_off + (halo_fmask & (zoff_f - zoff));
        W.u = ld_pair(sb.u, oo);
#ifdef DN_CF3_ABL_REQ                      // timing experiment only (results are wrong): u is the one array requested, the other values are made up from it
        W.n = W.u + 0.5f; W.f = W.u; W.h = W.u.x;
        W.m[0] = W.m[1] = (uint16_t)(W.u.x == 123.456f); W.hm[0] = W.hm[1] = (uint8_t)W.m[0];
        W.mf[0] = W.mf[1] = 0.f; W.hmf[0] = W.hmf[1] = 0.f;
        return;
#endif
        if constexpr (HAS_NU) W.n = ld_pair(sb.nu, oo);
        if constexpr (F_ARR) W.f = ld_pair(sb.f, zoff_f + own_off);
#ifdef DN_CF3_ABL_HALO                     // timing experiment only (results are wrong): no halo requests
        W.h = W.u.x; W.hm[0] = W.hm[1] = 0; W.hmf[0] = W.hmf[1] = 0.f;
#else
        W.h = halo_src[oh];
#endif
        if constexpr (IMG) {
            const unsigned ohm = zoff + halo_off;
#pragma unroll
            for (int kk = 0; kk < NMASK; ++kk) {
#ifdef DN_CF3_ABL_MASK                     // timing experiment only (results are wrong): no mask requests for the own pair
                W.m[kk] = (uint16_t)(W.u.x == 123.456f); W.mf[kk] = 0.f;
#else
                if constexpr (MASK_F32) W.mf[kk] = ld_pair(mask32[kk], oo);
                else W.m[kk] = ld_at<uint16_t>(mask8[kk], oo);
#endif
#ifndef DN_CF3_ABL_HALO
                if constexpr (MASK_F32) W.hmf[kk] = ld_at<float>(mask32[kk], ohm);
                else W.hm[kk] = ld_at<uint8_t>(mask8[kk], ohm);
#endif
            }
        }
    };
    // Dirichlet conditions of one node: set[k]: the node is fixed by condition k (its mask image and / or its box faces).  Applied in their order
    // (the reference's torch.where lines, IBN_3D.py:119-122): uu <- value, keep <- 0
    const float bcv1 = has_mask[0] ? p.bc[0].value : p.bc[1].value;      // NMASK == 1, no box faces: the value of the one condition
    auto fix_node = [&](float& uu, float& keep, const bool (&img)[2], const bool (&box)[2]) {
        keep = 1.f;
        if constexpr (NMASK == 1 && !BOX) {
            uu = img[0] ? bcv1 : uu;
            keep = img[0] ? 0.f : 1.f;
        } else if constexpr (IMG || BOX) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bool sset = false;
                if constexpr (NMASK == 2) sset = has_mask[kk] && img[kk];
                if constexpr (NMASK == 1) sset = has_mask[kk] && img[0];
                if constexpr (BOX) sset = sset || box[kk];
                uu = sset ? p.bc[kk].value : uu;
                keep = sset ? 0.f : keep;
            }
        }
    };
    // nodal forcing: the owner keeps the two planes before the one in W (own pair, and its halo node in the f lanes) and publishes
    //     f'(m) = czl f(m - 1) + czd f(m) + czu f(m + 1),      the 3-point stencil of the rule's 1-D mass matrix along z (boundary planes: one side)
    v2f fh0 = 0.f, fh1 = 0.f;
    float hh0 = 0.f, hh1 = 0.f;
    auto plane_publish = [&](const RawNodes& W, int zpl, int slot) {
        bool s0[2] = {false, false}, s1[2] = {false, false}, sh[2] = {false, false};
        if constexpr (IMG) {
#pragma unroll
            for (int kk = 0; kk < NMASK; ++kk) {
                if constexpr (MASK_F32) { s0[kk] = W.mf[kk].x > 0.5f; s1[kk] = W.mf[kk].y > 0.5f; sh[kk] = W.hmf[kk] > 0.5f; }
                else { s0[kk] = (W.m[kk] & 0xffu) != 0; s1[kk] = (W.m[kk] >> 8) != 0; sh[kk] = W.hm[kk] != 0; }
            }
        }
        bool b0[2] = {false, false}, b1[2] = {false, false}, bh[2] = {false, false};
        const int zc = min(zpl, p.nz - 1);
        if constexpr (BOX) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const bool zf = ((bfaces[kk] & DN_FACE_ZLO) && zc == 0) || ((bfaces[kk] & DN_FACE_ZHI) && zc == p.nz - 1);
                b0[kk] = bxy0[kk] || zf; b1[kk] = bxy1[kk] || zf; bh[kk] = bxyh[kk] || zf;
            }
        }
        v2f fz = 0.f;
        float hv = W.h * halo_mul;
        if constexpr (HAS_F) {
            // c01 (f(m-1) + f(m+1)) + czd f(m), czd by the plane's position (Cf3Consts::czd; wave-uniform)
            const float czd = cf3_usel_lt(0, zc, cf3_usel_lt(zc, p.nz - 1, k.czd[0], k.czd[2]), cf3_usel_lt(zc, p.nz - 1, k.czd[1], k.czd[3]));
            fz = vfma(czd, fh1, k.c01 * (fh0 + W.f));
            fh0 = fh1; fh1 = W.f;
            const float hz = fmaf(czd, hh1, k.c01 * (hh0 + W.h));
            hh0 = hh1; hh1 = W.h;
            hv = halo_is_f ? hz : hv;
        } else if constexpr (LOADV) {
            fz = W.f;
        }
        float u0 = W.u.x, u1 = W.u.y, k0, k1;
        bool box_fast = false;
        if constexpr (BOX && !IMG) {
            const bool zf0 = ((bfaces[0] & DN_FACE_ZLO) && zc == 0) || ((bfaces[0] & DN_FACE_ZHI) && zc == p.nz - 1);
            const bool zf1 = ((bfaces[1] & DN_FACE_ZLO) && zc == 0) || ((bfaces[1] & DN_FACE_ZHI) && zc == p.nz - 1);
            box_fast = !(zf0 || zf1);                        // (wave-uniform)
        }
        if (box_fast) {
            u0 = bfix0 ? bval0 : u0; k0 = bkeep0;
            u1 = bfix1 ? bval1 : u1; k1 = bkeep1;
        } else {
            fix_node(u0, k0, s0, b0);
            fix_node(u1, k1, s1, b1);
        }
        lds_wr2(lds_own, slot * PLANE + OFF_U, slot * PLANE + OFF_U + HALF, u0, u1);
        if constexpr (HAS_NU) {
            const v2f nr = k.snu * W.n;
            lds_wr2(lds_own, slot * PLANE + OFF_N, slot * PLANE + OFF_N + HALF, nr.x, nr.y);
        }
        if constexpr (F_ARR) lds_wr2(lds_own, slot * PLANE + OFF_F, slot * PLANE + OFF_F + HALF, fz.x, fz.y);
        if constexpr (IMG) *reinterpret_cast<float2*>(lds_pair + (OFF_K + (zpl & 3) * 512) * 4) = make_float2(k0, k1);      // (box faces alone: keep is known per thread)
        if (halo_lane) {
            if constexpr (IMG || BOX) {
                float kh;
                float hu = hv;
                if (box_fast) hu = bfixh ? bvalh : hu;
                else fix_node(hu, kh, sh, bh);
                hv = hgrp == 0 ? hu : hv;
            }
            halo_lds[slot * PLANE] = hv;
        }
    };

    // From here on a value of type v2f holds the same quantity of the thread's two elements (.x: element at node column x0, .y: at x0 + 1) or of its
    // two nodes: the element arithmetic runs on packed fp32 instructions, one per pair.
    // In-plane stage of one node plane, per element (t = lerp coordinate on the element face, Gauss points t0, t1):
    //   BX[j] = du/dx at y = t_j (constant in x),  CY[i] = du/dy at x = t_i,  U[j][i] = u(t_i, t_j),  V[j][i] = nu'(t_i, t_j)
    struct PlaneC { v2f BX[2], CY[2], U[2][2], V[2][2], own; };       // own: u' at the thread's own node pair (= the element pair's lower-left nodes)
    auto stage_vals = [&](const v2f v00, const v2f v10, const v2f v01, const v2f v11, v2f (&G)[2][2]) {
        const v2f dx0 = v10 - v00, dx1 = v11 - v01, dy0 = v01 - v00, xy = dx1 - dx0;
        const v2f A0 = vfma(k.t0, dy0, v00), A1 = vfma(k.t1, dy0, v00), B0 = vfma(k.t0, xy, dx0), B1 = vfma(k.t1, xy, dx0);
        G[0][0] = vfma(k.t0, B0, A0); G[0][1] = vfma(k.t1, B0, A0);
        G[1][0] = vfma(k.t0, B1, A1); G[1][1] = vfma(k.t1, B1, A1);
    };
    // The four node pairs of the thread's two elements in one field plane: (x0, x0 + 1), (x0 + 1, x0 + 2) of node rows ty and ty + 1, three fields.
    // gather_issue only ISSUES the reads (right after the barrier that published the plane, a whole finish + publish + request phase before they are
    // needed); gather_stage waits for them field by field and stages.
    // Inline asm: the compiler sorts the two offsets of a ds_read2_b32 it forms by address, which delivers (O[tx], E[tx + 1]) swapped and costs two
    // v_mov per pair.  It does not count asm LDS accesses, so the waits are written here: "at most N LGKM operations outstanding".  LDS accesses
    // complete in order, so once at most N are outstanding every LDS access but the last N issued has landed, whatever else (scalar loads) is
    // counted with them: the waits below name how many of the gather's own reads were issued AFTER the ones they wait for.
    struct RawPairs { v2f u00, u10, u01, u11, n00, n10, n01, n11, f00, f10, f01, f11; };
    RawPairs RP;
#define CF3_PAIRS(off, v00, v10, v01, v11)                                                                                                   \
    asm volatile("ds_read2_b32 %0, %4 offset0:%5 offset1:%6\n\tds_read2_b32 %1, %4 offset0:%6 offset1:%7\n\t"                                   \
                 "ds_read2_b32 %2, %4 offset0:%8 offset1:%9\n\tds_read2_b32 %3, %4 offset0:%9 offset1:%10"                                       \
                 : "=&v"(v00), "=&v"(v10), "=&v"(v01), "=&v"(v11)                                                                                \
                 : "v"(bu), "n"(off), "n"((off) + HALF), "n"((off) + 1), "n"((off) + ROW), "n"((off) + ROW + HALF), "n"((off) + ROW + 1))
    auto gather_issue = [&](int slot) {
        const unsigned bu = (unsigned)(uintptr_t)(slot ? lds_own_b1 : lds_own_b0);
        CF3_PAIRS(OFF_U, RP.u00, RP.u10, RP.u01, RP.u11);
        if constexpr (HAS_NU) CF3_PAIRS(OFF_N, RP.n00, RP.n10, RP.n01, RP.n11);
        if constexpr (HAS_F) CF3_PAIRS(OFF_F, RP.f00, RP.f10, RP.f01, RP.f11);
    };
#undef CF3_PAIRS
    auto gather_stage = [&](PlaneC& S, v2f (&F)[2][2]) {
        constexpr int LATER = 0;       // (LDS accesses the wave issues after the gather's may or may not exist: none are assumed)
        constexpr int NF = 1 + (HAS_NU ? 1 : 0) + (HAS_F ? 1 : 0);
#define CF3_LDS_WAIT(N, q0, q1, q2, q3) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "n"((N) > 15 ? 15 : (N)))
        CF3_LDS_WAIT(4 * (NF - 1) + LATER, RP.u00, RP.u10, RP.u01, RP.u11);
        {
            const v2f u00 = RP.u00, u10 = RP.u10, u01 = RP.u01, u11 = RP.u11;
            S.own = u00;
            const v2f dx0 = u10 - u00, dx1 = u11 - u01, dy0 = u01 - u00, xy = dx1 - dx0;
            S.BX[0] = vfma(k.t0, xy, dx0); S.BX[1] = vfma(k.t1, xy, dx0);
            S.CY[0] = vfma(k.t0, xy, dy0); S.CY[1] = vfma(k.t1, xy, dy0);
            const v2f A0 = vfma(k.t0, dy0, u00), A1 = vfma(k.t1, dy0, u00);
            S.U[0][0] = vfma(k.t0, S.BX[0], A0); S.U[0][1] = vfma(k.t1, S.BX[0], A0);
            S.U[1][0] = vfma(k.t0, S.BX[1], A1); S.U[1][1] = vfma(k.t1, S.BX[1], A1);
        }
        if constexpr (HAS_NU) {
            CF3_LDS_WAIT((HAS_F ? 4 : 0) + LATER, RP.n00, RP.n10, RP.n01, RP.n11);
            stage_vals(RP.n00, RP.n10, RP.n01, RP.n11, S.V);
        }
        if constexpr (HAS_F) {
            CF3_LDS_WAIT(LATER, RP.f00, RP.f10, RP.f01, RP.f11);
            stage_vals(RP.f00, RP.f10, RP.f01, RP.f11, F);
        }
#undef CF3_LDS_WAIT
        __builtin_amdgcn_sched_barrier(0);
    };
    PlaneC SA, SB;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) SA.V[j][i] = SB.V[j][i] = k.snu;       // nu absent: the constant field 1

    // carried cotangents of the plane between two layers, from the layer below: cX / cY of BX / CY (closed-form sums, with the plane's own
    // product kappa * PX * BX counted twice -- once for each of its layers; a boundary plane takes one back), cU of U
    v2f cX[2], cY[2], cU[2][2];
    float e2_acc = 0.f, sq_acc = 0.f, ut_acc = 0.f;      // per-thread sums over both elements / nodes (one register each: the kernel sits at its register cap)

    // adjoint of the in-plane stage: cotangents of BX, CY (up to the factors 1 and ry), U (up to rz) -> contributions to each element's 2 x 2 nodes
    auto plane_adjoint = [&](const v2f (&GX)[2], const v2f (&GY)[2], const v2f (&GU)[2][2], v2f (&o)[2][2]) {
        const v2f gA0 = k.rz * (GU[0][0] + GU[0][1]), gA1 = k.rz * (GU[1][0] + GU[1][1]);
        const v2f gB0 = vfma(k.rzt1, GU[0][1], vfma(k.rzt0, GU[0][0], GX[0]));
        const v2f gB1 = vfma(k.rzt1, GU[1][1], vfma(k.rzt0, GU[1][0], GX[1]));
        const v2f g_u00 = gA0 + gA1;
        const v2f g_dy0 = vfma(k.t1, gA1, vfma(k.t0, gA0, k.ry * (GY[0] + GY[1])));
        const v2f g_dx0 = gB0 + gB1;
        const v2f g_xy = vfma(k.ryt1, GY[1], vfma(k.ryt0, GY[0], vfma(k.t1, gB1, k.t0 * gB0)));
        const v2f o10 = g_dx0 - g_xy, o01 = g_dy0 - g_xy;
        o[0][1] = okv * o10; o[1][1] = okv * g_xy; o[1][0] = okv * o01; o[0][0] = okv * (g_u00 - o10 - g_dy0);
    };

    const unsigned out_base = (unsigned)ey * (unsigned)p.nx + (unsigned)x0;     // the own pair's offset in a plane of the output
    const int from_left = (int)(((unsigned)tid - 1u) & 63u) << 2;
    const float nfirst = tx > 0 ? 1.f : 0.f;
    float2 pend_v = make_float2(0.f, 0.f);
#ifdef DN_CF3_ABL_PUB
    float abl_sink = 0.f;
#endif
    unsigned pend_off = 0u;
    bool pend_st = false;
    // (The store as one asm block that selects the owning lanes through the exec mask -- no branch, so that the compiler's vmcnt waits for the raw
    // plane's loads stay exact instead of vmcnt(0) -- was built and measured, with one and two planes in flight: equal, profiles/r4_3d_cf_times.txt.)
    auto flush_store = [&]() {
#ifdef DN_CF3_ABL_STORE                    // timing experiment only (results are wrong): nothing is stored
        if (pend_st && pend_v.x == 123.456f) st_at<float2>(sb.out, pend_off, pend_v);
#else
        if (pend_st) st_at<float2>(sb.out, pend_off, pend_v);
#endif
        pend_st = false;
    };
    // One element layer between the staged plane L (lower, plane ez, LDS slot zslot) and the plane above (slot 1 - zslot), whose node pairs were
    // requested from LDS right after the previous layer's barrier.  Order of a layer (every step overlaps the latencies of the ones before it):
    //   1  request the node pairs of the upper plane from LDS (asm, not waited for yet); read the own pair's load vector of plane ez
    //   2  publish plane ez + 2 (requested from memory two layers ago) into slot zslot: every thread gathered plane ez before the previous barrier
    //   3  request plane ez + 4 from memory into the raw set just published; the deferred store of plane ez - 1
    //   4  stage the upper plane (its LDS reads have had steps 2-3 to land), element arithmetic, adjoint of the in-plane stage
    //   5  hand-over: right-hand contributions to the lane on the right (ds_bpermute), the upper node row's to the thread above (LDS slot)
    //   6  ONE barrier (LDS only): it publishes plane ez + 2 and the hand-over slots
    //   7  read the hand-over slot and the own pair's keep
    //   8  finish the own node pair of plane ez: sums, Dirichlet rows, value to store
    // o[node row][node column of the element]: contributions of the thread's two elements (.x, .y) to their 2 x 2 nodes.  The thread's node columns:
    // c0 = o[.][0].x (+ the left thread's o[.][1].y), c1 = o[.][1].x + o[.][0].y; o[.][1].y goes right.
    // (the own pair's u' is the first pair of the plane's gather, carried from there; keep lies in a four-slot array of its own -- plane & 3 -- and is
    // read when it is needed; only the load vector has to be taken from the plane's slot before step 2)
    struct OwnVals { v2f b; };
    auto own_read = [&](int zslot, OwnVals& ov) {
        ov.b = v2f{0.f, 0.f};
        if constexpr (LOADV) ov.b = lds_rd2(lds_own, zslot * PLANE + OFF_F, zslot * PLANE + OFF_F + HALF);       // the load vector at the own node pair
    };
    // steps 5-8
    auto emit_plane = [&](const v2f (&o)[2][2], const v2f uown, const OwnVals& ov, int z, int zslot, bool owned_plane) {
        DN_STAMP(stamp_C);
#ifdef DN_CF3_ABL_XCH                      // timing experiment only (results are wrong): no lane exchange, no hand-over slot
        const float left0 = o[0][1].y;
#else
        const float left0 = cf3_from_left(o[0][1].y, from_left, nfirst);
        lds_st2(lds_pair, OFF_X + zslot * 512, o[1][0].x + cf3_from_left(o[1][1].y, from_left, nfirst), o[1][1].x + o[1][0].y);
#endif
#ifdef DN_CF3_ABL_BAR                      // timing experiment only (results are wrong): no workgroup barrier
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
        DN_STAMP(stamp_D);
        v2f up = {0.f, 0.f};
#ifndef DN_CF3_ABL_XCH
        if (ty > 0) up = lds_ld2(lds_pair, OFF_X + zslot * 512 - 32);       // the hand-over of the thread one node row below (tid - 16)
#endif
        v2f keep = {1.f, 1.f};
        if constexpr (IMG) keep = lds_ld2(lds_pair + (z & 3) * 2048, OFF_K);
        if constexpr (BOX && !IMG) {           // in-plane faces: per thread; a plane on a fixed z face: every node
            const bool zfix = (((bfaces[0] | bfaces[1]) & DN_FACE_ZLO) && z == 0) || (((bfaces[0] | bfaces[1]) & DN_FACE_ZHI) && z == p.nz - 1);
            const float zk = cf3_usel_lt(0, zfix ? 1 : 0, 0.f, 1.f);
            keep = v2f{bkeep0 * zk, bkeep1 * zk};
        }
        v2f t = {o[0][0].x + left0, o[0][1].x + o[0][0].y};
        t += up;
        const bool st = owned_plane && owner && noderow_ok;
        // the sums take the owned nodes only: factor 1 / 0 (threads beyond the mesh in x hold clamped duplicates with contributions 0; the load
        // vector's share needs the explicit test)
        const float stf = st ? 1.f : 0.f;
        if constexpr (LOADV) {
            // out_a -= beta * wscale * b_a,  sum f u = sum_a u_a b_a (u after the Dirichlet conditions): one FMA each per owned node
            const float inm = (st && x0 < p.nx) ? 1.f : 0.f;
            const v2f bm = inm * ov.b;
            t = vfma(k.nbw, bm, t);
            const v2f ub = uown * bm;
            e2_acc += ub.x + ub.y;
        }
        t *= stf;
        const v2f tu = t * uown;               // sum_a u_a out_a, before the Dirichlet rows are zeroed
        ut_acc += tu.x + tu.y;
        if constexpr (IMG || BOX) t *= keep;
        sq_acc = fmaf(t.x, t.x, fmaf(t.y, t.y, sq_acc));
        pend_v = make_float2(t.x * p.out_scale, t.y * p.out_scale);
#ifdef DN_CF3_ABL_PUB
        pend_v.x += abl_sink;
#endif
        pend_off = (unsigned)z * npl + out_base;
        pend_st = st && sb.out != nullptr && x0 < p.nx;
    };
    // W: the raw set holding plane ez + 2 (nullptr: the strip's last layers publish and request nothing)
    auto layer = [&](int ez, int zslot, const PlaneC& L, PlaneC& Up, RawNodes* W) {
        const bool own_layer = ez >= ez_own;
        // (the reads are issued and waited for inside ONE layer: carried over the loop's back edge, their target registers could be copied by the
        // compiler before the data has landed -- it does not know that an asm LDS read completes later)
#ifndef DN_CF3_ABL_GATHER                  // timing experiment only (with DN_CF3_ABL_MATH; results are wrong): no LDS gather
        gather_issue(1 - zslot);
#endif
        OwnVals ov;
        own_read(zslot, ov);
        if (W != nullptr) {
#ifndef DN_CF3_ABL_PUB                     // timing experiment only (results are wrong): nothing is published into LDS -- the loaded values only feed the stored value
            plane_publish(*W, ez + 2, zslot);
#else
            abl_sink += W->u.x + W->u.y + W->h;
            if constexpr (HAS_NU) abl_sink += W->n.x + W->n.y;
            if constexpr (F_ARR) abl_sink += W->f.x + W->f.y;
            if constexpr (IMG && !MASK_F32) abl_sink += (float)(W->m[0] + W->hm[0]);
#endif
            DN_STAMP(stamp_A1);
            plane_request(ez + 2 + DN_CF3_PF, *W);
        }
        flush_store();
        DN_STAMP(stamp_A);
        v2f F[2][2];
#ifdef DN_CF3_ABL_MATH                     // timing experiment only (results are wrong): the memory side alone -- every load, LDS access, hand-over, barrier and store, no element arithmetic
        {
#ifdef DN_CF3_ABL_GATHER
            RP.u00 = RP.u10 = RP.u01 = RP.u11 = RP.n00 = RP.n10 = RP.n01 = RP.n11 = RP.f00 = RP.f10 = RP.f01 = RP.f11 = L.own;
#endif
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(RP.u00), "+v"(RP.u10), "+v"(RP.u01), "+v"(RP.u11));
            v2f acc = RP.u00 + RP.u10 + RP.u01 + RP.u11;
            if constexpr (HAS_NU) { asm volatile("" : "+v"(RP.n00), "+v"(RP.n10), "+v"(RP.n01), "+v"(RP.n11)); acc += RP.n00 + RP.n10 + RP.n01 + RP.n11; }
            if constexpr (HAS_F) { asm volatile("" : "+v"(RP.f00), "+v"(RP.f10), "+v"(RP.f01), "+v"(RP.f11)); acc += RP.f00 + RP.f10 + RP.f01 + RP.f11; }
            v2f oa[2][2] = {{acc, acc * 0.5f}, {acc * 0.25f, acc * 0.125f}};
            Up.own = RP.u00;
            emit_plane(oa, L.own, ov, ez, zslot, own_layer);
            DN_STAMP(stamp_E);
            return;
        }
#endif
        gather_stage(Up, F);
        DN_STAMP(stamp_B);
        v2f GX[2], GY[2], GU[2][2], Sz[2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                Sz[j][i] = L.V[j][i] + Up.V[j][i];
                const v2f q = Sz[j][i] * (Up.U[j][i] - L.U[j][i]);
                GU[j][i] = cU[j][i] - q;
                if constexpr (HAS_F) cU[j][i] = vfma(k.nbz, F[j][i], q);
                else cU[j][i] = q;
            }
        if constexpr (HAS_F) {
            // sum_g f_g u_g of the upper plane's share: counted where that plane is owned (the plane ez_end belongs to the next strip, the mesh's top
            // plane to the last one)
            const bool own_up_u = ez + 1 >= ez_own && (ez + 1 < ez_end || ez_end == p.nelz);
            const v2f s = vfma(F[1][1], Up.U[1][1], vfma(F[1][0], Up.U[1][0], vfma(F[0][1], Up.U[0][1], F[0][0] * Up.U[0][0])));
            const v2f cs = (cf3_usel_lt(0, own_up_u ? 1 : 0, 1.f, 0.f) * okown) * s;
            e2_acc += cs.x + cs.y;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const v2f T = (Sz[j][0] + Sz[j][1]) * (L.BX[j] + Up.BX[j]);
            GX[j] = cX[j] + T;
            cX[j] = vfma(k.kappa2, (Up.V[j][0] + Up.V[j][1]) * Up.BX[j], T);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const v2f T = (Sz[0][i] + Sz[1][i]) * (L.CY[i] + Up.CY[i]);
            GY[i] = cY[i] + T;
            cY[i] = vfma(k.kappa2, (Up.V[0][i] + Up.V[1][i]) * Up.CY[i], T);
        }
        v2f o[2][2];
        plane_adjoint(GX, GY, GU, o);
        emit_plane(o, L.own, ov, ez, zslot, own_layer);
        DN_STAMP(stamp_E);
    };

    // prologue: planes ez_begin (slot 0) and ez_begin + 1 (slot 1) into LDS (requested together), the lower one staged
    // DN_CF3_PF planes are in flight per thread: the plane published at the start of a layer was requested DN_CF3_PF layers earlier (with two, WA
    // carries the even and WB the odd planes after the first; measured equal to one, which is the default)
    RawNodes WA, WB;
    {
        RawNodes& W0 = WA;
        RawNodes& W = WB;
        if constexpr (HAS_F) {         // the forcing of the two planes before the first published one's upper neighbour
            const unsigned zm = (unsigned)max(ez_begin - 1, 0) * npl, z0 = (unsigned)ez_begin * npl;
            fh0 = ld_pair(sb.f, zm + own_off); fh1 = ld_pair(sb.f, z0 + own_off);
            hh0 = halo_src[zm + halo_off]; hh1 = halo_src[z0 + halo_off];
        }
        plane_request(ez_begin, W0);
        plane_request(ez_begin + 1, W);
        plane_publish(W0, ez_begin, 0);
        plane_publish(W, ez_begin + 1, 1);
        plane_request(ez_begin + 2, WA);
#if DN_CF3_PF == 2
        plane_request(ez_begin + 3, WB);
#endif
    }
    __syncthreads();
    {
        v2f F0[2][2];
        gather_issue(0);
        gather_stage(SA, F0);
        // the first plane's own terms (they matter for the mesh's bottom plane only: every other strip recomputes its first layer for the plane above)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            cX[j] = k.kappa * ((SA.V[j][0] + SA.V[j][1]) * SA.BX[j]);
            cY[j] = k.kappa * ((SA.V[0][j] + SA.V[1][j]) * SA.CY[j]);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if constexpr (HAS_F) cU[j][i] = k.nbz * F0[j][i];
                else cU[j][i] = zero2;
            }
        }
        if constexpr (HAS_F) {
            const v2f s = vfma(F0[1][1], SA.U[1][1], vfma(F0[1][0], SA.U[1][0], vfma(F0[0][1], SA.U[0][1], F0[0][0] * SA.U[0][0])));
            const v2f cs = (cf3_usel_lt(ez_begin, ez_own, 0.f, 1.f) * okown) * s;
            e2_acc += cs.x + cs.y;
        }
    }
    // every thread has read plane ez_begin (slot 0) before the first layer publishes plane ez_begin + 2 into that slot
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    int ez = ez_begin;
#if defined(DN_STAMP3D)
    stamp_last = __builtin_amdgcn_s_memtime();
#endif
#pragma nounroll
    for (; ez + 1 < ez_end; ez += 2) {
        layer(ez, 0, SA, SB, &WA);                // publishes plane ez + 2 (requested two layers ago), requests plane ez + 4
#if DN_CF3_PF == 2
        layer(ez + 1, 1, SB, SA, &WB);
#else
        layer(ez + 1, 1, SB, SA, &WA);
#endif
#if defined(DN_STAMP3D)
        stamp_n += 2;
#endif
    }
    // the last strip owns the top boundary plane: only the layer below contributes -- take back one of the plane's two products.  (Called on SA
    // or SB by name: selecting between the two states through a reference would put both into scratch memory.)
    auto top_plane = [&](const PlaneC& Tp, int zslot) {
        v2f GX[2], GY[2], o[2][2];
        OwnVals ov;
        own_read(zslot, ov);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            GX[j] = vfma(-k.kappa, (Tp.V[j][0] + Tp.V[j][1]) * Tp.BX[j], cX[j]);
            GY[j] = vfma(-k.kappa, (Tp.V[0][j] + Tp.V[1][j]) * Tp.CY[j], cY[j]);
        }
        plane_adjoint(GX, GY, cU, o);
        emit_plane(o, Tp.own, ov, p.nz - 1, zslot, true);
        flush_store();
    };
    if (ez < ez_end) {
        layer(ez, 0, SA, SB, nullptr);
        flush_store();
        if (ez_end == p.nelz) top_plane(SB, 1);
    } else {
        flush_store();
        if (ez_end == p.nelz) top_plane(SA, 0);
    }
#if defined(DN_STAMP3D)
    if (tid == 0) {                               // wave 0 of every workgroup (tools/stamp3d.py)
        const unsigned slot = blockIdx.x;
        if (slot < 8192u) {
            unsigned long long* d = dn_stamp_buf + slot * 8u;
            const unsigned long long hwid = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)), xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));
            d[0] = stamp_A | (stamp_A1 << 32); d[1] = stamp_B; d[2] = stamp_C; d[3] = stamp_D;       // A1 (gather issue + publish) in the upper half
            d[4] = stamp_E | ((__builtin_amdgcn_s_memrealtime() - stamp_rt0) << 40);
            d[5] = stamp_n | (hwid << 16) | ((xcc & 0xffull) << 48);
            d[6] = stamp_t0; d[7] = __builtin_amdgcn_s_memtime();
        }
    }
#endif
    if (p.want_sums) {
        // sum_a u_a out_a = wscale ( alpha sum nu |grad u|^2 - beta sum f u ): this thread's share of the stiffness sum (the identity holds for the total)
        const float e1_acc = (ut_acc * k.inv_esc + k.beta * e2_acc) * k.inv_alpha;
        finish_sums(p, e1_acc, e2_acc, sq_acc, tid, 256, red, &last_flag, (double)p.T.esc);
    }
}

// ---- dispatch ---------------------------------------------------------------------------------------------
template <int FLB>
static void launch_cf3_bc(const PoissonParams& pp, const Cf3Consts& kc, const dim3& grid, const Geom3D& g, hipStream_t s) {
    const bool any = pp.bc[0].mask || pp.bc[1].mask;
    const bool one = (pp.bc[0].mask != nullptr) != (pp.bc[1].mask != nullptr);
    bool f32 = false;
    for (int kk = 0; kk < 2; ++kk)
        if (pp.bc[kk].mask && !pp.bc[kk].mask_is_u8) f32 = true;
    const dim3 block(16, 16);
#define DN_CF3(FLAGS) hipLaunchKernelGGL((poisson3d_q1_cf_kernel<(FLAGS)>), grid, block, 0, s, pp, kc, g.chunks, g.tiles, g.strips)
    if (pp.bc[0].kind == DN_MASK_BOX || pp.bc[1].kind == DN_MASK_BOX) {       // box faces (no array), alone or beside ONE mask image
        if (!any) DN_CF3(FLB | CF3_BOX);
        else if (!f32) DN_CF3(FLB | CF3_BOX | CF3_IMG | CF3_ONE);
        else DN_CF3(FLB | CF3_BOX | CF3_IMG | CF3_F32 | CF3_ONE);
        return;
    }
    if (!any) DN_CF3(FLB);
    else if (!f32 && one) DN_CF3(FLB | CF3_IMG | CF3_ONE);
    else if (!f32) DN_CF3(FLB | CF3_IMG);
    else if (one) DN_CF3(FLB | CF3_IMG | CF3_F32 | CF3_ONE);
    else DN_CF3(FLB | CF3_IMG | CF3_F32);
#undef DN_CF3
}

// May this launch run the closed-form kernel?  (dn_poisson_apply has checked q1n2_ok(): exact 2-point rule, even nx, aligned pairs, nodal /
// absent / pre-assembled forcing, constant-value conditions as images of one kind or box faces.)  It needs a stiffness part (the scales of its
// records divide by alpha) and takes the energy from the nodal values.
bool poisson3d_q1_cf_ok(const PoissonParams& pp) {
    if (config(CFG_Q1_3D_N2) != nullptr || config(CFG_Q1_3D_E1SUM) != nullptr) return false;
    if (!(pp.T.alpha != 0.f) || !std::isfinite(pp.T.alpha)) return false;
    if (!(pp.T.kap[0] != 0.f && pp.T.kap[1] != 0.f && pp.T.kap[2] != 0.f)) return false;
    if (std::fabs(pp.T.b[0][1] + pp.T.b[1][1] - 1.0f) > 1e-6f) return false;        // the closed form assumes the symmetric rule
    return true;
}

int launch_poisson3d_q1_cf(const PoissonParams& pp, const Geom3D& g, int batch, hipStream_t s) {
    if (!(g.E == 2 && g.TX == 16 && g.TY == 16)) return DN_E_UNSUPPORTED;
    // constants of the closed form, in double from the rule's own table entries (unit weights)
    const double t0 = pp.T.b[0][1], t1 = pp.T.b[1][1];
    const double sig = 0.5 * (t0 * t0 * t0 + t1 * t1 * t1 + (1 - t0) * (1 - t0) * (1 - t0) + (1 - t1) * (1 - t1) * (1 - t1));
    const double tau = 0.5 * (t0 * t0 * (1 - t0) + t1 * t1 * (1 - t1) + (1 - t0) * (1 - t0) * t0 + (1 - t1) * (1 - t1) * t1);
    const double kx = pp.T.kap[0], ky = pp.T.kap[1], kz = pp.T.kap[2];
    Cf3Consts kc;
    kc.t0 = (float)t0; kc.t1 = (float)t1;
    kc.snu = (float)(kx * tau);
    const double ry = ky / kx, rz = kz / (kx * tau);
    kc.ry = (float)ry; kc.rz = (float)rz;
    kc.rzt0 = (float)(rz * t0); kc.rzt1 = (float)(rz * t1); kc.ryt0 = (float)(ry * t0); kc.ryt1 = (float)(ry * t1);
    kc.kappa = (float)(sig / tau - 1.0); kc.kappa2 = (float)(2.0 * (sig / tau - 1.0));
    kc.nbw = pp.T.nbw;
    kc.nbz = (float)((double)pp.T.nbw / rz);
    kc.c00 = pp.T.q1my[0]; kc.c01 = pp.T.q1my[1]; kc.c11 = pp.T.q1my[2];
    // f'(m) = c01 (f(m-1) + f(m+1)) + czd f(m); on the mesh's bottom / top plane the clamped load of the missing neighbour returns f(m) itself: its c01
    // and the missing element's diagonal share are taken off the middle coefficient
    kc.czd[0] = kc.c00 + kc.c11; kc.czd[1] = kc.c00 - kc.c01; kc.czd[2] = kc.c11 - kc.c01; kc.czd[3] = -2.f * kc.c01;
    kc.inv_esc = (float)(1.0 / (double)pp.T.esc); kc.beta = pp.T.beta; kc.inv_alpha = (float)(1.0 / (double)pp.T.alpha);
    const dim3 grid((unsigned)((long long)g.chunks * g.tiles * g.strips * batch));
    const int sel = (pp.nu ? 1 : 0) | (pp.f ? 2 : 0);
    if (pp.f && pp.f_is_load) {                    // the forcing as the assembled load vector (dn_poisson_args.f_is_load)
        if (sel == 2) launch_cf3_bc<CF3_F | CF3_LOAD>(pp, kc, grid, g, s);
        else launch_cf3_bc<CF3_NU | CF3_F | CF3_LOAD>(pp, kc, grid, g, s);
        return 0;
    }
    switch (sel) {
        case 0: launch_cf3_bc<0>(pp, kc, grid, g, s); break;
        case 1: launch_cf3_bc<CF3_NU>(pp, kc, grid, g, s); break;
        case 2: launch_cf3_bc<CF3_F>(pp, kc, grid, g, s); break;
        default: launch_cf3_bc<CF3_NU | CF3_F>(pp, kc, grid, g, s); break;
    }
    return 0;
}

}  // namespace dn
