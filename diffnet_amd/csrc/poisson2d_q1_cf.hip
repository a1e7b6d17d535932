// 2-D Q1 fused Poisson kernel, closed-form element (the default for nodal / absent forcing).
//
// With u, nu, f bilinear on the element,
//     u(a, c) = U0 + a UX + c UY + a c UXY,   a, c in [0, 1] the lerp coordinates of the 1-D Gauss points,
// every quadrature sum of the reference's loss bodies is a polynomial in (a_i, c_j) summed against the rule, i.e. a
// combination of the rule's moments  M_r = sum_g w_g b_g^r  (r = 0..3; the x moments carry the Jacobian / user scale):
//     sum_ij W_ij nu_ij (u_x)^2_j = hs0^2 (UX^2 A0 + 2 UX UXY A1 + UXY^2 A2),  A_k = P My_k + Q My_{k+1},
//                                   P = Mx0 N0 + Mx1 NX,  Q = Mx0 NY + Mx1 NXY          (and likewise in y with B_k),
//     sum_ij W_ij f_ij u_ij       = L0 U0 + LX UX + LY UY + LXY UXY                     (L.: moment combinations of f).
// This is algebra, not a change of rule: the moments are computed from the rule's own points and weights (including the
// reference's truncated 3- and 4-point literals), so the value equals the Gauss sum over any ngp x ngp points to rounding,
// while the element costs ~75 VALU instructions whatever ngp is (115 for the per-point form at 3 x 3).
//
// Mapping as the other 2-D kernels: a thread owns E consecutive elements of a strip and marches over element rows; it
// carries the lower node row (raw nodal values) and the layer-below contributions to that row's nodes; two rows per loop
// trip with the two raw rows / carry sets swapping roles (no copies); node shared with the right neighbour through LDS.
#include <cstdlib>
#include <type_traits>

#define DN_NT_STORES 1            // the gradient rows are written once and never re-read by the launch: 54.3 -> 52.8 us (profiles/r2_ab2d_nt.txt)
#include "poisson_common.h"

namespace dn {

// CF_BC_PACKED: bit-packed masks / box faces with constant values; CF_PK_NB1 / CF_PK_NB2: one / two of the conditions are bit arrays
// (compile-time: a load inside a wave-uniform branch costs an s_waitcnt vmcnt(0) where the branch joins)
enum : int { CF_NU = 1, CF_F = 2, CF_BC = 8, CF_BC_U8C = 16, CF_BC_PACKED = 32, CF_PK_NB1 = 64, CF_PK_NB2 = 128, CF_UA = 256 };
// CF_UA (E = 4, vector accesses, one strip per workgroup): rows of 4 k + 1 nodes -- the 2^n + 1 meshes.  The 16-byte row accesses are then aligned to 4 bytes
// only (the hardware takes that), and the mesh's LAST node column belongs to the last full thread column (its node x0 + 4) instead of to a thread column of
// its own: the launch has the geometry of the 4 k mesh (513^2 x 64: 112.9 -> the time of 512^2, tools/time_2d_sizes.py).

template <int E>
struct CfRow {
    float u[E + 1], n[E + 1], f[E + 1];
    float g[E + 1];          // x-stage of the forcing term: (1-D element mass matrix in x) applied to f over the thread's own elements
    float keep[E];
    float keepx;             // CF_UA: keep of the node x0 + E (the mesh's last node column in the last thread column)
    BcRaw<E> bc;
    uint32_t m8[2][2];
    uint32_t mb[2];          // CF_BC_PACKED: bits of the row's nodes x0, x0 + 1, ... from the bit arrays (bit n = node x0 + n)
    uint32_t bx[2];          //                 box-face bits of those nodes, per condition
};

#ifndef DN_Q1_2D_WAVES
#define DN_Q1_2D_WAVES 2
#endif
#ifndef DN_PRIO_ROT
#define DN_PRIO_ROT 3
#endif

#ifndef DN_CF_DPPX
#define DN_CF_DPPX 0              // 1: E = 4 takes the shared node x0 + 4 over DPP from the neighbouring lane instead of a strided per-lane load; measured equal (profiles/r2_2d_ab.txt)
#endif                            // 2: the same over ds_bpermute (load_seg4_shfl)
#ifndef DN_CF_NT_LD
#define DN_CF_NT_LD 0             // with DN_CF_DPPX == 2: 1 = non-temporal row loads of nu and f, 2 = of u as well
#endif
#ifndef DN_CF_FMASS
#define DN_CF_FMASS 1
#endif
#ifndef DN_CF_NT_COEF
#define DN_CF_NT_COEF 1           // 1: non-temporal vector loads for nu and f (read once per launch; the node shared with the right neighbour stays a plain
                                  // load).  Round 2 measured it on ONE re-evaluated batch, where it gives up the Infinity-Cache hits (47 -> 53 us) and left
                                  // it off; on batches in rotation -- every launch streams from HBM, the regime bench.py now times -- it wins: mask bits
                                  // 58.3 -> 55.5 us, box 56.5 -> 56.0 (profiles/r3_rotate_nt.txt).  2: u as well (slower: 59.8)
#endif
#ifndef DN_CF_REV
#define DN_CF_REV 0               // 1: neighbouring strips march in opposite directions and share an XCD (see the kernel): the rows two strips share
#endif                            // are fetched once (measured traffic 1.09x -> 1.002x of the algorithmic bytes).  Faster when one batch is re-evaluated and
                                  // its arrays partly live in the Infinity Cache (45.4 vs 47.5 us), SLOWER when every launch streams from HBM (different
                                  // batches in rotation: 59.5 vs 56.4 us, profiles/r2_rotate_variants.txt) -- off.  0: every strip upwards, dispatch order
#ifndef DN_CF_PF
#define DN_CF_PF 0                // 1 / 2: software-pipelined rows (one / two raw rows in flight while a layer is computed).  Steady state, box condition
                                  // (profiles/r2_plan2d_steady.txt): default plan 46.9 us, PF=1 48.1, PF=2 50.6; strips of 32 rows 54.5 / 48.9 / 46.3.  Off: 79 VGPRs
#endif

#ifndef DN_CF_PK
#define DN_CF_PK 0                // 1: the element arithmetic of a layer runs on element PAIRS in packed fp32 registers (v_pk_fma_f32 ...; E even): 722 -> 536 VALU
                                  // instructions per two rows, but 105 -> 154 VGPRs, i.e. 3 instead of 4 waves per SIMD: the 4096 waves of the bench launch no longer fit one
                                  // round (69 us); with 22-row strips (one round at 3 waves) 55.2 us against 57.4 for the scalar form on that plan and 55.5 for the scalar
                                  // form on the default plan -- no gain, off (profiles/r3_2d_packed.txt)
#endif
#ifndef DN_CF_W
#define DN_CF_W 4                 // sub-strips CHAINED per workgroup where the launch plan asks for it ("PLAN2D" "T,E,R,W"; not the default, see plan2d in
#endif                            // poisson_fused.hip): W x 2 waves march W neighbouring strips; a strip takes the row it shares with the strip below / above
                                  // through LDS instead of re-reading it from HBM and recomputing the seam layer: a workgroup of W strips of R rows reads
                                  // W R + 2 rows instead of W (R + 2)
constexpr int CF_TS = 128;        // threads per chained sub-strip (two waves: 128 x 4 elements = one 512-node row segment)
constexpr int CF_NSLOT = 64;      // hand-over slots between the two waves of a sub-strip: one per emitted row, never reused (R + 3 <= 64)
constexpr unsigned CF_SPIN_MAX = 1u << 20;   // bound of every LDS flag poll: a producer wave that never arrives must not hang the GPU.  Reaching the bound is NOT
                                             // silent: the wave poisons everything it writes afterwards with NaN (output rows, its partial sums -> the
                                             // launch's energy / sumsq are NaN) and sets the sticky error word of the workspace header (dn_workspace_status)

// LDS hand-overs of the chained strips are spelled as inline asm: a `volatile` LDS access makes the compiler wait for EVERY outstanding
// memory operation around it (s_waitcnt vmcnt(0) lgkmcnt(0): 237 of them in the first build of the chained kernel, i.e. the row loads
// no longer overlapped anything, +5 us per launch).  The asm forms touch lgkmcnt only; LDS returns a wave's accesses in order.
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p; }
__device__ __forceinline__ void lds_st(unsigned a, float v) { asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_st(unsigned a, unsigned v) { asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_ld_nowait(float& d, unsigned a) { asm volatile("ds_read_b32 %0, %1" : "=v"(d) : "v"(a) : "memory"); }
__device__ __forceinline__ unsigned lds_ld_u(unsigned a) {
    unsigned d;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(d) : "v"(a) : "memory");
    return d;
}
__device__ __forceinline__ float lds_ld(unsigned a) {
    float d;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(d) : "v"(a) : "memory");
    return d;
}
// after a group of lds_ld_nowait: the wait, tied to the destinations so that no use of them is scheduled above it
template <int N>
__device__ __forceinline__ void lds_wait(float (&v)[N]) {
    if constexpr (N == 5) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4])::"memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3])::"memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2])::"memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1])::"memory");
    else { static_assert(N == 1, "lds_wait: 1..5 values"); asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0])::"memory"); }
}

#ifdef DN_STAMP2D
// Diagnostic build only (tools/clock2d.py): shader-clock ticks (s_memtime) and constant-100-MHz ticks (s_memrealtime) of every workgroup's
// lifetime -> the clock the kernel really ran at (the MI355X lowers it under load: profiles/r2_clock_under_load.txt)
__device__ unsigned long long dn_stamp2d_buf[8192 * 4];      // per workgroup: shader ticks of its march, then constant-clock (10 ns) stamps: start, end of the march, end of the kernel
extern "C" int dn_debug_stamps2d(void* dst, size_t bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(dn_stamp2d_buf), bytes); }
#endif

template <int E, bool VEC, int FL, int W>
__global__ void __launch_bounds__(W > 1 ? CF_TS * W : 256, W > 1 ? 4 : DN_Q1_2D_WAVES) poisson2d_q1_cf_kernel(const PoissonParams p) {
#ifdef DN_STAMP2D
    const unsigned long long stamp_t0 = __builtin_amdgcn_s_memtime(), stamp_rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    constexpr int NW = E;
    constexpr bool HAS_NU = (FL & CF_NU) != 0, HAS_F = (FL & CF_F) != 0;
    constexpr bool BC_ANY = (FL & (CF_BC | CF_BC_U8C | CF_BC_PACKED)) != 0, BC_U8C = (FL & CF_BC_U8C) != 0, BC_PACKED = (FL & CF_BC_PACKED) != 0;
    constexpr bool UA = (FL & CF_UA) != 0;
    static_assert(!UA || (E == 4 && VEC && W == 1 && !DN_CF_REV && !DN_CF_PF && DN_CF_DPPX == 0), "CF_UA: four elements per thread, vector accesses, plain upward march");
    static_assert(W == 1 || (!DN_CF_REV && !DN_CF_PF), "chained sub-strips: plain upward march only");
    // W > 1: the workgroup holds W sub-strips of CF_TS threads each; sub (wave-uniform) is this thread's sub-strip
    const int T = W > 1 ? CF_TS : (int)blockDim.x;
    const int sub = W > 1 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x / CF_TS) : 0;       // wave-uniform: everything derived from it stays scalar
    const int tid = W > 1 ? (int)threadIdx.x - sub * CF_TS : (int)threadIdx.x;
#if DN_CF_REV
    // Workgroup -> (chunk, strip, sample) so that NEIGHBOURING strips run on the same XCD (one L2) at the same time: the dispatcher
    // hands consecutive workgroups (x fastest, then y, z) to consecutive XCDs; XCD k takes the k-th contiguous range of strips.
    unsigned lid = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
#if defined(DN_CF_REV_PAIRS)          // (round 4 experiment) only the two strips of a PAIR share an XCD, dispatched 8 workgroups apart; consecutive pairs go round the
    {                                 // XCDs in dispatch order as in the default launch (no contiguous range of samples per XCD)
        const unsigned nwg = gridDim.x * gridDim.y * gridDim.z;
        if ((nwg & 15u) == 0u) {
            const unsigned xcd = lid & 7u, member = (lid >> 3) & 1u, q = lid >> 4;
#if DN_CF_REV_PAIRS == 2              // pairs (2 k + 1, 2 k + 2): the strips that read their shared rows at their START (all workgroups start together: the second read
            lid = (2u * (8u * q + xcd) + member + nwg - 1u) % nwg;     // meets the first in the L2; at the ends the two marches are microseconds apart)
#else
            lid = 2u * (8u * q + xcd) + member;
#endif
        }
    }
#elif !defined(DN_CF_REV_NOXCD)       // (round 4 experiment NOXCD: opposite marches in plain dispatch order -- the second read of a shared row then meets the first in the Infinity Cache, not in an L2)
    {
        const unsigned nwg = gridDim.x * gridDim.y * gridDim.z, xcd = lid & 7u, idx = lid >> 3, base = nwg >> 3, rem = nwg & 7u;
        lid = xcd * base + min(xcd, rem) + idx;
    }
#endif
    const int chunk = (int)(lid % gridDim.x), strip = (int)((lid / gridDim.x) % gridDim.y), b = (int)(lid / (gridDim.x * gridDim.y));
#else
    const int chunk = blockIdx.x, strip = W > 1 ? (int)blockIdx.y * W + sub : selected_strip(p, (int)blockIdx.y), b = blockIdx.z;      // (chained launches cover every strip)
#endif
    const bool active = W == 1 || strip < p.nstrips;                         // the last workgroup of a sample may hold fewer than W strips
    const bool chain_dn = W > 1 && sub > 0;                                  // the strip below is in this workgroup
    const bool chain_up = W > 1 && sub + 1 < W && strip + 1 < p.nstrips;     // the strip above is
    const int q = chunk * (T - 1) + tid;  // logical thread column (chunks overlap by one thread)
    const int ex0 = q * E;
    const int x0 = ex0;
    const bool col_owner = !(chunk > 0 && tid == 0);
    const bool lastcol = UA && x0 + NW == p.nx - 1;      // this thread's shared node is the mesh's last node column: nobody to the right owns it
    const int64_t nps = (int64_t)p.nx * p.ny;
    const SampleBases sb = sample_bases(p, b, nps);
    const int R = p.rows_per_strip;
    // The strip owns the node rows [sb0, sc0) (the last strip also the top row of the domain) and the element layers between them; it
    // also computes the layer below its first row (the seam: that row's other half), so it reads the node rows sb0 - 1 .. sc0.
    // DN_CF_REV: odd strips march DOWNWARDS, in the mirrored row coordinate y' = nely - y (the element is mirror-symmetric: the rule's
    // moments of b and of 1 - b are the same numbers, so the same formulas apply with lower / upper rows exchanged).  An even strip and
    // the odd one above it then read the two rows they share at the same time (both at their end), the odd strip and the even one above
    // it both at their start: with the strips of a pair on one XCD the second read is an L2 hit instead of HBM traffic.
    const int sb0 = strip * R, sc0 = min(sb0 + R, p.nely);
    const bool rev = DN_CF_REV && (strip & 1);
    const int ysgn = rev ? -1 : 1, yoff = rev ? p.nely : 0;                  // physical row of the logical row y: ysgn * y + yoff
    // logical layers [ey_begin, ey_end); energy counted for layers e_from <= ey <= e_until, rows stored for ey >= r_from
    const int ey_begin = rev ? p.nely - sc0 : ((sb0 > 0 && !chain_dn) ? sb0 - 1 : sb0);     // chain_dn: no seam layer, the carry comes from the strip below
    const int ey_end = rev ? (sb0 > 0 ? p.nely - sb0 + 1 : p.nely) : sc0;
    const int r_from = rev ? (sc0 == p.nely ? ey_begin : ey_begin + 1) : sb0;
    const int e_from = rev ? 0 : sb0, e_until = rev ? p.nely - 1 - sb0 : p.nely;
    const bool top_row = rev ? sb0 == 0 : sc0 == p.nely;                     // the strip also finishes the last logical row (no layer above it)

    __shared__ float xch[2][256];
    __shared__ double red[W > 1 ? 4 * W : 8];
    __shared__ int last_flag;
    // W > 1 -- hand-overs inside the workgroup, all through LDS words guarded by flags (no workgroup barrier: the sub-strips, and the two
    // waves of a sub-strip, run at their own pace; a producer never waits for a consumer, so there is no cycle of waits):
    //   cf_edge / cf_xflag   per emitted row, wave 0 of a sub-strip -> wave 1: the contribution to the node the two waves share
    //   cf_seam / cf_sflag   once per strip: the strip above publishes its first node row (Dirichlet applied, forcing x-staged) for the
    //                        last layer of the strip below [0 .. 3 NW + 2]; the strip below answers with that layer's contributions
    //                        to this row [3 NW + 3 .. 4 NW + 3]; the strip above parks its own half of the row meanwhile [4 NW + 4 ..]
    constexpr int CW = W > 1 ? W : 2, SEAM_WORDS = 6 * (E + 1);
    __shared__ float cf_edge[W > 1 ? W : 1][W > 1 ? CF_NSLOT : 1];
    __shared__ unsigned cf_xflag[W > 1 ? W : 1];
    __shared__ float cf_seam[CW - 1][W > 1 ? SEAM_WORDS : 1][W > 1 ? CF_TS : 1];
    __shared__ unsigned cf_sflag[CW - 1][2][2];          // [seam][0: row published, 1: carry published][wave of the sub-strip]
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned seq = 0u;                                   // rows handed over so far (same count in both waves of a sub-strip)
    if constexpr (W > 1) {
        static_assert(CF_TS == 128, "two waves per sub-strip");
        if (threadIdx.x < (unsigned)W) cf_xflag[threadIdx.x] = 0u;
        if (threadIdx.x < 4u * (W - 1)) (&cf_sflag[0][0][0])[threadIdx.x] = 0u;
        __syncthreads();
    }
    // dn_poisson_args.fold_prev: the launch's first workgroup closes the evaluation before it (1-2 us once per launch, off every critical path)
    if ((blockIdx.x | blockIdx.y | blockIdx.z) == 0u) fold_prev_sums(p, (int)threadIdx.x, (int)blockDim.x, red);
    float spin_poison = 0.f;                                              // NaN once a poll has run into its bound (wave-uniform)
    auto spin_until = [&](const unsigned* flag, unsigned want) {          // wave-uniform poll of an LDS word
        const unsigned fa = lds_addr(flag);
        const unsigned lim = p.spin_limit > 0 ? (unsigned)p.spin_limit : CF_SPIN_MAX;
        unsigned n = 0;
        for (; n < lim; ++n) {
            if ((unsigned)__builtin_amdgcn_readfirstlane((int)lds_ld_u(fa)) >= want) break;
            __builtin_amdgcn_s_sleep(1);
        }
        if (n == lim) spin_poison = __builtin_nanf("");
    };

    const bool has_mask[2] = {sb.mask[0] != nullptr, sb.mask[1] != nullptr};
    const uint8_t* mask8[2];
    mask8[0] = reinterpret_cast<const uint8_t*>(has_mask[0] ? sb.mask[0] : sb.mask[1]);
    mask8[1] = reinterpret_cast<const uint8_t*>(has_mask[1] ? sb.mask[1] : sb.mask[0]);

    // CF_BC_PACKED: per condition either one bit per node in HBM (one 2-byte load per thread, row and bit array) or the domain faces
    // (no load).  Branch-free: NB loads are always issued; what a condition contributes is selected with wave-uniform masks.
    constexpr unsigned NBITS = (1u << (NW + 1)) - 1u;
    constexpr int NB = !BC_PACKED ? 0 : ((FL & CF_PK_NB2) ? 2 : ((FL & CF_PK_NB1) ? 1 : 0));
    const bool isbits[2] = {p.bc[0].kind == DN_MASK_BITS, p.bc[1].kind == DN_MASK_BITS};
    const unsigned bsel[2] = {isbits[0] ? NBITS : 0u, isbits[1] ? NBITS : 0u};
    // load slot j reads the bit array of condition src[j]; with one array it is whichever condition has it
    const int src0 = (NB == 2 || isbits[0]) ? 0 : 1;
    const uint32_t* bptr[2] = {reinterpret_cast<const uint32_t*>(sb.mask[src0]), reinterpret_cast<const uint32_t*>(sb.mask[1])};
    // two ALIGNED dwords per bit array and row (the word that holds node x0 and the next one; 8 lanes share an address) and one
    // v_alignbit: an unaligned 2-byte window costs the address path 16 cycles per wave-instruction, an aligned dword 4.5
    const unsigned brow[2] = {(unsigned)p.bc[src0].row_bytes / 4u, (unsigned)p.bc[1].row_bytes / 4u};       // words per node row
    unsigned bw0[2], bw1[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int rw = max((int)brow[j], 1);
        bw0[j] = (unsigned)min(x0 >> 5, rw - 1);
        bw1[j] = (unsigned)min((x0 >> 5) + 1, rw - 1);
    }
    const unsigned bsh = (unsigned)x0 & 31u;
    unsigned boxx[2] = {0u, 0u};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int faces = p.bc[k].kind == DN_MASK_BOX ? p.bc[k].box_faces : 0;
#pragma unroll
        for (int n = 0; n <= NW; ++n) {
            const bool on = ((faces & DN_FACE_XLO) && x0 + n == 0) || ((faces & DN_FACE_XHI) && x0 + n == p.nx - 1);
            boxx[k] |= on ? (1u << n) : 0u;
        }
    }
    const int ylo[2] = {(p.bc[0].kind == DN_MASK_BOX && (p.bc[0].box_faces & DN_FACE_YLO)) ? 0 : -1,
                        (p.bc[1].kind == DN_MASK_BOX && (p.bc[1].box_faces & DN_FACE_YLO)) ? 0 : -1};
    const int yhi[2] = {(p.bc[0].kind == DN_MASK_BOX && (p.bc[0].box_faces & DN_FACE_YHI)) ? p.ny - 1 : -1,
                        (p.bc[1].kind == DN_MASK_BOX && (p.bc[1].box_faces & DN_FACE_YHI)) ? p.ny - 1 : -1};
    auto packed_issue = [&](int yc, CfRow<E>& r) {                   // yc: node row, already clamped into the mesh
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const unsigned ro = (unsigned)yc * brow[j];
            const uint32_t w0 = ld_at<uint32_t>(bptr[j], ro + bw0[j]), w1 = ld_at<uint32_t>(bptr[j], ro + bw1[j]);
            r.mb[j] = __builtin_amdgcn_alignbit(w1, w0, bsh);          // bit n = node x0 + n
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) r.bx[k] = (yc == ylo[k] || yc == yhi[k]) ? NBITS : boxx[k];
    };

    auto lseg = [&](auto base, unsigned rowoff, auto& dst, auto nt) {
        if constexpr (E == 4 && VEC && DN_CF_DPPX == 2) load_seg4_shfl<decltype(nt)::value>(base, rowoff, x0, p.nx, dst);
        else if constexpr (E == 4 && VEC && DN_CF_DPPX) load_seg4_dpp(base, rowoff, x0, p.nx, dst);
        else load_seg<NW, VEC>(base, rowoff, x0, p.nx, dst);
    };
    constexpr std::integral_constant<bool, (DN_CF_NT_LD >= 2)> NT_U{};
    constexpr std::integral_constant<bool, (DN_CF_NT_LD >= 1)> NT_C{};
    constexpr std::false_type NT_NO{};
    auto row_issue = [&](int yr, CfRow<E>& r) {
        const int yp = ysgn * min(yr, p.ny - 1) + yoff;
        const unsigned rowoff = (unsigned)yp * (unsigned)p.nx;
#if DN_CF_NT_COEF >= 2
        load_seg_stream<NW, VEC>(sb.u, rowoff, x0, p.nx, r.u);
#else
        lseg(sb.u, rowoff, r.u, NT_U);
#endif
#if DN_CF_NT_COEF
        if constexpr (HAS_NU) load_seg_stream<NW, VEC>(sb.nu, rowoff, x0, p.nx, r.n);
        if constexpr (HAS_F) load_seg_stream<NW, VEC>(sb.f, rowoff, x0, p.nx, r.f);
#else
        if constexpr (HAS_NU) lseg(sb.nu, rowoff, r.n, NT_C);
        if constexpr (HAS_F) lseg(sb.f, rowoff, r.f, NT_C);
#endif
        if constexpr (BC_PACKED) {
            packed_issue(yp, r);
        } else if constexpr (BC_U8C) {
            // both mask slots are loaded unconditionally (an absent one re-reads the other and is ignored): a load inside a
            // wave-uniform branch makes the compiler wait vmcnt(0) where the branch joins, which would drain the pipelined rows
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (DN_CF_PF || has_mask[k]) {        // not pipelined: an absent condition costs no load (wave-uniform branch)
                    uint8_t t[NW + 1];
#if DN_NT_MASK
                    load_seg_stream<NW, VEC>(mask8[k], rowoff, x0, p.nx, t);
#else
                    lseg(mask8[k], rowoff, t, NT_NO);
#endif
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
    auto row_bc = [&](CfRow<E>& r) {
#pragma unroll
        for (int n = 0; n < NW; ++n) r.keep[n] = 1.f;
        if constexpr (UA) r.keepx = 1.f;
        if constexpr (BC_PACKED) {
            unsigned b0 = r.bx[0], b1 = r.bx[1];
            if constexpr (NB == 2) { b0 |= r.mb[0] & bsel[0]; b1 |= r.mb[1] & bsel[1]; }
            if constexpr (NB == 1) { b0 |= r.mb[0] & bsel[0]; b1 |= r.mb[0] & bsel[1]; }
            const float v0 = p.bc[0].value, v1 = p.bc[1].value;
#pragma unroll
            for (int n = 0; n <= NW; ++n) {
                r.u[n] = (b0 & (1u << n)) ? v0 : r.u[n];
                r.u[n] = (b1 & (1u << n)) ? v1 : r.u[n];
                if (n < NW) r.keep[n] = ((b0 | b1) & (1u << n)) ? 0.f : 1.f;
            }
            if constexpr (UA) r.keepx = ((b0 | b1) & (1u << NW)) ? 0.f : 1.f;
        } else if constexpr (BC_U8C) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (DN_CF_PF || has_mask[k]) {
                    const float val = p.bc[k].value;
#pragma unroll
                    for (int n = 0; n <= NW; ++n) {
                        const bool set = has_mask[k] && (n < NW ? ((r.m8[k][0] >> (8 * n)) & 0xffu) != 0u : r.m8[k][1] != 0u);
                        r.u[n] = set ? val : r.u[n];
                        if (n < NW) r.keep[n] = set ? 0.f : r.keep[n];
                        else if constexpr (UA) r.keepx = set ? 0.f : r.keepx;
                    }
                }
            }
        } else if constexpr (BC_ANY) {
            if constexpr (UA) {
                float k5[NW + 1];
                bc_apply_all<NW>(p, sb, r.bc, r.u, k5);
#pragma unroll
                for (int n = 0; n < NW; ++n) r.keep[n] = k5[n];
                r.keepx = k5[NW];
            } else {
                bc_apply<NW>(p, sb, r.bc, r.u, r.keep);
            }
        }
    };

    float e1_acc = 0.f, e2_acc = 0.f, sq_acc = 0.f;
    int par = 0;

    // The finished row is stored by flush_store(), after the next row has been consumed and the one after it requested: a store
    // issued here would be younger than the loads the next consumer waits for, and because it sits in a divergent branch the
    // compiler turns that wait into vmcnt(0), i.e. into a wait for the store itself.
    float pend_v[NW];
    unsigned pend_row = 0u;
    bool pend_st = false;
    float pend_x = 0.f;          // CF_UA: the last node column's value (last thread column)
    auto flush_store = [&]() {
        if (pend_st) store_seg<NW, VEC>(sb.out, pend_row, x0, p.nx, pend_v);
        if constexpr (UA) {
            if (pend_st && lastcol) st_at<float>(sb.out, pend_row + (unsigned)(p.nx - 1), pend_x);
        }
        pend_st = false;
    };
    auto emit_row = [&](const float (&o)[NW + 1], const float (&keep)[NW], int yr, bool owned_row, float keepx = 1.f) {
        float left;
        if constexpr (W > 1) {
            // inside a wave: lane l takes o[NW] of lane l - 1 (ds_bpermute); across the two waves: one LDS word per row + a row counter
            const float up = __shfl_up(o[NW], 1, 64);
            const unsigned ea = lds_addr(&cf_edge[sub][seq & (CF_NSLOT - 1)]);
            float edge = 0.f;
            if (wv == 0) {
                if (lane == 63) { lds_st(ea, o[NW]); lds_st(lds_addr(&cf_xflag[sub]), seq + 1u); }     // LDS executes a wave's accesses in order: value before counter
            } else {
                spin_until(&cf_xflag[sub], seq + 1u);
                edge = lds_ld(ea);
            }
            left = lane > 0 ? up : edge;
            ++seq;
        } else {
            xch[par][tid] = o[NW];
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // LDS-only barrier (loads stay in flight)
            left = (tid > 0) ? xch[par][tid - 1] : 0.f;
            par ^= 1;
        }
        // (CF_UA: the thread column right of the last full one receives the last node column's contribution over the hand-over, but does not own it)
        const bool st = owned_row && col_owner && (!UA || x0 + NW <= p.nx - 1);
#pragma unroll
        for (int n = 0; n < NW; ++n) {
            const float t = (o[n] + (n == 0 ? left : 0.f)) * keep[n];
            sq_acc = st ? fmaf(t, t, sq_acc) : sq_acc;       // nodes beyond the domain receive no contribution: t == 0
            pend_v[n] = t * p.out_scale;
            if constexpr (W > 1) pend_v[n] += spin_poison;
        }
        if constexpr (UA) {               // the mesh's last node column: finished by the thread column left of it (no thread to hand it to)
            const float t = o[NW] * keepx;
            sq_acc = (st && lastcol) ? fmaf(t, t, sq_acc) : sq_acc;
            pend_x = t * p.out_scale;
        }
        pend_row = (unsigned)(ysgn * yr + yoff) * (unsigned)p.nx;
        pend_st = st && sb.out != nullptr;
        if (!DN_CF_PF) flush_store();
    };

    const float mx0 = p.T.mxs[0], mx1 = p.T.mxs[1], mx2 = p.T.mxs[2], mx3 = p.T.mxs[3];
    const float my0 = p.T.m[0], my1 = p.T.m[1], my2 = p.T.m[2], my3 = p.T.m[3];
    const float k0 = p.T.q1c[0], k1 = p.T.q1c[1], h0 = p.T.q1c[2], h1 = p.T.q1c[3], nb = -p.T.beta;

    // forcing term as a tensor-product mass matrix applied row by row (14 instead of 28 instructions per element: -11 % VALU instructions,
    // time unchanged -- the kernel moves its ~312 MB at 5.45 TB/s whatever the arithmetic costs, profiles/r2_ab2d_fmass.txt); not in the
    // generic mask / value-field form, whose register budget it would push from 4 to 3 waves per SIMD
    constexpr bool FMASS = HAS_F && DN_CF_FMASS && (FL & CF_BC) == 0;
    const float cx00 = p.T.q1mx[0], cx01 = p.T.q1mx[1], cx11 = p.T.q1mx[2], cy00 = p.T.q1my[0], cy01 = p.T.q1my[1], cy11 = p.T.q1my[2];
    // sum_g W_g f_g N_a(g) with f bilinear is (mass_x (x) mass_y) f.  x-stage, once per node row: g[n] = the row's forcing seen through
    // the thread's own elements (elements beyond the mesh excluded; the node shared with the right neighbour gets the rest over the
    // hand-over that o[] takes anyway, by linearity)
    auto fstage = [&](CfRow<E>& r) {
        if constexpr (FMASS) {
#pragma unroll
            for (int n = 0; n <= NW; ++n) r.g[n] = 0.f;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                if (ex0 + e < p.nelx) {
                    r.g[e] = fmaf(cx01, r.f[e + 1], fmaf(cx00, r.f[e], r.g[e]));
                    r.g[e + 1] = fmaf(cx11, r.f[e + 1], cx01 * r.f[e]);
                }
            }
        }
    };

    // one element layer between the lower row L (Dirichlet applied) and the freshly landed upper row U; cin holds the
    // contributions of the layer below to L's nodes, cout receives this layer's contributions to U's nodes
    // fresh: U has just landed from HBM (Dirichlet select and forcing x-stage still to do); not fresh: U came finished from the strip above
    auto layer = [&](auto fresh, int ey, const CfRow<E>& L, CfRow<E>& U, const float (&cin)[NW + 1], float (&cout)[NW + 1]) {
        const bool own_layer = ey >= e_from && ey <= e_until;
        const float cnt = (own_layer && col_owner) ? 1.f : 0.f;
        if (!DN_CF_PF && decltype(fresh)::value) { row_bc(U); fstage(U); }
        float o[NW + 1], le1 = 0.f, le2 = 0.f;
#pragma unroll
        for (int n = 0; n <= NW; ++n) { o[n] = cin[n]; cout[n] = 0.f; }
        if constexpr (FMASS) {            // y-stage of the forcing term for the whole row segment, and its energy  u . (M f)
#pragma unroll
            for (int n = 0; n <= NW; ++n) {
                const float tlo = fmaf(cy01, U.g[n], cy00 * L.g[n]), tup = fmaf(cy11, U.g[n], cy01 * L.g[n]);
                o[n] = fmaf(nb, tlo, o[n]);
                cout[n] = nb * tup;
                le2 = fmaf(U.u[n], tup, fmaf(L.u[n], tlo, le2));
            }
        }
        if constexpr (E % 2 == 0 && DN_CF_PK) {
            // two elements per instruction: .x = element e, .y = element e + 1 (v2f: poisson_elem.h).  Elements beyond the mesh take part with
            // their differences of u (and their forcing) zeroed -- every contribution is linear in those --, on data clamped into the mesh
            v2f le1v = 0.f, le2v = 0.f;
#pragma unroll
            for (int e = 0; e < E; e += 2) {
                const v2f ok = {ex0 + e < p.nelx ? 1.f : 0.f, ex0 + e + 1 < p.nelx ? 1.f : 0.f};
                const v2f Ll = {L.u[e], L.u[e + 1]}, Lr = {L.u[e + 1], L.u[e + 2]}, Ul = {U.u[e], U.u[e + 1]}, Ur = {U.u[e + 1], U.u[e + 2]};
                const v2f UX = ok * (Lr - Ll), UY = ok * (Ul - Ll), UXY = ok * (Ur - Ul) - UX;
                v2f P = mx0, Q = 0.f, Pp = my0, Qp = 0.f;
                if constexpr (HAS_NU) {
                    const v2f N0 = {L.n[e], L.n[e + 1]}, Nr = {L.n[e + 1], L.n[e + 2]}, Nu = {U.n[e], U.n[e + 1]}, Nur = {U.n[e + 1], U.n[e + 2]};
                    const v2f NX = Nr - N0, NY = Nu - N0, NXY = (Nur - Nu) - NX;
                    P = vfma(mx1, NX, mx0 * N0);
                    Q = vfma(mx1, NXY, mx0 * NY);
                    Pp = vfma(my1, NY, my0 * N0);
                    Qp = vfma(my1, NXY, my0 * NX);
                }
                const v2f A0 = vfma(my1, Q, my0 * P), A1 = vfma(my2, Q, my1 * P), A2 = vfma(my3, Q, my2 * P);
                const v2f B0 = vfma(mx1, Qp, mx0 * Pp), B1 = vfma(mx2, Qp, mx1 * Pp), B2 = vfma(mx3, Qp, mx2 * Pp);
                const v2f tX0 = vfma(UXY, A1, UX * A0), tX1 = vfma(UXY, A2, UX * A1);
                const v2f tY0 = vfma(UXY, B1, UY * B0), tY1 = vfma(UXY, B2, UY * B1);
                le1v += vfma(h1, vfma(UXY, tY1, UY * tY0), h0 * vfma(UXY, tX1, UX * tX0));
                v2f cU0 = 0.f, cUX = k0 * tX0, cUY = k1 * tY0, cUXY = vfma(k0, tX1, k1 * tY1);
                if constexpr (HAS_F && !FMASS) {
                    const v2f U0 = Ll;
                    const v2f Fl = {L.f[e], L.f[e + 1]}, Fr = {L.f[e + 1], L.f[e + 2]}, Fu = {U.f[e], U.f[e + 1]}, Fur = {U.f[e + 1], U.f[e + 2]};
                    const v2f F0 = ok * Fl, FX = ok * (Fr - Fl), FY = ok * (Fu - Fl), FXY = ok * (Fur - Fu) - FX;
                    const v2f S0 = vfma(mx1, FX, mx0 * F0), S1 = vfma(mx1, FXY, mx0 * FY);
                    const v2f T0 = vfma(mx2, FX, mx1 * F0), T1 = vfma(mx2, FXY, mx1 * FY);
                    const v2f L0 = vfma(my1, S1, my0 * S0), LX = vfma(my1, T1, my0 * T0);
                    const v2f LY = vfma(my2, S1, my1 * S0), LXY = vfma(my2, T1, my1 * T0);
                    le2v += vfma(LXY, UXY, vfma(LY, UY, vfma(LX, UX, L0 * U0)));
                    cU0 = nb * L0;
                    cUX = vfma(nb, LX, cUX);
                    cUY = vfma(nb, LY, cUY);
                    cUXY = vfma(nb, LXY, cUXY);
                }
                const v2f g01 = cUX - cUXY, g10 = cUY - cUXY, g00 = (cU0 - cUX) - g10;
                o[e] += g00.x;
                o[e + 1] += g01.x + g00.y;
                o[e + 2] += g01.y;
                cout[e] += g10.x;
                cout[e + 1] += cUXY.x + g10.y;
                cout[e + 2] += cUXY.y;
                __builtin_amdgcn_sched_barrier(0);      // one pair at a time: interleaving the pairs doubles the live set
            }
            le1 += le1v.x + le1v.y;
            le2 += le2v.x + le2v.y;
        } else {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (ex0 + e < p.nelx) {       // elements beyond the domain are skipped (and: scheduling fence between elements)
                const float U0 = L.u[e], UX = L.u[e + 1] - L.u[e], UY = U.u[e] - L.u[e], UXY = (U.u[e + 1] - U.u[e]) - UX;
                float P = mx0, Q = 0.f, Pp = my0, Qp = 0.f;
                if constexpr (HAS_NU) {
                    const float N0 = L.n[e], NX = L.n[e + 1] - N0, NY = U.n[e] - N0, NXY = (U.n[e + 1] - U.n[e]) - NX;
                    P = fmaf(mx1, NX, mx0 * N0);
                    Q = fmaf(mx1, NXY, mx0 * NY);
                    Pp = fmaf(my1, NY, my0 * N0);
                    Qp = fmaf(my1, NXY, my0 * NX);
                }
                const float A0 = fmaf(Q, my1, P * my0), A1 = fmaf(Q, my2, P * my1), A2 = fmaf(Q, my3, P * my2);
                const float B0 = fmaf(Qp, mx1, Pp * mx0), B1 = fmaf(Qp, mx2, Pp * mx1), B2 = fmaf(Qp, mx3, Pp * mx2);
                const float tX0 = fmaf(UXY, A1, UX * A0), tX1 = fmaf(UXY, A2, UX * A1);
                const float tY0 = fmaf(UXY, B1, UY * B0), tY1 = fmaf(UXY, B2, UY * B1);
                le1 += fmaf(h1, fmaf(UXY, tY1, UY * tY0), h0 * fmaf(UXY, tX1, UX * tX0));
                float cU0 = 0.f, cUX = k0 * tX0, cUY = k1 * tY0, cUXY = fmaf(k0, tX1, k1 * tY1);
                if constexpr (HAS_F && !FMASS) {
                    const float F0 = L.f[e], FX = L.f[e + 1] - F0, FY = U.f[e] - F0, FXY = (U.f[e + 1] - U.f[e]) - FX;
                    const float S0 = fmaf(mx1, FX, mx0 * F0), S1 = fmaf(mx1, FXY, mx0 * FY);
                    const float T0 = fmaf(mx2, FX, mx1 * F0), T1 = fmaf(mx2, FXY, mx1 * FY);
                    const float L0 = fmaf(my1, S1, my0 * S0), LX = fmaf(my1, T1, my0 * T0);
                    const float LY = fmaf(my2, S1, my1 * S0), LXY = fmaf(my2, T1, my1 * T0);
                    le2 += fmaf(LXY, UXY, fmaf(LY, UY, fmaf(LX, UX, L0 * U0)));
                    cU0 = nb * L0;
                    cUX = fmaf(nb, LX, cUX);
                    cUY = fmaf(nb, LY, cUY);
                    cUXY = fmaf(nb, LXY, cUXY);
                }
                const float g01 = cUX - cUXY, g10 = cUY - cUXY;
                o[e] += (cU0 - cUX) - g10;
                o[e + 1] += g01;
                cout[e] += g10;
                cout[e + 1] += cUXY;
            }
        }
        }
        e1_acc = fmaf(cnt, le1, e1_acc);
        e2_acc = fmaf(cnt, le2, e2_acc);
        if (W > 1 && chain_dn && ey == sb0) {
            // the strip's first row still lacks the contributions of the layer below it, which the strip below computes at its very end:
            // park this half in LDS (each thread re-reads its own words) and finish the row after the march
            const unsigned s0 = lds_addr(&cf_seam[W > 1 ? sub - 1 : 0][0][tid]);
#pragma unroll
            for (int n = 0; n <= NW; ++n) lds_st(s0 + (4 * NW + 4 + n) * CF_TS * 4, o[n]);
#pragma unroll
            for (int n = 0; n < NW; ++n) lds_st(s0 + (5 * NW + 5 + n) * CF_TS * 4, L.keep[n]);
        } else {
            emit_row(o, L.keep, ey, ey >= r_from, L.keepx);
        }
    };
    // chained strips: row / carry hand-over with the neighbouring strip of the workgroup.  Thread t of one strip talks to thread t of the
    // other, which sits in the wave of the same number: one flag per wave and direction
    auto publish_row = [&](const CfRow<E>& r) {
        const unsigned s0 = lds_addr(&cf_seam[W > 1 ? sub - 1 : 0][0][tid]);
#pragma unroll
        for (int n = 0; n <= NW; ++n) {
            lds_st(s0 + n * CF_TS * 4, r.u[n]);
            if constexpr (HAS_NU) lds_st(s0 + (NW + 1 + n) * CF_TS * 4, r.n[n]);
            if constexpr (FMASS) lds_st(s0 + (2 * NW + 2 + n) * CF_TS * 4, r.g[n]);
            else if constexpr (HAS_F) lds_st(s0 + (2 * NW + 2 + n) * CF_TS * 4, r.f[n]);
        }
        if (lane == 0) lds_st(lds_addr(&cf_sflag[W > 1 ? sub - 1 : 0][0][wv]), 1u);
    };
    auto fetch_row = [&](CfRow<E>& r) {
        spin_until(&cf_sflag[sub][0][wv], 1u);
        const unsigned s0 = lds_addr(&cf_seam[sub][0][tid]);
#pragma unroll
        for (int n = 0; n <= NW; ++n) lds_ld_nowait(r.u[n], s0 + n * CF_TS * 4);
        lds_wait(r.u);
        if constexpr (HAS_NU) {
#pragma unroll
            for (int n = 0; n <= NW; ++n) lds_ld_nowait(r.n[n], s0 + (NW + 1 + n) * CF_TS * 4);
            lds_wait(r.n);
        }
        if constexpr (FMASS) {
#pragma unroll
            for (int n = 0; n <= NW; ++n) lds_ld_nowait(r.g[n], s0 + (2 * NW + 2 + n) * CF_TS * 4);
            lds_wait(r.g);
        } else if constexpr (HAS_F) {
#pragma unroll
            for (int n = 0; n <= NW; ++n) lds_ld_nowait(r.f[n], s0 + (2 * NW + 2 + n) * CF_TS * 4);
            lds_wait(r.f);
        }
    };
    auto publish_carry = [&](const float (&c)[NW + 1]) {
        const unsigned s0 = lds_addr(&cf_seam[sub][0][tid]);
#pragma unroll
        for (int n = 0; n <= NW; ++n) lds_st(s0 + (3 * NW + 3 + n) * CF_TS * 4, c[n]);
        if (lane == 0) lds_st(lds_addr(&cf_sflag[sub][1][wv]), 1u);
    };

    auto set_prio = [&](int e) {
#if DN_PRIO_ROT
        switch (((ey_end - e) >> 1) & 3) {          // progress-dependent wave priority (profiles/README.md)
            case 0: __builtin_amdgcn_s_setprio(0); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            default: __builtin_amdgcn_s_setprio(3); break;
        }
#endif
    };

    constexpr std::true_type FRESH{};
    constexpr std::false_type LANDED{};
    if (active) {                     // (a workgroup's unused sub-strips only join the final reduction)
    CfRow<E> RA, RB;
    float carryA[NW + 1], carryB[NW + 1];
#pragma unroll
    for (int n = 0; n <= NW; ++n) carryA[n] = carryB[n] = 0.f;
    row_issue(ey_begin, RA);
    row_bc(RA);
    fstage(RA);
    if constexpr (W > 1) {
        if (chain_dn) publish_row(RA);
    }
    int ey = ey_begin;
    bool odd = false;
#if DN_CF_PF == 2
    {
        // software pipeline, two raw rows ahead: WR0 / WR1 hold rows k + 1 and k + 2 while layer k - 1 runs.  Rows past the strip's last one
        // are re-reads of that row (cache hits, no HBM traffic; a wave-uniform branch around the loads would drain the pipeline)
        CfRow<E> WR0, WR1;
        auto consume = [&](CfRow<E>& r, const CfRow<E>& WR) {
            r = WR;
            row_bc(r);
            fstage(r);
        };
        row_issue(min(ey_begin + 1, ey_end), WR0);
        row_issue(min(ey_begin + 2, ey_end), WR1);
        for (; ey + 1 < ey_end; ey += 2) {
            set_prio(ey);
            consume(RB, WR0);
            row_issue(min(ey + 3, ey_end), WR0);
            flush_store();
            layer(FRESH, ey, RA, RB, carryA, carryB);
            consume(RA, WR1);
            row_issue(min(ey + 4, ey_end), WR1);
            flush_store();
            layer(FRESH, ey + 1, RB, RA, carryB, carryA);
        }
        if (ey < ey_end) {
            set_prio(ey);
            consume(RB, WR0);
            flush_store();
            layer(FRESH, ey, RA, RB, carryA, carryB);
            odd = true;
        }
        flush_store();
    }
#elif DN_CF_PF
    {
        // software pipeline: WR holds the raw row k + 2 while layer k runs; consume = Dirichlet select + copy into the row state
        CfRow<E> WR;
        auto consume = [&](CfRow<E>& r) {
            r = WR;
            row_bc(r);
            fstage(r);
        };
        row_issue(min(ey_begin + 1, ey_end), WR);
        for (; ey + 1 < ey_end; ey += 2) {
            set_prio(ey);
            consume(RB);
            row_issue(min(ey + 2, ey_end), WR);
            flush_store();
            layer(FRESH, ey, RA, RB, carryA, carryB);
            consume(RA);
            row_issue(min(ey + 3, ey_end), WR);
            flush_store();
            layer(FRESH, ey + 1, RB, RA, carryB, carryA);
        }
        if (ey < ey_end) {
            set_prio(ey);
            consume(RB);
            flush_store();
            layer(FRESH, ey, RA, RB, carryA, carryB);
            odd = true;
        }
        flush_store();
    }
#else
    const int n_main = chain_up ? ey_end - 1 : ey_end;         // chain_up: the strip's last layer takes its upper row from the strip above (LDS)
    for (; ey + 1 < n_main; ey += 2) {
        set_prio(ey);
        row_issue(ey + 1, RB);
        layer(FRESH, ey, RA, RB, carryA, carryB);
        row_issue(ey + 2, RA);
        layer(FRESH, ey + 1, RB, RA, carryB, carryA);
    }
    if (ey < n_main) {
        set_prio(ey);
        row_issue(ey + 1, RB);
        layer(FRESH, ey, RA, RB, carryA, carryB);
        odd = true;
        ++ey;
    }
    if constexpr (W > 1) {
        if (chain_up) {               // ey == ey_end - 1
            __builtin_amdgcn_s_setprio(0);
            if (odd) { fetch_row(RA); layer(LANDED, ey, RB, RA, carryB, carryA); publish_carry(carryA); }
            else { fetch_row(RB); layer(LANDED, ey, RA, RB, carryA, carryB); publish_carry(carryB); }
        }
    }
#endif
    if (top_row) {                // the last logical row of the domain: only the layer below it contributes
        float o[NW + 1], keep[NW];
#pragma unroll
        for (int n = 0; n <= NW; ++n) o[n] = odd ? carryB[n] : carryA[n];
#pragma unroll
        for (int n = 0; n < NW; ++n) keep[n] = odd ? RB.keep[n] : RA.keep[n];
        emit_row(o, keep, p.ny - 1, true, odd ? RB.keepx : RA.keepx);
        flush_store();
    }
    if constexpr (W > 1) {
        if (chain_dn) {               // finish the strip's first row: parked half + the carry of the strip below
            spin_until(&cf_sflag[sub - 1][1][wv], 1u);
            const unsigned s0 = lds_addr(&cf_seam[sub - 1][0][tid]);
            float o[NW + 1], cy[NW + 1], keep[NW];
#pragma unroll
            for (int n = 0; n <= NW; ++n) lds_ld_nowait(o[n], s0 + (4 * NW + 4 + n) * CF_TS * 4);
            lds_wait(o);
#pragma unroll
            for (int n = 0; n <= NW; ++n) lds_ld_nowait(cy[n], s0 + (3 * NW + 3 + n) * CF_TS * 4);
            lds_wait(cy);
#pragma unroll
            for (int n = 0; n < NW; ++n) lds_ld_nowait(keep[n], s0 + (5 * NW + 5 + n) * CF_TS * 4);
            lds_wait(keep);
#pragma unroll
            for (int n = 0; n <= NW; ++n) o[n] += cy[n];
            emit_row(o, keep, sb0, true);
            flush_store();
        }
    }
    }                                 // active

#ifdef DN_STAMP2D
    if (tid == 0) {
        const unsigned slot = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        if (slot < 8192u) {
            dn_stamp2d_buf[4 * slot] = __builtin_amdgcn_s_memtime() - stamp_t0;
            dn_stamp2d_buf[4 * slot + 1] = stamp_rt0;
            dn_stamp2d_buf[4 * slot + 2] = __builtin_amdgcn_s_memrealtime();
        }
    }
#endif
    if constexpr (W > 1) {            // a hand-over poll that ran into its bound: NaN sums + the sticky error word (never silent)
        e1_acc += spin_poison;
        sq_acc += spin_poison;
        if (spin_poison != spin_poison && p.counter != nullptr && (threadIdx.x & 63u) == 0u) atomicOr(p.counter + DN_WS_ERRWORD, 1u);
    }
    if (p.want_sums) finish_sums(p, e1_acc, e2_acc, sq_acc, (int)threadIdx.x, (int)blockDim.x, red, &last_flag);
#ifdef DN_STAMP2D
    if (tid == 0) {
        const unsigned slot = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        if (slot < 8192u) dn_stamp2d_buf[4 * slot + 3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

template <int E, bool VEC, int FL>
static void cf_launch_one(const PoissonParams& pp, const Geom2D& g, int batch, hipStream_t s) {
    if constexpr (E == 4 && VEC && DN_CF_W > 1 && (FL & CF_UA) == 0 && !DN_CF_REV && !DN_CF_PF) {
        if (g.W == DN_CF_W) {         // chained sub-strips (plan2d): W strips per workgroup
            hipLaunchKernelGGL((poisson2d_q1_cf_kernel<E, VEC, FL, DN_CF_W>), dim3(g.chunks, (g.strips + DN_CF_W - 1) / DN_CF_W, batch),
                               dim3(CF_TS * DN_CF_W), 0, s, pp);
            return;
        }
    }
    hipLaunchKernelGGL((poisson2d_q1_cf_kernel<E, VEC, FL, 1>), dim3(g.chunks, g.strips, batch), dim3(g.T), 0, s, pp);
}

template <int E, bool VEC, int UAF = 0>
static void cf_launch_flags(const PoissonParams& pp, const Geom2D& g, int batch, hipStream_t s) {
    const bool any = pp.bc[0].kind >= 0 || pp.bc[1].kind >= 0;
    bool u8c = any, packed = false;
    for (int k = 0; k < 2; ++k) {
        if (pp.bc[k].kind >= 0 && (!pp.bc[k].mask_is_u8 || pp.bc[k].field)) u8c = false;
        if (pp.bc[k].kind == DN_MASK_BITS || pp.bc[k].kind == DN_MASK_BOX) packed = true;     // dn_poisson_apply admits no mix with mask images
    }
    const int nbits = (pp.bc[0].kind == DN_MASK_BITS) + (pp.bc[1].kind == DN_MASK_BITS);
    const int nf = (pp.nu ? CF_NU : 0) | (pp.f ? CF_F : 0);
#define DN_CF(FLAGS)                                                                         \
    (!any ? cf_launch_one<E, VEC, (FLAGS) | UAF>(pp, g, batch, s)                                  \
          : packed ? (nbits == 2 ? cf_launch_one<E, VEC, (FLAGS) | UAF | CF_BC_PACKED | CF_PK_NB2>(pp, g, batch, s)          \
                     : nbits == 1 ? cf_launch_one<E, VEC, (FLAGS) | UAF | CF_BC_PACKED | CF_PK_NB1>(pp, g, batch, s)        \
                                  : cf_launch_one<E, VEC, (FLAGS) | UAF | CF_BC_PACKED>(pp, g, batch, s))                   \
          : u8c ? cf_launch_one<E, VEC, (FLAGS) | UAF | CF_BC_U8C>(pp, g, batch, s)                \
                : cf_launch_one<E, VEC, (FLAGS) | UAF | CF_BC>(pp, g, batch, s))
    switch (nf) {
        case 0: DN_CF(0); break;
        case CF_NU: DN_CF(CF_NU); break;
        case CF_F: DN_CF(CF_F); break;
        default: DN_CF(CF_NU | CF_F); break;
    }
#undef DN_CF
}

int poisson2d_q1_cf_chain() { return DN_CF_W > 1 && !DN_CF_REV && !DN_CF_PF ? DN_CF_W : 1; }       // what plan2d may put into Geom2D::W

int launch_poisson2d_q1_cf(const PoissonParams& pp, const Geom2D& g, int batch, bool vec, hipStream_t s) {
    if (g.W > 1 && !(g.W == poisson2d_q1_cf_chain() && g.E == 4 && vec && g.T == CF_TS && g.R + 3 <= CF_NSLOT)) return DN_E_BADARG;
    if (g.ua) {                   // rows of 4 k + 1 nodes (plan2d): the vector kernel on 4-byte aligned rows, last node column in the last thread column
        if (g.E != 4 || g.W != 1 || (pp.nx & 3) != 1 || pp.nx < 9) return DN_E_BADARG;
#if DN_CF_REV || DN_CF_PF || DN_CF_DPPX
        return DN_E_UNSUPPORTED;  // (experiment builds: no 4 k + 1 form)
#else
        cf_launch_flags<4, true, CF_UA>(pp, g, batch, s);
        return 0;
#endif
    }
    if (g.E == 4 && vec) { cf_launch_flags<4, true>(pp, g, batch, s); return 0; }
    if (g.E == 2 && vec) { cf_launch_flags<2, true>(pp, g, batch, s); return 0; }
    if (g.E == 2) { cf_launch_flags<2, false>(pp, g, batch, s); return 0; }
    return DN_E_UNSUPPORTED;
}

}  // namespace dn
