// gauss_pt_eval (DiffNet/DiffNetFEM.py:7-18) for an arbitrary user table list, its adjoint, and the
// element->node assembly helper (e8_2d_poisson_mms.py:85-90, e8_3d_poisson_mms.py:78-87) with its adjoint.
//
// The reference issues one single-channel strided conv per Gauss point and concatenates; here one
// launch produces all G channels: a thread owns one element, keeps its nbf^nsd nodal values in
// registers and streams the G x nbf^nsd table out of LDS (wave-uniform address => broadcast reads).
// HBM: read 4 B/node (+ overlap served by L1/L2), write 4*G B/element -- write dominated.
// The adjoint writes every output exactly once and sums in a fixed order.  Standard FEM layout (stride = nbf - 1): tiled
// element-centric kernel (gpe_bwd_tiled_kernel: gradient planes read once, coalesced, 4 planes x R elements in flight
// per thread; contributions staged in LDS, nodes gathered from LDS); branch-free per-node gather kernels where the LDS
// tile does not fit (3-D Q2/Q3) and a general gather for arbitrary strides.  Assembly: branch-free per-node gather.
#include <cstdlib>

#include "dn_common.h"

namespace dn {

struct GpeGeom {
    int batch, nsd, nbf, stride, G;
    int n[3];      // nodes  x,y,z
    int nel[3];    // elements x,y,z
};

template <int NSD, int NB>
__global__ void __launch_bounds__(256) gpe_fwd_kernel(const float* __restrict__ in, const float* __restrict__ tables,
                                                      float* __restrict__ out, const GpeGeom g) {
    constexpr int NBT = NSD == 1 ? NB : (NSD == 2 ? NB * NB : NB * NB * NB);
    extern __shared__ float tab[];
    for (int i = threadIdx.x; i < g.G * NBT; i += blockDim.x) tab[i] = tables[i];
    __syncthreads();
    const int64_t nel_s = (int64_t)g.nel[0] * g.nel[1] * g.nel[2];
    const int64_t nps = (int64_t)g.n[0] * g.n[1] * g.n[2];
    const int64_t total = nel_s * g.batch;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(idx / nel_s);
        const int64_t e = idx - (int64_t)b * nel_s;
        const int ex = (int)(e % g.nel[0]);
        const int ey = (int)((e / g.nel[0]) % g.nel[1]);
        const int ez = (int)(e / ((int64_t)g.nel[0] * g.nel[1]));
        const float* src = in + (int64_t)b * nps + ((int64_t)ez * g.stride * g.n[1] + (int64_t)ey * g.stride) * g.n[0] + (int64_t)ex * g.stride;
        float v[NBT];
#pragma unroll
        for (int a = 0; a < NBT; ++a) {
            const int ib = a % NB, jb = (a / NB) % NB, kb = a / (NB * NB);
            v[a] = src[((int64_t)kb * g.n[1] + jb) * g.n[0] + ib];
        }
        float* dst = out + (int64_t)b * g.G * nel_s + e;
        for (int gi = 0; gi < g.G; ++gi) {
            float s = 0.f;
#pragma unroll
            for (int a = 0; a < NBT; ++a) s = fmaf(tab[gi * NBT + a], v[a], s);
            dst[(int64_t)gi * nel_s] = s;
        }
    }
}

template <int NSD, int NB>
__global__ void __launch_bounds__(256) gpe_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ tables,
                                                      float* __restrict__ gin, const GpeGeom g) {
    constexpr int NBT = NSD == 1 ? NB : (NSD == 2 ? NB * NB : NB * NB * NB);
    extern __shared__ float tab[];
    for (int i = threadIdx.x; i < g.G * NBT; i += blockDim.x) tab[i] = tables[i];
    __syncthreads();
    const int64_t nel_s = (int64_t)g.nel[0] * g.nel[1] * g.nel[2];
    const int64_t nps = (int64_t)g.n[0] * g.n[1] * g.n[2];
    const int64_t total = nps * g.batch;
    const int S = g.stride;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(idx / nps);
        const int64_t nd = idx - (int64_t)b * nps;
        const int x = (int)(nd % g.n[0]);
        const int y = (int)((nd / g.n[0]) % g.n[1]);
        const int z = (int)(nd / ((int64_t)g.n[0] * g.n[1]));
        // candidate (window, local index) pairs per axis: every window e with 0 <= x - e*S <= NB-1 (<= 4 of them)
        int ce[3][4], cl[3][4], cn[3];
        const int xyz[3] = {x, y, z};
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            cn[d] = 0;
            if (d >= NSD) { ce[d][0] = 0; cl[d][0] = 0; cn[d] = 1; continue; }
            // every element e with 0 <= xyz - e*S <= NB-1
            const int ehi = min(xyz[d] / S, g.nel[d] - 1);
            for (int e = ehi; e >= 0 && xyz[d] - e * S <= NB - 1 && cn[d] < 4; --e) { ce[d][cn[d]] = e; cl[d][cn[d]] = xyz[d] - e * S; ++cn[d]; }
        }
        float s = 0.f;
        const float* gsrc = gout + (int64_t)b * g.G * nel_s;
        for (int kz = cn[2] - 1; kz >= 0; --kz)
            for (int ky = cn[1] - 1; ky >= 0; --ky)
                for (int kx = cn[0] - 1; kx >= 0; --kx) {
                    const int64_t e = ((int64_t)ce[2][kz] * g.nel[1] + ce[1][ky]) * g.nel[0] + ce[0][kx];
                    const int aa = (cl[2][kz] * NB + cl[1][ky]) * NB + cl[0][kx];   // cl is 0 on unused axes
                    for (int gi = 0; gi < g.G; ++gi) s = fmaf(tab[gi * NBT + aa], gsrc[(int64_t)gi * nel_s + e], s);
                }
        gin[idx] = s;
    }
}

// Standard FEM layout (stride == NB - 1: neighbouring elements share one node layer): a node lies in at most two elements
// per axis -- e1 = min(x / S, nel - 1) with local index l1 = x - e1 S, and, when l1 == 0 and e1 >= 1, also e1 - 1 with local
// index S.  Branch-free gather: the <= 2^NSD candidate elements are addressed with clamped indices, their table-weighted
// sums over the G Gauss points run as independent chains (loads batched by the compiler), invalid candidates are
// selected away, and the partial sums are combined in a fixed order (lower element first, x fastest).
template <int NSD>
struct NodeCand {
    int e[3][2], l[3][2];
    bool ok[3][2];
};

template <int NSD, int NB>
__device__ __forceinline__ NodeCand<NSD> node_candidates(const int (&xyz)[3], const GpeGeom& g) {
    constexpr int S = NB - 1;
    NodeCand<NSD> c;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        if (d >= NSD) { c.e[d][0] = c.e[d][1] = 0; c.l[d][0] = c.l[d][1] = 0; c.ok[d][0] = false; c.ok[d][1] = true; continue; }
        const int e1 = min(xyz[d] / S, g.nel[d] - 1), l1 = xyz[d] - e1 * S;
        c.e[d][1] = e1; c.l[d][1] = l1; c.ok[d][1] = true;
        c.ok[d][0] = (l1 == 0) && (e1 >= 1);                 // slot 0: the lower neighbour, local index S
        c.e[d][0] = max(e1 - 1, 0); c.l[d][0] = S;
    }
    return c;
}

template <int NSD, int NB>
__global__ void __launch_bounds__(256) gpe_bwd_std_kernel(const float* __restrict__ gout, const float* __restrict__ tables,
                                                          float* __restrict__ gin, const GpeGeom g) {
    constexpr int NBT = NSD == 1 ? NB : (NSD == 2 ? NB * NB : NB * NB * NB);
    constexpr int NC = 1 << NSD;
    // table transposed to [local node][Gauss point], rows padded to a multiple of 4: a candidate's G weights are
    // contiguous and come in with float4 LDS reads
    extern __shared__ __attribute__((aligned(16))) float tabT[];
    const int GP = (g.G + 3) & ~3;
    for (int i = threadIdx.x; i < NBT * GP; i += blockDim.x) {
        const int a = i / GP, gi = i - a * GP;
        tabT[i] = gi < g.G ? tables[gi * NBT + a] : 0.f;
    }
    __syncthreads();
    const unsigned nel_s = (unsigned)(g.nel[0] * g.nel[1] * g.nel[2]);
    const int64_t nps = (int64_t)g.n[0] * g.n[1] * g.n[2];
    const int64_t total = nps * g.batch;
    const unsigned gmax = (unsigned)(g.G - 1);
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(idx / nps);
        const unsigned nd = (unsigned)(idx - (int64_t)b * nps);
        const int xyz[3] = {(int)(nd % (unsigned)g.n[0]), (int)((nd / (unsigned)g.n[0]) % (unsigned)g.n[1]),
                            (int)(nd / ((unsigned)g.n[0] * (unsigned)g.n[1]))};
        const NodeCand<NSD> c = node_candidates<NSD, NB>(xyz, g);
        const float* gsrc = gout + (int64_t)b * g.G * nel_s;
        unsigned eo[NC], ao[NC];
        bool ok[NC];
#pragma unroll
        for (int q = 0; q < NC; ++q) {               // q bits: x = bit 0, y = bit 1, z = bit 2; 0 = lower neighbour
            const int cx = q & 1, cy = NSD > 1 ? (q >> 1) & 1 : 1, cz = NSD > 2 ? (q >> 2) & 1 : 1;
            eo[q] = ((unsigned)c.e[2][cz] * (unsigned)g.nel[1] + (unsigned)c.e[1][cy]) * (unsigned)g.nel[0] + (unsigned)c.e[0][cx];
            ao[q] = (unsigned)((c.l[2][cz] * NB + c.l[1][cy]) * NB + c.l[0][cx]) * (unsigned)GP;
            ok[q] = c.ok[0][cx] && (NSD > 1 ? c.ok[1][cy] : true) && (NSD > 2 ? c.ok[2][cz] : true);
        }
        float part[NC];
#pragma unroll
        for (int q = 0; q < NC; ++q) part[q] = 0.f;
        for (int g0 = 0; g0 < GP; g0 += 4) {
            // the padded weights are zero; their (clamped, valid) loads contribute nothing
            const unsigned o0 = (unsigned)g0 * nel_s, o1 = min((unsigned)g0 + 1, gmax) * nel_s, o2 = min((unsigned)g0 + 2, gmax) * nel_s,
                           o3 = min((unsigned)g0 + 3, gmax) * nel_s;
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                const float4 w = *reinterpret_cast<const float4*>(&tabT[ao[q] + g0]);
                float p = part[q];
                p = fmaf(w.x, gsrc[o0 + eo[q]], p);
                p = fmaf(w.y, gsrc[o1 + eo[q]], p);
                p = fmaf(w.z, gsrc[o2 + eo[q]], p);
                p = fmaf(w.w, gsrc[o3 + eo[q]], p);
                part[q] = p;
            }
        }
        float sum = 0.f;
#pragma unroll
        for (int q = 0; q < NC; ++q) sum += ok[q] ? part[q] : 0.f;
        gin[idx] = sum;
    }
}

// Same gather with the table as given ([Gauss point][local node], one LDS read per weight): faster than the transposed
// form for 3-D Q2/Q3, where most nodes have few valid candidates and the 112-byte rows of the transposed table conflict.
template <int NSD, int NB>
__global__ void __launch_bounds__(256) gpe_bwd_std_scalar_kernel(const float* __restrict__ gout, const float* __restrict__ tables,
                                                                 float* __restrict__ gin, const GpeGeom g) {
    constexpr int NBT = NSD == 1 ? NB : (NSD == 2 ? NB * NB : NB * NB * NB);
    constexpr int NC = 1 << NSD;
    extern __shared__ __attribute__((aligned(16))) float tabT[];
    for (int i = threadIdx.x; i < g.G * NBT; i += blockDim.x) tabT[i] = tables[i];
    __syncthreads();
    const unsigned nel_s = (unsigned)(g.nel[0] * g.nel[1] * g.nel[2]);
    const int64_t nps = (int64_t)g.n[0] * g.n[1] * g.n[2];
    const int64_t total = nps * g.batch;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(idx / nps);
        const unsigned nd = (unsigned)(idx - (int64_t)b * nps);
        const int xyz[3] = {(int)(nd % (unsigned)g.n[0]), (int)((nd / (unsigned)g.n[0]) % (unsigned)g.n[1]),
                            (int)(nd / ((unsigned)g.n[0] * (unsigned)g.n[1]))};
        const NodeCand<NSD> c = node_candidates<NSD, NB>(xyz, g);
        const float* gsrc = gout + (int64_t)b * g.G * nel_s;
        unsigned eo[NC], ao[NC];
        bool ok[NC];
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            const int cx = q & 1, cy = NSD > 1 ? (q >> 1) & 1 : 1, cz = NSD > 2 ? (q >> 2) & 1 : 1;
            eo[q] = ((unsigned)c.e[2][cz] * (unsigned)g.nel[1] + (unsigned)c.e[1][cy]) * (unsigned)g.nel[0] + (unsigned)c.e[0][cx];
            ao[q] = (unsigned)((c.l[2][cz] * NB + c.l[1][cy]) * NB + c.l[0][cx]);
            ok[q] = c.ok[0][cx] && (NSD > 1 ? c.ok[1][cy] : true) && (NSD > 2 ? c.ok[2][cz] : true);
        }
        float part[NC];
#pragma unroll
        for (int q = 0; q < NC; ++q) part[q] = 0.f;
        for (int gi = 0; gi < g.G; ++gi) {
            const float* gp = gsrc + (size_t)gi * nel_s;
            const float* tp = tabT + gi * NBT;
#pragma unroll
            for (int q = 0; q < NC; ++q) part[q] = fmaf(tp[ao[q]], gp[eo[q]], part[q]);
        }
        float sum = 0.f;
#pragma unroll
        for (int q = 0; q < NC; ++q) sum += ok[q] ? part[q] : 0.f;
        gin[idx] = sum;
    }
}

// (Round 4: a marching 2-D Q1 adjoint -- independent waves of 63 element columns, whole rows read once, hand-over by ds_bpermute -- was built and
// measured against this kernel through the C ABI, 512^2 B = 16: 2 x 2 rule 17.2 vs 16.5 us, 3 x 3 35.8 vs 27.3 us (6.1 TB/s), 4 x 4 65.6 vs 46.8 us;
// not kept: profiles/r4_adjoint_sizes.txt.  The 56 us / 2.97 TB/s of profiles/r2_ops_b.txt was the HOST time of the autograd call, not this kernel.)
// Element-centric adjoint for the standard layout (the fast path): a workgroup owns a tile of 32 x 32 elements (2-D) or
// 32 x 8 x 8 elements (3-D); thread (tx, ty) walks the tile's element rows / planes.  Phase A: per element the G
// gradient values are read once (coalesced) and its NB^NSD nodal contributions sum_g tab[g][a] gout[g] (table through
// scalar loads: uniform index) go to LDS as contrib[a][element] -- one slot per (element, a), no conflicts.  Phase B: the
// tile's nodes are gathered from LDS (<= 2^NSD slots each, lower element first: fixed order) and written once.  Tiles
// overlap by one element layer on the high side (recomputed) and own the nodes above their low face.
// SMALL (round 4, 3-D Q2 / Q3): tiles of 8 x 8 x 8 (Q2: 55 KB of LDS) / 8 x 8 x 4 (Q3: 64 KB) elements, elements dealt to the threads in
// tile order -- the 32 x 8 x 8 tile needs 221 KB at Q2, and the per-node gather that served these meshes instead ran at 113 GB/s (27 Gauss points x
// 8 candidate elements = 216 scattered loads per node: profiles/r2_ops_b.txt).
template <int NSD, int NB, bool SMALL = false>
__global__ void __launch_bounds__(256) gpe_bwd_tiled_kernel(const float* __restrict__ gout, const float* __restrict__ tables,
                                                            float* __restrict__ gin, const GpeGeom g, const int tiles_x, const int tiles_y,
                                                            const int tiles_z) {
    constexpr int S = NB - 1;
    constexpr int NBT = NSD == 2 ? NB * NB : NB * NB * NB;
    constexpr int TX = SMALL ? 8 : 32, TY = SMALL ? 8 : (NSD == 2 ? 32 : 8), TZ = SMALL ? (NB >= 4 ? 4 : 8) : (NSD == 2 ? 1 : 8);       // elements per tile (incl. the recomputed layer)
    static_assert(!SMALL || NSD == 3, "small tiles: 3-D");
    constexpr int TE = TX * TY * TZ;
    constexpr int LX = TX * S + 1, LY = TY * S + 1, LZ = NSD == 2 ? 1 : TZ * S + 1;
    extern __shared__ float contrib[];                                          // [NBT][TZ][TY][TX]
    const int tx = SMALL ? (int)(threadIdx.x & 7) : (int)(threadIdx.x & 31), ty = SMALL ? (int)((threadIdx.x >> 3) & 7) : (int)(threadIdx.x >> 5);       // 32 x 8 threads (small: 8 x 8 x 4)
    const int tzs = (int)(threadIdx.x >> 6);                                    // small tiles: the thread's first element plane
    unsigned t = blockIdx.x;
    {   // XCD-aware decode: x-neighbouring tiles share cache lines of every element row (rows are not line-aligned); a contiguous range of
        // the logical tile order per XCD lets them meet in one L2
        const unsigned nwg = gridDim.x, xcd = t & 7u, idx = t >> 3, base = nwg >> 3, rem = nwg & 7u;
        t = xcd * base + min(xcd, rem) + idx;
    }
    const int tix = (int)(t % (unsigned)tiles_x); t /= (unsigned)tiles_x;
    const int tiy = (int)(t % (unsigned)tiles_y); t /= (unsigned)tiles_y;
    const int tiz = (int)(t % (unsigned)tiles_z);
    const int b = (int)(t / (unsigned)tiles_z);
    const int ex0 = tix * (TX - 1), ey0 = tiy * (TY - 1), ez0 = NSD == 2 ? 0 : tiz * (TZ - 1);
    const unsigned nel_s = (unsigned)(g.nel[0] * g.nel[1] * g.nel[2]);
    const int64_t nps = (int64_t)g.n[0] * g.n[1] * g.n[2];
    const float* gsrc = gout + (int64_t)b * g.G * nel_s;
    // phase A.  this thread's elements: 2-D: (tx, ty + 8 r), r < 4;  3-D: (tx, ty, r), r < 8.  The Gauss-point planes are
    // walked four at a time and all R elements' loads of a chunk are issued before their FMAs (4 R loads in flight per
    // thread: the planes are megabytes apart, so a one-load-at-a-time loop is latency bound).
    constexpr int R = TE / 256;
    const int ex = ex0 + tx;
    unsigned eoff[R];
    bool valid[R];
    float c[R][NBT];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int ly_e = NSD == 2 ? ty + 8 * r : ty, lz_e = NSD == 2 ? 0 : (SMALL ? tzs + 4 * r : r);
        const int ey = ey0 + ly_e, ez = ez0 + lz_e;
        valid[r] = ex < g.nel[0] && ey < g.nel[1] && ez < g.nel[2];
        eoff[r] = ((unsigned)min(ez, g.nel[2] - 1) * (unsigned)g.nel[1] + (unsigned)min(ey, g.nel[1] - 1)) * (unsigned)g.nel[0] +
                  (unsigned)min(ex, g.nel[0] - 1);
#pragma unroll
        for (int a = 0; a < NBT; ++a) c[r][a] = 0.f;
    }
    const int gmax = g.G - 1;
    for (int g0 = 0; g0 < g.G; g0 += 4) {
        float v[R][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t po = (size_t)min(g0 + k, gmax) * nel_s;          // clamped: the duplicate gets a zero weight below
#pragma unroll
            for (int r = 0; r < R; ++r) v[r][k] = gsrc[po + eoff[r]];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool live = g0 + k <= gmax;
            const float* tp = tables + min(g0 + k, gmax) * NBT;
#pragma unroll
            for (int a = 0; a < NBT; ++a) {
                const float w = live ? tp[a] : 0.f;
#pragma unroll
                for (int r = 0; r < R; ++r) c[r][a] = fmaf(w, v[r][k], c[r][a]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int ly_e = NSD == 2 ? ty + 8 * r : ty, lz_e = NSD == 2 ? 0 : (SMALL ? tzs + 4 * r : r);
        const int slot = (lz_e * TY + ly_e) * TX + tx;
#pragma unroll
        for (int a = 0; a < NBT; ++a) contrib[a * TE + slot] = valid[r] ? c[r][a] : 0.f;      // elements beyond the mesh contribute zero
    }
    __syncthreads();
    // phase B.  owned nodes: local index 1 .. (T-1) S per axis (0 too on the first tile; the whole tile on the last)
    float* dst = gin + (int64_t)b * nps;
    const int lox = tix == 0 ? 0 : 1, loy = tiy == 0 ? 0 : 1, loz = (NSD == 2 || tiz == 0) ? 0 : 1;
    const int HX = tix == tiles_x - 1 ? LX - 1 : (TX - 1) * S, HY = tiy == tiles_y - 1 ? LY - 1 : (TY - 1) * S;
    const int HZ = NSD == 2 ? 0 : (tiz == tiles_z - 1 ? LZ - 1 : (TZ - 1) * S);
    for (int i = threadIdx.x; i < LX * LY * LZ; i += 256) {
        const int lx = i % LX, ly = (i / LX) % LY, lz = i / (LX * LY);
        const int x = ex0 * S + lx, y = ey0 * S + ly, z = ez0 * S + lz;
        if (!(lx >= lox && lx <= HX && ly >= loy && ly <= HY && lz >= loz && lz <= HZ && x < g.n[0] && y < g.n[1] && z < g.n[2])) continue;
        // per axis: upper element e1 = min(l / S, T - 1) with local index l - e1 S; lower neighbour (local S) if that is 0 and e1 >= 1
        int e1[3], l1[3];
        bool two[3];
        const int lxyz[3] = {lx, ly, lz};
        const int TT[3] = {TX, TY, TZ};
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            if (d >= NSD) { e1[d] = 0; l1[d] = 0; two[d] = false; continue; }
            e1[d] = min(lxyz[d] / S, TT[d] - 1);
            l1[d] = lxyz[d] - e1[d] * S;
            two[d] = l1[d] == 0 && e1[d] >= 1;
        }
        float sum = 0.f;
#pragma unroll
        for (int q = 0; q < (1 << NSD); ++q) {          // bit = 0: lower neighbour (if any), 1: upper element; x fastest
            const int cx = q & 1, cy = (q >> 1) & 1, cz = NSD > 2 ? (q >> 2) & 1 : 1;
            const bool ok = (cx || two[0]) && (cy || two[1]) && (NSD == 2 || cz || two[2]);
            const int exl = cx ? e1[0] : max(e1[0] - 1, 0), eyl = cy ? e1[1] : max(e1[1] - 1, 0), ezl = cz ? e1[2] : max(e1[2] - 1, 0);
            const int ia = cx ? l1[0] : S, ja = cy ? l1[1] : S, ka = NSD == 2 ? 0 : (cz ? l1[2] : S);
            const float v = contrib[((ka * NB + ja) * NB + ia) * TE + (ezl * TY + eyl) * TX + exl];
            sum += ok ? v : 0.f;
        }
        dst[((size_t)z * g.n[1] + y) * g.n[0] + x] = sum;
    }
}

// Marching adjoint for 3-D Q1 (standard layout, any number G of tables): the scheme of the fused 3-D kernel without the physics.
// A workgroup owns 16 x 16 elements of a plane (tiles step by 15) and marches over R element planes (+ one recomputed plane below the
// strip); a thread reads its element's G values once per plane (coalesced dwords), forms the 8 nodal contributions sum_g tab[g][a] v_g
// (table entries through scalar loads: uniform index), carries the four of the upper plane in registers and hands the in-plane ones
// over to the right / upper / upper-right neighbour (ds_bpermute + one LDS slot, one barrier per plane).  Every Gauss-point value is
// read once (the tiled kernel re-reads 35 % of them and gathers the nodes with per-node index arithmetic): 128^3 x 2, G = 8: 80 -> ~45 us.
// Summation order per node is fixed (own element, left, lower, then the plane below): deterministic.
__device__ __forceinline__ float gpe_from_left(float v, int from, float nf) {
    return nf * __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(from, __builtin_bit_cast(int, v)));
}

template <bool G8>            // G8: exactly 8 tables (the 2 x 2 x 2 rule): the next plane's 8 values are in flight while this plane is reduced
__global__ void __launch_bounds__(256) gpe_bwd_march3d_q1_kernel(const float* __restrict__ gout, const float* __restrict__ tables,
                                                                  float* __restrict__ gin, const GpeGeom g, const int chunks_x,
                                                                  const int tiles_y, const int strips_z, const int R) {
    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * 16 + tx;
    unsigned lid = blockIdx.x;
    {   // XCD-aware decode (workgroups are dealt round-robin to the 8 XCDs): every XCD gets a contiguous range of the logical order, so
        // x-neighbouring tiles -- which share every 128-byte line of a 64-byte tile row -- meet in one L2
        const unsigned nwg = gridDim.x, xcd = lid & 7u, idx = lid >> 3, base = nwg >> 3, rem = nwg & 7u;
        lid = xcd * base + min(xcd, rem) + idx;
    }
    const int chunk = (int)(lid % (unsigned)chunks_x); lid /= (unsigned)chunks_x;
    const int tile = (int)(lid % (unsigned)tiles_y); lid /= (unsigned)tiles_y;
    const int strip = (int)(lid % (unsigned)strips_z), b = (int)(lid / (unsigned)strips_z);
    const int ex = chunk * 15 + tx, ey = tile * 15 + ty;                 // element == its lower-left node
    const bool owner = !(chunk > 0 && tx == 0) && !(tile > 0 && ty == 0);
    const bool node_ok = ex < g.n[0] && ey < g.n[1];
    const float okf = (ex < g.nel[0] && ey < g.nel[1]) ? 1.f : 0.f;       // elements beyond the mesh: clamped loads, zero weight
    const unsigned epl = (unsigned)(g.nel[0] * g.nel[1]), eps = epl * (unsigned)g.nel[2];
    const unsigned npl = (unsigned)(g.n[0] * g.n[1]);
    const float* src = gout + (size_t)b * g.G * eps;
    float* dst = gin + (size_t)b * npl * g.n[2];
    const unsigned eoff = (unsigned)min(ey, g.nel[1] - 1) * (unsigned)g.nel[0] + (unsigned)min(ex, g.nel[0] - 1);
    const unsigned noff = (unsigned)ey * (unsigned)g.n[0] + (unsigned)ex;
    const int ez_own = strip * R, ez_begin = ez_own > 0 ? ez_own - 1 : 0, ez_end = min(ez_own + R, g.nel[2]);
    __shared__ float xch[2][256];
    const int from_left = (int)(((unsigned)tid - 1u) & 63u) << 2;
    const float nfirst = tx > 0 ? 1.f : 0.f;
    int par = 0;
    float carry[4] = {0.f, 0.f, 0.f, 0.f};          // contributions of the plane below to this plane's nodes (i, j) = [j * 2 + i]

    auto emit = [&](const float (&o)[4], int z, bool owned_plane) {      // o: this element's share of its four nodes in node plane z
        const float left = gpe_from_left(o[1], from_left, nfirst);
        xch[par][tid] = o[2] + gpe_from_left(o[3], from_left, nfirst);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        float t = o[0] + left;
        if (ty > 0) t += xch[par][tid - 16];
        par ^= 1;
        if (owned_plane && owner && node_ok) dst[(size_t)z * npl + noff] = t;
    };

    float vn[8];
    if constexpr (G8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) vn[k] = src[(size_t)k * eps + (size_t)ez_begin * epl + eoff];
    }
    for (int ez = ez_begin; ez < ez_end; ++ez) {
        const float* pe = src + (size_t)ez * epl + eoff;
        float c[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if constexpr (G8) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = vn[k];
            const size_t nxt = (size_t)min(ez + 1, g.nel[2] - 1) * epl + eoff;          // beyond the strip: re-read the last plane (unused)
#pragma unroll
            for (int k = 0; k < 8; ++k) vn[k] = src[(size_t)k * eps + nxt];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float* tp = tables + k * 8;
#pragma unroll
                for (int a = 0; a < 8; ++a) c[a] = fmaf(tp[a], v[k], c[a]);
            }
        } else
        for (int g0 = 0; g0 < g.G; g0 += 4) {        // four Gauss-point planes in flight (they are megabytes apart)
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = pe[(size_t)min(g0 + k, g.G - 1) * eps];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool live = g0 + k < g.G;
                const float* tp = tables + min(g0 + k, g.G - 1) * 8;
#pragma unroll
                for (int a = 0; a < 8; ++a) c[a] = fmaf(live ? tp[a] : 0.f, v[k], c[a]);
            }
        }
        float o[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) { o[a] = fmaf(okf, c[a], carry[a]); carry[a] = okf * c[4 + a]; }
        emit(o, ez, ez >= ez_own);
    }
    if (ez_end == g.nel[2]) emit(carry, g.n[2] - 1, true);                // the last strip owns the top node plane
}

// Standard-layout assembly: all candidate loads first, then added in ascending local id (upper element = local 0 first),
// the order of the reference's sliced "+=" lines.
template <int NSD, int NB>
__global__ void __launch_bounds__(256) assemble_std_kernel(const float* __restrict__ rs, float* __restrict__ out, const GpeGeom g,
                                                           const int accumulate) {
    constexpr int NBT = NSD == 2 ? NB * NB : NB * NB * NB;
    constexpr int NC = 1 << NSD;
    const unsigned nel_s = (unsigned)(g.nel[0] * g.nel[1] * g.nel[2]);
    const int64_t nps = (int64_t)g.n[0] * g.n[1] * g.n[2];
    const int64_t total = nps * g.batch;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(idx / nps);
        const unsigned nd = (unsigned)(idx - (int64_t)b * nps);
        const int xyz[3] = {(int)(nd % (unsigned)g.n[0]), (int)((nd / (unsigned)g.n[0]) % (unsigned)g.n[1]),
                            (int)(nd / ((unsigned)g.n[0] * (unsigned)g.n[1]))};
        const NodeCand<NSD> c = node_candidates<NSD, NB>(xyz, g);
        const float* src = rs + (int64_t)b * NBT * nel_s;
        float v[NC];
        bool ok[NC];
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            // ascending local id: local index 0 (slot 1 when the node is shared... i.e. the upper element) before local S
            // (slot 0); per axis "first" = slot 1, "second" = slot 0; z is the slowest digit of the local id
            const int fx = q & 1, fy = (q >> 1) & 1, fz = NSD > 2 ? (q >> 2) & 1 : 0;
            const int cx = 1 - fx, cy = 1 - fy, cz = NSD > 2 ? 1 - fz : 1;
            const unsigned e = ((unsigned)c.e[2][cz] * (unsigned)g.nel[1] + (unsigned)c.e[1][cy]) * (unsigned)g.nel[0] + (unsigned)c.e[0][cx];
            const unsigned a = (unsigned)(NSD == 2 ? c.l[1][cy] * NB + c.l[0][cx] : (c.l[2][cz] * NB + c.l[1][cy]) * NB + c.l[0][cx]);
            ok[q] = c.ok[0][cx] && c.ok[1][cy] && (NSD > 2 ? c.ok[2][cz] : true);
            v[q] = src[(size_t)a * nel_s + e];
        }
        float s = accumulate ? out[idx] : 0.f;
#pragma unroll
        for (int q = 0; q < NC; ++q) s += ok[q] ? v[q] : 0.f;
        out[idx] = s;
    }
}

// out[b,node] (+)= sum_{(e,a) -> node} r_split[b,a,e]; contributions are added in ascending local id a,
// i.e. the order of the reference's sliced "+=" lines.
template <int NSD, int NB>
__global__ void __launch_bounds__(256) assemble_kernel(const float* __restrict__ rs, float* __restrict__ out, const GpeGeom g,
                                                       const int accumulate) {
    constexpr int NBT = NSD == 2 ? NB * NB : NB * NB * NB;
    const int64_t nel_s = (int64_t)g.nel[0] * g.nel[1] * g.nel[2];
    const int64_t nps = (int64_t)g.n[0] * g.n[1] * g.n[2];
    const int64_t total = nps * g.batch;
    const int S = g.stride;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(idx / nps);
        const int64_t nd = idx - (int64_t)b * nps;
        const int xyz[3] = {(int)(nd % g.n[0]), (int)((nd / g.n[0]) % g.n[1]), (int)(nd / ((int64_t)g.n[0] * g.n[1]))};
        int ce[3][4], cl[3][4], cn[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            cn[d] = 0;
            if (d >= NSD) { ce[d][0] = 0; cl[d][0] = 0; cn[d] = 1; continue; }
            const int ehi = min(xyz[d] / S, g.nel[d] - 1);
            for (int e = ehi; e >= 0 && xyz[d] - e * S <= NB - 1 && cn[d] < 4; --e) { ce[d][cn[d]] = e; cl[d][cn[d]] = xyz[d] - e * S; ++cn[d]; }
        }
        // candidates are stored with the local index ascending (upper element first => local id 0 first)
        float s = accumulate ? out[idx] : 0.f;
        const float* src = rs + (int64_t)b * NBT * nel_s;
        for (int kz = 0; kz < cn[2]; ++kz)
            for (int ky = 0; ky < cn[1]; ++ky)
                for (int kx = 0; kx < cn[0]; ++kx) {
                    const int64_t e = ((int64_t)ce[2][kz] * g.nel[1] + ce[1][ky]) * g.nel[0] + ce[0][kx];
                    const int a = NSD == 2 ? cl[1][ky] * NB + cl[0][kx] : (cl[2][kz] * NB + cl[1][ky]) * NB + cl[0][kx];
                    s += src[(int64_t)a * nel_s + e];
                }
        out[idx] = s;
    }
}

// adjoint of assemble: grad_split[b,a,e] = grad_out[b,node(e,a)]
template <int NSD, int NB>
__global__ void __launch_bounds__(256) assemble_bwd_kernel(const float* __restrict__ gout, float* __restrict__ gs, const GpeGeom g) {
    constexpr int NBT = NSD == 2 ? NB * NB : NB * NB * NB;
    const int64_t nel_s = (int64_t)g.nel[0] * g.nel[1] * g.nel[2];
    const int64_t nps = (int64_t)g.n[0] * g.n[1] * g.n[2];
    const int64_t total = nel_s * g.batch;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(idx / nel_s);
        const int64_t e = idx - (int64_t)b * nel_s;
        const int ex = (int)(e % g.nel[0]);
        const int ey = (int)((e / g.nel[0]) % g.nel[1]);
        const int ez = (int)(e / ((int64_t)g.nel[0] * g.nel[1]));
        const float* src = gout + (int64_t)b * nps + ((int64_t)ez * g.stride * g.n[1] + (int64_t)ey * g.stride) * g.n[0] + (int64_t)ex * g.stride;
        float* dst = gs + (int64_t)b * NBT * nel_s + e;
#pragma unroll
        for (int a = 0; a < NBT; ++a) {
            const int ib = a % NB, jb = (a / NB) % NB, kb = a / (NB * NB);
            dst[(int64_t)a * nel_s] = src[((int64_t)kb * g.n[1] + jb) * g.n[0] + ib];
        }
    }
}

static int make_geom(GpeGeom& g, int batch, int nsd, const int32_t n[3], int nbf, int stride, int G) {
    if (!n || batch < 1 || nsd < 1 || nsd > 3 || nbf < 2 || nbf > 4 || stride < 1 || G < 1) return DN_E_BADARG;
    g.batch = batch; g.nsd = nsd; g.nbf = nbf; g.stride = stride; g.G = G;
    for (int d = 0; d < 3; ++d) {
        g.n[d] = d < nsd ? n[d] : 1;
        if (g.n[d] < 1) return DN_E_BADARG;
        g.nel[d] = d < nsd ? (g.n[d] - nbf) / stride + 1 : 1;   // conv output length
        if (d < nsd && g.n[d] < nbf) return DN_E_BADARG;
    }
    return 0;
}

static int grid_for(int64_t total) {
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;   // grid-stride beyond 16 workgroups per CU
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

#define DN_DISPATCH_NSD_NB(KERNEL, nsd, nbf, ...)                                                   \
    switch ((nsd) * 10 + (nbf)) {                                                                   \
        case 12: KERNEL(1, 2, __VA_ARGS__); break;                                                  \
        case 13: KERNEL(1, 3, __VA_ARGS__); break;                                                  \
        case 14: KERNEL(1, 4, __VA_ARGS__); break;                                                  \
        case 22: KERNEL(2, 2, __VA_ARGS__); break;                                                  \
        case 23: KERNEL(2, 3, __VA_ARGS__); break;                                                  \
        case 24: KERNEL(2, 4, __VA_ARGS__); break;                                                  \
        case 32: KERNEL(3, 2, __VA_ARGS__); break;                                                  \
        case 33: KERNEL(3, 3, __VA_ARGS__); break;                                                  \
        case 34: KERNEL(3, 4, __VA_ARGS__); break;                                                  \
        default: return DN_E_UNSUPPORTED;                                                           \
    }

}  // namespace dn

using namespace dn;

extern "C" int dn_gauss_pt_eval_fwd(const float* in, const float* tables, float* out, int32_t batch, int32_t nsd,
                                    const int32_t n[3], int32_t nbf, int32_t stride, int32_t G, void* stream) {
    GpeGeom g;
    int rc = make_geom(g, batch, nsd, n, nbf, stride, G);
    if (rc) return rc;
    if (!in || !tables || !out) return DN_E_BADARG;
    int nbt = 1;
    for (int d = 0; d < nsd; ++d) nbt *= nbf;
    const size_t lds = sizeof(float) * (size_t)G * nbt;
    if (lds > 64 * 1024) return DN_E_UNSUPPORTED;
    const int64_t total = (int64_t)g.nel[0] * g.nel[1] * g.nel[2] * batch;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define K_FWD(NSD, NB, ...) hipLaunchKernelGGL((gpe_fwd_kernel<NSD, NB>), dim3(grid_for(total)), dim3(256), lds, s, in, tables, out, g)
    DN_DISPATCH_NSD_NB(K_FWD, nsd, nbf, 0)
#undef K_FWD
    DN_LAUNCH_CHECK();
    return 0;
}

extern "C" int dn_gauss_pt_eval_bwd(const float* grad_out, const float* tables, float* grad_in, int32_t batch, int32_t nsd,
                                    const int32_t n[3], int32_t nbf, int32_t stride, int32_t G, void* stream) {
    GpeGeom g;
    int rc = make_geom(g, batch, nsd, n, nbf, stride, G);
    if (rc) return rc;
    if (!grad_out || !tables || !grad_in) return DN_E_BADARG;
    int nbt = 1;
    for (int d = 0; d < nsd; ++d) nbt *= nbf;
    const size_t lds = sizeof(float) * (size_t)G * nbt;
    if (lds > 64 * 1024) return DN_E_UNSUPPORTED;
    const int64_t total = (int64_t)g.n[0] * g.n[1] * g.n[2] * batch;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const size_t lds_t = sizeof(float) * (size_t)nbt * ((G + 3) & ~3);
    if (stride == nbf - 1 && lds_t <= 64 * 1024 && (int64_t)g.nel[0] * g.nel[1] * g.nel[2] * G < (1ll << 31) &&
        (int64_t)g.n[0] * g.n[1] * g.n[2] < (1ll << 31)) {
    if (nsd == 3 && nbf == 2 && g.n[0] >= 2 && config(CFG_GPE_GATHER) == nullptr && config(CFG_GPE_TILED) == nullptr) {
        // 3-D Q1: marching adjoint (every value read once).  Strip height: >= 4 workgroups per CU where the mesh allows it
        // one thread column / row per NODE column / row (the closing one has no element of its own), tiles overlap by one thread
        const int chunks = g.n[0] <= 16 ? 1 : (g.n[0] - 1 + 14) / 15, tiles = g.n[1] <= 16 ? 1 : (g.n[1] - 1 + 14) / 15;
        int R = 32;
        while (R > 8 && (int64_t)chunks * tiles * batch * ((g.nel[2] + R - 1) / R) < 2560) R /= 2;
        if (R > g.nel[2]) R = g.nel[2];
        const int strips = (g.nel[2] + R - 1) / R;
        const int64_t nblk = (int64_t)chunks * tiles * strips * batch;
        if (nblk < (1ll << 31)) {
            if (G == 8) hipLaunchKernelGGL(gpe_bwd_march3d_q1_kernel<true>, dim3((unsigned)nblk), dim3(16, 16), 0, s, grad_out, tables, grad_in, g, chunks, tiles, strips, R);
            else hipLaunchKernelGGL(gpe_bwd_march3d_q1_kernel<false>, dim3((unsigned)nblk), dim3(16, 16), 0, s, grad_out, tables, grad_in, g, chunks, tiles, strips, R);
            DN_LAUNCH_CHECK();
            return 0;
        }
    }
    {   // tiled element-centric adjoint where its LDS tile fits (all 2-D cases, 3-D Q1 / Q2)
        const int S = nbf - 1;
        const size_t tile_floats = (size_t)nbt * (nsd == 2 ? 1024 : 2048);       // contrib[NB^nsd][tile elements]
        (void)S;
        if (nsd >= 2 && tile_floats * sizeof(float) <= 96 * 1024 && config(CFG_GPE_GATHER) == nullptr) {
            const int tx_ = g.nel[0] <= 32 ? 1 : (g.nel[0] - 1 + 30) / 31;
            const int ty_ = nsd == 2 ? (g.nel[1] <= 32 ? 1 : (g.nel[1] - 1 + 30) / 31) : (g.nel[1] <= 8 ? 1 : (g.nel[1] - 1 + 6) / 7);
            const int tz_ = nsd == 2 ? 1 : (g.nel[2] <= 8 ? 1 : (g.nel[2] - 1 + 6) / 7);
            const int64_t nblk = (int64_t)tx_ * ty_ * tz_ * batch;
            if (nblk < (1ll << 31)) {
#define K_BWDT(NSD, NB) hipLaunchKernelGGL((gpe_bwd_tiled_kernel<NSD, NB>), dim3((unsigned)nblk), dim3(256), tile_floats * sizeof(float), s, grad_out, tables, grad_in, g, tx_, ty_, tz_)
                switch (nsd * 10 + nbf) {
                    case 22: K_BWDT(2, 2); break;
                    case 23: K_BWDT(2, 3); break;
                    case 24: K_BWDT(2, 4); break;
                    case 32: K_BWDT(3, 2); break;
                    default: K_BWDT(3, 3); break;
                }
#undef K_BWDT
                DN_LAUNCH_CHECK();
                return 0;
            }
        }
    }
    if (nsd == 3 && nbf >= 3 && config(CFG_GPE_GATHER) == nullptr) {       // 3-D Q2 / Q3: small tiles
        const int tz_el = nbf >= 4 ? 4 : 8;
        const int tx_ = g.nel[0] <= 8 ? 1 : (g.nel[0] - 1 + 6) / 7, ty_ = g.nel[1] <= 8 ? 1 : (g.nel[1] - 1 + 6) / 7;
        const int tz_ = g.nel[2] <= tz_el ? 1 : (g.nel[2] - 1 + tz_el - 2) / (tz_el - 1);
        const int64_t nblk = (int64_t)tx_ * ty_ * tz_ * batch;
        const size_t ldsb = sizeof(float) * (size_t)nbt * 64 * tz_el;
        // (Q3 on small meshes stays on the per-node gather: 21^3 elements are 63 tiles of 64 planes each -- 110 us against 49, gpurun_out/t11_ops.txt)
        if (nblk < (1ll << 31) && (nbf == 3 || nblk >= 512)) {
            if (nbf == 3) hipLaunchKernelGGL((gpe_bwd_tiled_kernel<3, 3, true>), dim3((unsigned)nblk), dim3(256), ldsb, s, grad_out, tables, grad_in, g, tx_, ty_, tz_);
            else hipLaunchKernelGGL((gpe_bwd_tiled_kernel<3, 4, true>), dim3((unsigned)nblk), dim3(256), ldsb, s, grad_out, tables, grad_in, g, tx_, ty_, tz_);
            DN_LAUNCH_CHECK();
            return 0;
        }
    }
#define K_BWDS(NSD, NB, ...)                                                                                                                  \
    do {                                                                                                                                      \
        if (NSD == 3 && NB >= 3)                                                                                                              \
            hipLaunchKernelGGL((gpe_bwd_std_scalar_kernel<NSD, NB>), dim3(grid_for(total)), dim3(256), lds, s, grad_out, tables, grad_in, g); \
        else                                                                                                                                  \
            hipLaunchKernelGGL((gpe_bwd_std_kernel<NSD, NB>), dim3(grid_for(total)), dim3(256), lds_t, s, grad_out, tables, grad_in, g);      \
    } while (0)
        DN_DISPATCH_NSD_NB(K_BWDS, nsd, nbf, 0)
#undef K_BWDS
    } else {
#define K_BWD(NSD, NB, ...) hipLaunchKernelGGL((gpe_bwd_kernel<NSD, NB>), dim3(grid_for(total)), dim3(256), lds, s, grad_out, tables, grad_in, g)
        DN_DISPATCH_NSD_NB(K_BWD, nsd, nbf, 0)
#undef K_BWD
    }
    DN_LAUNCH_CHECK();
    return 0;
}

extern "C" int dn_assemble(const float* r_split, float* out, int32_t batch, int32_t nsd, const int32_t n[3], int32_t nbf,
                           int32_t stride, int32_t accumulate, void* stream) {
    GpeGeom g;
    int rc = make_geom(g, batch, nsd, n, nbf, stride, 1);
    if (rc) return rc;
    if (!r_split || !out || nsd < 2 || stride != nbf - 1) return DN_E_BADARG;
    const int64_t total = (int64_t)g.n[0] * g.n[1] * g.n[2] * batch;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool small = (int64_t)g.nel[0] * g.nel[1] * g.nel[2] * nbf * nbf * (nsd == 3 ? nbf : 1) < (1ll << 31) &&
                       (int64_t)g.n[0] * g.n[1] * g.n[2] < (1ll << 31);
#define K_ASM(NSD, NB, ...)                                                                                                               \
    do {                                                                                                                                  \
        if (small) hipLaunchKernelGGL((assemble_std_kernel<NSD, NB>), dim3(grid_for(total)), dim3(256), 0, s, r_split, out, g, accumulate); \
        else hipLaunchKernelGGL((assemble_kernel<NSD, NB>), dim3(grid_for(total)), dim3(256), 0, s, r_split, out, g, accumulate);          \
    } while (0)
    switch (nsd * 10 + nbf) {
        case 22: K_ASM(2, 2); break;
        case 23: K_ASM(2, 3); break;
        case 24: K_ASM(2, 4); break;
        case 32: K_ASM(3, 2); break;
        case 33: K_ASM(3, 3); break;
        case 34: K_ASM(3, 4); break;
        default: return DN_E_UNSUPPORTED;
    }
#undef K_ASM
    DN_LAUNCH_CHECK();
    return 0;
}

extern "C" int dn_assemble_bwd(const float* grad_out, float* grad_split, int32_t batch, int32_t nsd, const int32_t n[3],
                               int32_t nbf, int32_t stride, void* stream) {
    GpeGeom g;
    int rc = make_geom(g, batch, nsd, n, nbf, stride, 1);
    if (rc) return rc;
    if (!grad_out || !grad_split || nsd < 2 || stride != nbf - 1) return DN_E_BADARG;
    const int64_t total = (int64_t)g.nel[0] * g.nel[1] * g.nel[2] * batch;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define K_ASMB(NSD, NB, ...) hipLaunchKernelGGL((assemble_bwd_kernel<NSD, NB>), dim3(grid_for(total)), dim3(256), 0, s, grad_out, grad_split, g)
    switch (nsd * 10 + nbf) {
        case 22: K_ASMB(2, 2); break;
        case 23: K_ASMB(2, 3); break;
        case 24: K_ASMB(2, 4); break;
        case 32: K_ASMB(3, 2); break;
        case 33: K_ASMB(3, 3); break;
        case 34: K_ASMB(3, 4); break;
        default: return DN_E_UNSUPPORTED;
    }
#undef K_ASMB
    DN_LAUNCH_CHECK();
    return 0;
}
