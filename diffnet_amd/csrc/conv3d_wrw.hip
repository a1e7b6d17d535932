// Weight gradient of the generator's 4x4x4, stride-2, padding-1 3-D convolutions and transposed convolutions
// (reference DiffNet/networks/wgan3d.py:23-55: UNetDown `Conv3d(cin, cout, 4, 2, 1)`, UNetUp `ConvTranspose3d(cin, cout, 4, 2, 1)`).
// Both are the same reduction over the coarse grid:
//     gW[m][cn][kz][ky][kx] = sum_{b, (i,j,k)} coarse[b][m][i][j][k] * fine[b][cn][2i+kz-1][2j+ky-1][2k+kx-1]      (zero outside)
//   Conv3d:          coarse = grad_out (m = cout), fine = input    (cn = cin),  weight (cout, cin, 4,4,4)
//   ConvTranspose3d: coarse = input    (m = cin),  fine = grad_out (cn = cout), weight (cin, cout, 4,4,4)
// MIOpen runs these through a batched-GEMM / naive fallback that takes 35 ms for the first layer (1 -> 16 on 128^3) and 5 ms for
// the 16 -> 32 and 64 -> 16 layers (profiles/r1_gen3d_step.txt).  Here: a workgroup owns one fine channel cn and a share of the
// coarse positions; per tile of 64 positions it stages the 64-tap patches S[pos][tap] and the coarse values V[m][pos] in
// LDS and accumulates the (M x 64) block  V * S  on the matrix cores (v_mfma_f32_16x16x4_f32: exact fp32, one k-ordered fma
// chain per output -- cdna_hip_programming.md "FP32-input MFMA"; wave w owns tap columns 16w..16w+15, MR row tiles of 16
// coarse channels); per-workgroup partials are summed in index order.
#include "dn_common.h"

namespace dn {

constexpr int WR_TP = 64;        // coarse positions per tile
constexpr int WR_FEW = 16;       // up to this many workgroups per fine channel: thread-per-output final sum

typedef float wr_f32x4 __attribute__((ext_vector_type(4)));

// MR = ceil(M / 16) row tiles of 16 coarse channels.  Software pipeline over the position tiles: the global loads of tile
// t + 1 are issued before the MFMAs of tile t and written to the other LDS buffer afterwards -- one barrier per tile, the
// load latency hidden behind the matrix work.
template <int MR>
__global__ void __launch_bounds__(256) conv3d_wrw_kernel(const float* __restrict__ fine, const float* __restrict__ coarse,
                                                         float* __restrict__ part, int B, int CN, int M, int d, int h, int w,
                                                         int tiles_per_wg) {
    // row strides 80 / 68 floats: the MFMA operand reads (16 lanes along a row, 4 lane groups along k) hit 64 distinct banks
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float (*S)[WR_TP][80] = reinterpret_cast<float (*)[WR_TP][80]>(lds);                                 // [2][64][80]
    float (*V)[MR * 16][68] = reinterpret_cast<float (*)[MR * 16][68]>(lds + 2 * WR_TP * 80);            // [2][16 MR][68]
    const int tid = threadIdx.x;
    // (round 4, measured and not kept: a 1-D grid with the fine channel fastest and an XCD-aware decode, so that the CN workgroups of one share of the positions
    // read the coarse values into one L2 instead of each re-reading them -- GoodGenerator 128^3 step 2.00 -> 3.98 ms, 256^3 8.7 -> 9.4: sixteen fine channels
    // per XCD at once evict each other's patches, and the coarse re-reads were coming from the 256 MB cache anyway)
    const int cn = blockIdx.y;
    const int D2 = 2 * d, H2 = 2 * h, W2 = 2 * w;
    const size_t cvol = (size_t)d * h * w, fvol = (size_t)D2 * H2 * W2;
    const long npos = (long)B * cvol;
    const int pos = tid & (WR_TP - 1), quarter = tid >> 6;                      // staging roles
    const int lane = tid & 63, wv = tid >> 6, li = lane & 15, lk = lane >> 4;   // MFMA operand roles: A[row li][k lk], B[k lk][col li]
    wr_f32x4 acc[MR];
#pragma unroll
    for (int r = 0; r < MR; ++r) acc[r] = (wr_f32x4){0.f, 0.f, 0.f, 0.f};
    float sreg[16], vreg[MR * 4];

    auto issue = [&](int t) {             // global loads of tile t: one 4 x 4 patch plane (kz = quarter) + MR * 4 coarse values
        const long p = ((long)blockIdx.x * tiles_per_wg + t) * WR_TP + pos;
        const bool ok = p < npos;
        int b = 0, i = 0, j = 0, k = 0;
        if (ok) {
            b = (int)(p / (long)cvol);
            int r = (int)(p % (long)cvol);
            i = r / (h * w); r %= h * w; j = r / w; k = r % w;
        }
        const int z = 2 * i + quarter - 1;
        const bool zok = ok && z >= 0 && z < D2;
        const float* fb = fine + ((size_t)b * CN + cn) * fvol + (size_t)(zok ? z : 0) * H2 * W2;
#pragma unroll
        for (int ky = 0; ky < 4; ++ky) {
            const int y = 2 * j + ky - 1;
            const bool yok = zok && y >= 0 && y < H2;
            const float* fr = fb + (size_t)(yok ? y : 0) * W2;
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) {
                const int x = 2 * k + kx - 1;
                sreg[ky * 4 + kx] = (yok && x >= 0 && x < W2) ? fr[x] : 0.f;
            }
        }
        const float* cb = coarse + (size_t)b * M * cvol + ((size_t)i * h + j) * w + k;
#pragma unroll
        for (int r = 0; r < MR * 4; ++r) {
            const int m = quarter + 4 * r;
            vreg[r] = (ok && m < M) ? cb[(size_t)m * cvol] : 0.f;
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int ky = 0; ky < 4; ++ky)
            *reinterpret_cast<float4*>(&S[buf][pos][quarter * 16 + ky * 4]) =
                make_float4(sreg[ky * 4], sreg[ky * 4 + 1], sreg[ky * 4 + 2], sreg[ky * 4 + 3]);
#pragma unroll
        for (int r = 0; r < MR * 4; ++r) V[buf][quarter + 4 * r][pos] = vreg[r];
    };

    issue(0);
    commit(0);
    __syncthreads();
    for (int t = 0; t < tiles_per_wg; ++t) {
        const int buf = t & 1;
        const bool more = t + 1 < tiles_per_wg;
        if (more) issue(t + 1);
#pragma unroll 4
        for (int q0 = 0; q0 < WR_TP; q0 += 4) {
            const float bq = S[buf][q0 + lk][16 * wv + li];
#pragma unroll
            for (int r = 0; r < MR; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[buf][16 * r + li][q0 + lk], bq, acc[r], 0, 0, 0);
        }
        if (more) commit(buf ^ 1);        // its last readers finished before the barrier that ended the previous trip
        __syncthreads();
    }
    // partial layout: few workgroups -> [workgroup][output] (summed by one thread per output, coalesced; with a single
    // workgroup `part` IS grad_weight); many -> [output][workgroup] (summed by one wave per output, coalesced).
    // C/D layout of the 16x16 tile: column = lane & 15, row = 4 (lane >> 4) + register
    const size_t nwg = gridDim.x, nout = (size_t)M * CN * 64;
    const bool wg_major = nwg <= WR_FEW;
#pragma unroll
    for (int r = 0; r < MR; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = 16 * r + 4 * lk + q;
            if (m < M) {
                const size_t o = (size_t)(m * CN + cn) * 64 + 16 * wv + li;
                part[wg_major ? blockIdx.x * nout + o : o * nwg + blockIdx.x] = acc[r][q];
            }
        }
}

// few workgroups: one thread per output, partials [workgroup][output]
__global__ void __launch_bounds__(256) conv3d_wsum_few_kernel(const float* __restrict__ part, float* __restrict__ gw, int nwg, long n) {
    const long k = (long)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    double s = 0.0;
    for (int g = 0; g < nwg; ++g) s += (double)part[(size_t)g * n + k];
    gw[k] = (float)s;
}

// one wave per output: coalesced fp64 sum of the per-workgroup partials
__global__ void __launch_bounds__(256) conv3d_wsum_kernel(const float* __restrict__ part, float* __restrict__ gw, int nwg, long n) {
    const long k = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (k >= n) return;
    const float* pk = part + (size_t)k * nwg;
    double s = 0.0;
    for (int g = lane; g < nwg; g += 64) s += (double)pk[g];
    s = wave_sum(s);
    if (lane == 0) gw[k] = (float)s;
}

static void wr_plan(int64_t npos, int64_t CN, int& nwg, int& tiles_per_wg) {
    const int64_t tiles = (npos + WR_TP - 1) / WR_TP;
    int64_t want = (4096 + CN - 1) / CN;       // ~16 workgroups per CU over all fine channels
    if (want < 1) want = 1;
    if (want > tiles) want = tiles;
    tiles_per_wg = (int)((tiles + want - 1) / want);
    nwg = (int)((tiles + tiles_per_wg - 1) / tiles_per_wg);
}

}  // namespace dn

using namespace dn;

static int wr_check(int64_t B, int64_t CN, int64_t M, int64_t d, int64_t h, int64_t w) {
    if (B < 1 || CN < 1 || M < 1 || d < 1 || h < 1 || w < 1) return DN_E_BADARG;
    if (M > 128 || CN > 65535 || 8 * d * h * w >= (1ll << 31) || B * d * h * w >= (1ll << 40)) return DN_E_UNSUPPORTED;
    return 0;
}

extern "C" int64_t dn_conv3d_k4s2_wrw_workspace_bytes(int64_t B, int64_t CN, int64_t M, int64_t d, int64_t h, int64_t w) {
    if (int rc = wr_check(B, CN, M, d, h, w)) return rc;
    int nwg, tpw;
    wr_plan(B * d * h * w, CN, nwg, tpw);
    return nwg == 1 ? 0 : (int64_t)sizeof(float) * M * CN * 64 * nwg;
}

extern "C" int dn_conv3d_k4s2_wrw(const float* fine, const float* coarse, float* grad_weight, int64_t B, int64_t CN, int64_t M, int64_t d,
                                  int64_t h, int64_t w, void* workspace, int64_t workspace_bytes, void* stream) {
    if (int rc = wr_check(B, CN, M, d, h, w)) return rc;
    if (!fine || !coarse || !grad_weight) return DN_E_BADARG;
    int nwg, tpw;
    wr_plan(B * d * h * w, CN, nwg, tpw);
    if (nwg > 1 && (!workspace || workspace_bytes < dn_conv3d_k4s2_wrw_workspace_bytes(B, CN, M, d, h, w))) return DN_E_WORKSPACE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float* part = nwg == 1 ? grad_weight : static_cast<float*>(workspace);
    const dim3 grid((unsigned)nwg, (unsigned)CN), block(256);
    const int MR = (int)((M + 15) / 16);
#define DN_WR(R)                                                                                                                   \
    do {                                                                                                                           \
        const size_t lds_bytes = sizeof(float) * 2 * (WR_TP * 80 + (R) * 16 * 68);                                                 \
        static bool raised = false;   /* above 64 KB of dynamic LDS the limit has to be raised once per kernel */                  \
        if (!raised && lds_bytes > 64 * 1024) {                                                                                    \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_wrw_kernel<R>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    (int)lds_bytes) != hipSuccess)                                                                 \
                return DN_E_UNSUPPORTED;                                                                                           \
            raised = true;                                                                                                         \
        }                                                                                                                          \
        hipLaunchKernelGGL((conv3d_wrw_kernel<R>), grid, block, lds_bytes, s, fine, coarse, part, (int)B, (int)CN, (int)M, (int)d, (int)h, \
                           (int)w, tpw);                                                                                           \
    } while (0)
    switch (MR) {
        case 1: DN_WR(1); break;
        case 2: DN_WR(2); break;
        case 3: case 4: DN_WR(4); break;
        default: DN_WR(8); break;
    }
#undef DN_WR
    const long n = (long)M * CN * 64;
    if (nwg > WR_FEW) hipLaunchKernelGGL(conv3d_wsum_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, part, grad_weight, nwg, n);
    else if (nwg > 1) hipLaunchKernelGGL(conv3d_wsum_few_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, part, grad_weight, nwg, n);
    DN_LAUNCH_CHECK();
    return 0;
}
