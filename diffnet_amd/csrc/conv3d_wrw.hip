// Weight gradient of the generator's 4x4x4, stride-2, padding-1 3-D convolutions and transposed convolutions
// (reference DiffNet/networks/wgan3d.py:23-55: UNetDown `Conv3d(cin, cout, 4, 2, 1)`, UNetUp `ConvTranspose3d(cin, cout, 4, 2, 1)`).
// Both are the same reduction over the coarse grid:
//     gW[m][cn][kz][ky][kx] = sum_{b, (i,j,k)} coarse[b][m][i][j][k] * fine[b][cn][2i+kz-1][2j+ky-1][2k+kx-1]      (zero outside)
//   Conv3d:          coarse = grad_out (m = cout), fine = input    (cn = cin),  weight (cout, cin, 4,4,4)
//   ConvTranspose3d: coarse = input    (m = cin),  fine = grad_out (cn = cout), weight (cin, cout, 4,4,4)
// MIOpen runs these through a batched-GEMM / naive fallback that takes 35 ms for the first layer (1 -> 16 on 128^3) and 5 ms for
// the 16 -> 32 and 64 -> 16 layers (profiles/r1_gen3d_step.txt).  Here: a workgroup owns one fine channel cn and a share of the
// coarse positions; per tile of 64 positions it stages the 64-tap patches S[pos][tap] and the coarse values V[m][pos] in
// LDS and accumulates the (M x 64) block  V * S  in registers; per-workgroup partials are summed in index order.
#include "dn_common.h"

namespace dn {

constexpr int WR_TP = 64;        // coarse positions per tile
constexpr int WR_FEW = 16;       // up to this many workgroups per fine channel: thread-per-output final sum

template <int MR>                // MR = ceil(M / 16): rows per thread
__global__ void __launch_bounds__(256) conv3d_wrw_kernel(const float* __restrict__ fine, const float* __restrict__ coarse,
                                                         float* __restrict__ part, int B, int CN, int M, int d, int h, int w,
                                                         int tiles_per_wg) {
    __shared__ __attribute__((aligned(16))) float S[WR_TP][68];
    __shared__ float V[MR * 16][WR_TP + 1];
    const int tid = threadIdx.x;
    const int cn = blockIdx.y;
    const int D2 = 2 * d, H2 = 2 * h, W2 = 2 * w;
    const size_t cvol = (size_t)d * h * w, fvol = (size_t)D2 * H2 * W2;
    const long npos = (long)B * cvol;
    const int pos = tid & (WR_TP - 1), quarter = tid >> 6;        // staging roles
    const int t4 = (tid & 15) * 4, m0 = tid >> 4;                 // accumulation roles
    float acc[MR][4];
#pragma unroll
    for (int r = 0; r < MR; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[r][q] = 0.f;
    for (int t = 0; t < tiles_per_wg; ++t) {
        const long p = ((long)blockIdx.x * tiles_per_wg + t) * WR_TP + pos;
        const bool ok = p < npos;
        int b = 0, i = 0, j = 0, k = 0;
        if (ok) {
            b = (int)(p / (long)cvol);
            int r = (int)(p % (long)cvol);
            i = r / (h * w); r %= h * w; j = r / w; k = r % w;
        }
        __syncthreads();
        {   // patch plane kz = quarter of this thread's position: 4 x 4 values
            const int z = 2 * i + quarter - 1;
            const bool zok = ok && z >= 0 && z < D2;
            const float* fb = fine + ((size_t)b * CN + cn) * fvol + (size_t)(zok ? z : 0) * H2 * W2;
#pragma unroll
            for (int ky = 0; ky < 4; ++ky) {
                const int y = 2 * j + ky - 1;
                const bool yok = zok && y >= 0 && y < H2;
                const float* fr = fb + (size_t)(yok ? y : 0) * W2;
                float v[4];
#pragma unroll
                for (int kx = 0; kx < 4; ++kx) {
                    const int x = 2 * k + kx - 1;
                    v[kx] = (yok && x >= 0 && x < W2) ? fr[x] : 0.f;
                }
                *reinterpret_cast<float4*>(&S[pos][quarter * 16 + ky * 4]) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
        {
            const float* cb = coarse + (size_t)b * M * cvol + ((size_t)i * h + j) * w + k;
#pragma unroll
            for (int r = 0; r < MR * 4; ++r) {
                const int m = quarter + 4 * r;
                V[m][pos] = (ok && m < M) ? cb[(size_t)m * cvol] : 0.f;
            }
        }
        __syncthreads();
#pragma unroll 4
        for (int q = 0; q < WR_TP; ++q) {
            const float4 sv = *reinterpret_cast<const float4*>(&S[q][t4]);
#pragma unroll
            for (int r = 0; r < MR; ++r) {
                const float v = V[m0 + 16 * r][q];
                acc[r][0] = fmaf(v, sv.x, acc[r][0]);
                acc[r][1] = fmaf(v, sv.y, acc[r][1]);
                acc[r][2] = fmaf(v, sv.z, acc[r][2]);
                acc[r][3] = fmaf(v, sv.w, acc[r][3]);
            }
        }
    }
    // partial layout: few workgroups -> [workgroup][output] (summed by one thread per output, coalesced; with a single
    // workgroup `part` IS grad_weight); many -> [output][workgroup] (summed by one wave per output, coalesced)
    const size_t nwg = gridDim.x, nout = (size_t)M * CN * 64;
    const bool wg_major = nwg <= WR_FEW;
#pragma unroll
    for (int r = 0; r < MR; ++r) {
        const int m = m0 + 16 * r;
        if (m < M) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const size_t o = (size_t)(m * CN + cn) * 64 + t4 + q;
                part[wg_major ? blockIdx.x * nout + o : o * nwg + blockIdx.x] = acc[r][q];
            }
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
#define DN_WR(R) hipLaunchKernelGGL((conv3d_wrw_kernel<R>), grid, block, 0, s, fine, coarse, part, (int)B, (int)CN, (int)M, (int)d, (int)h, (int)w, tpw)
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
