// 4 x 4 x 4, stride-2, padding-1 convolution family of the 3-D generator on the fp32 matrix cores (v_mfma_f32_16x16x4_f32), NCDHW:
// the `down` and `up` contractions (forward / input gradient of both layers); the weight gradient is conv3d_wrw.hip.
//
// Reference layers: DiffNet/networks/wgan3d.py:23-55 (`UNetDown`: Conv3d(cin, cout, 4, 2, 1, bias=False); `UNetUp`:
// ConvTranspose3d(cin, cout, 4, 2, 1, bias=False)).  As in conv2d_k4s2.hip, with w[m][c][kz][ky][kx] (m = coarse channel, c = fine
// channel) being Conv3d's (cout, cin, 4,4,4) weight and ConvTranspose3d's (cin, cout, 4,4,4) weight as they stand:
//   down : coarse[b,m,i,j,k] = sum_{c,taps} w[m,c,kz,ky,kx] * fine[b,c,2i+kz-1,2j+ky-1,2k+kx-1]             Conv3d forward, ConvT3d input gradient
//   up   : fine[b,c,z,y,x]   = sum_{m,taps} w[m,c,kz,ky,kx] * coarse[b,m,(z+1-kz)/2,(y+1-ky)/2,(x+1-kx)/2]  ConvT3d forward, Conv3d input gradient
// Implicit GEMMs, 64 x 64 tiles per 256-thread workgroup (2 x 2 waves of 32 x 32), one fine channel (64 taps) per K-step of `down`,
// two coarse channels per K-step of `up`; `up` runs one GEMM per output parity with K = (m, 2 x 2 x 2 taps), the z parity split
// over blockIdx.z (halves the accumulators and the staged neighbourhood), both x parities of a position stored as one float2.
#include <algorithm>

#include "dn_common.h"

namespace dn {

typedef float c3_f32x4 __attribute__((ext_vector_type(4)));
constexpr int C3_TN = 64, C3_SA = 66, C3_SB = 80;

__device__ __forceinline__ c3_f32x4 mfma4_3(float a, float b, c3_f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// ---- down ---------------------------------------------------------------------------------------------------------------------
template <int TM>
__global__ void __launch_bounds__(256) conv3d_k4s2_down_kernel(const float* __restrict__ fine, const float* __restrict__ w,
                                                               float* __restrict__ coarse, int B, int C, int M, int D, int H, int W, int cs,
                                                               size_t slice_stride) {
    constexpr int RT = TM / 32;
    // split K (round 4): blockIdx.z = slice of `cs` fine channels, its partial result goes to coarse + slice * slice_stride (conv3d_ksum_kernel adds the
    // slices in order); one slice = the whole contraction written to `coarse` itself
    const int c_lo = (int)blockIdx.z * cs, c_hi = min(C, c_lo + cs);
    coarse += (size_t)blockIdx.z * slice_stride;
    __shared__ __attribute__((aligned(16))) float As[2][TM][C3_SA];
    __shared__ __attribute__((aligned(16))) float Bs[2][64][C3_SB];
    const int tid = threadIdx.x;
    const int HW = H * W, vol = D * HW, tiles_per_sample = (vol + C3_TN - 1) / C3_TN;
    const int b = blockIdx.x / tiles_per_sample, p0 = (blockIdx.x % tiles_per_sample) * C3_TN;
    const int m0 = blockIdx.y * TM;
    const int D2 = 2 * D, H2 = 2 * H, W2 = 2 * W;
    const size_t fvol = (size_t)D2 * H2 * W2;
    const int n = tid & 63, q = tid >> 6;                 // staging: position p0 + n, tap plane kz = q
    const int p = p0 + n;
    const bool pok = p < vol;
    const int pi = pok ? p / HW : 0, pj = pok ? (p % HW) / W : 0, pk = pok ? p % W : 0;
    const float* fb = fine + (size_t)b * C * fvol;
    const int lane = tid & 63, wv = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int wm = wv >> 1, wn = wv & 1;
    const int row0 = wm * (TM / 2), col0 = wn * 32;
    c3_f32x4 acc[RT][2];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int s = 0; s < 2; ++s) acc[r][s] = (c3_f32x4){0.f, 0.f, 0.f, 0.f};
    float breg[16];
    float4 areg[TM / 16];
    const int z = 2 * pi + q - 1;
    const bool zok = pok && z >= 0 && z < D2;
    auto issue = [&](int c) {
        const float* fc = fb + (size_t)c * fvol + (size_t)(zok ? z : 0) * H2 * W2;
#pragma unroll
        for (int ky = 0; ky < 4; ++ky) {
            const int y = 2 * pj + ky - 1;
            const bool yok = zok && y >= 0 && y < H2;
            const float* fr = fc + (size_t)(yok ? y : 0) * W2;
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) {
                const int x = 2 * pk + kx - 1;
                breg[ky * 4 + kx] = (yok && x >= 0 && x < W2) ? fr[x] : 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < TM / 16; ++r) {               // A tile: w[m0 + mm][c][64 taps]
            const int idx4 = tid + 256 * r, mm = idx4 >> 4, kk = (idx4 & 15) * 4;
            areg[r] = (m0 + mm < M) ? *reinterpret_cast<const float4*>(w + ((size_t)(m0 + mm) * C + c) * 64 + kk) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int t = 0; t < 16; ++t) Bs[buf][q * 16 + t][n] = breg[t];
#pragma unroll
        for (int r = 0; r < TM / 16; ++r) {
            const int idx4 = tid + 256 * r, mm = idx4 >> 4, kk = (idx4 & 15) * 4;
            *reinterpret_cast<float2*>(&As[buf][mm][kk]) = make_float2(areg[r].x, areg[r].y);
            *reinterpret_cast<float2*>(&As[buf][mm][kk + 2]) = make_float2(areg[r].z, areg[r].w);
        }
    };
    issue(c_lo);
    commit(0);
    __syncthreads();
    for (int c = c_lo; c < c_hi; ++c) {
        const int buf = (c - c_lo) & 1;
        const bool more = c + 1 < c_hi;
        if (more) issue(c + 1);
#pragma unroll 4
        for (int k4 = 0; k4 < 16; ++k4) {
            float a[RT], bb[2];
#pragma unroll
            for (int r = 0; r < RT; ++r) a[r] = As[buf][row0 + 16 * r + li][4 * k4 + lk];
#pragma unroll
            for (int s = 0; s < 2; ++s) bb[s] = Bs[buf][4 * k4 + lk][col0 + 16 * s + li];
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int s = 0; s < 2; ++s) acc[r][s] = mfma4_3(a[r], bb[s], acc[r][s]);
        }
        if (more) commit(buf ^ 1);
        __syncthreads();
    }
    float* ob = coarse + (size_t)b * M * vol;
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int pp = p0 + col0 + 16 * s + li;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int m = m0 + row0 + 16 * r + 4 * lk + qq;
                if (m < M && pp < vol) ob[(size_t)m * vol + pp] = acc[r][s][qq];
            }
        }
}

// ---- up -------------------------------------------------------------------------------------------------------------------------
// Output parity (pz, py, px); along one axis: parity 0 -> taps (k, d) = (1, 0), (3, -1); parity 1 -> (0, +1), (2, 0).
// blockIdx.z = pz.  One MFMA k-step = one coarse channel x the 4 in-plane taps (a_y, a_x) of one z tap a_z.
template <int TC>
__global__ void __launch_bounds__(256) conv3d_k4s2_up_kernel(const float* __restrict__ coarse, const float* __restrict__ w,
                                                             float* __restrict__ fine, int B, int C, int M, int D, int H, int W, int ss,
                                                             size_t slice_stride) {
    constexpr int KC = 2;                                // coarse channels per K-step
    constexpr int RT = TC / 32;
    // weights of the step, the 32 taps whose kz has this block's z parity: Ws[mm][az][ky * 4 + kx][c]
    __shared__ __attribute__((aligned(16))) float Ws[2][KC][2][16][TC + 8];
    // coarse neighbourhood: Ps[mm][az][(dj + 1) * 3 + dk + 1][n]
    __shared__ __attribute__((aligned(16))) float Ps[2][KC][2][9][C3_SB];
    const int tid = threadIdx.x;
    const int pz = blockIdx.z & 1, slice = blockIdx.z >> 1;      // split K: slice of `ss` K-steps, partial result at fine + slice * slice_stride
    fine += (size_t)slice * slice_stride;
    const int HW = H * W, vol = D * HW, tiles_per_sample = (vol + C3_TN - 1) / C3_TN;
    const int b = blockIdx.x / tiles_per_sample, p0 = (blockIdx.x % tiles_per_sample) * C3_TN;
    const int c0 = blockIdx.y * TC;
    const int n = tid & 63, q = tid >> 6;                 // staging: position p0 + n; q = (mm, az)
    const int P = p0 + n;
    const bool pok = P < vol;
    const int I = pok ? P / HW : 0, J = pok ? (P % HW) / W : 0, K = pok ? P % W : 0;
    const float* cb = coarse + (size_t)b * M * vol;
    const int lane = tid & 63, wv = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int wm = wv >> 1, wn = wv & 1;
    const int row0 = wm * (TC / 2), col0 = wn * 32;
    const int ta = lk >> 1, tb = lk & 1;
    c3_f32x4 acc[4][RT][2];
#pragma unroll
    for (int par = 0; par < 4; ++par)
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int s = 0; s < 2; ++s) acc[par][r][s] = (c3_f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int NWL = KC * 2 * 16 * TC / 4 / 256;       // float4 weight loads per thread and step
    float preg[9];
    float4 wreg[NWL];
    const int smm = q >> 1, saz = q & 1;                  // this thread stages coarse channel mstep + smm, z tap saz
    const int kz_of[2] = {pz == 0 ? 1 : 0, pz == 0 ? 3 : 2};
    const int di_of[2] = {pz == 0 ? 0 : 1, pz == 0 ? -1 : 0};
    auto issue = [&](int mstep) {
        const int m = mstep + smm;
        const int ii = I + di_of[saz];
        const bool iok = pok && m < M && ii >= 0 && ii < D;
        const float* cm = cb + (size_t)(iok ? m : 0) * vol + (size_t)(iok ? ii : 0) * HW;
#pragma unroll
        for (int dj = -1; dj <= 1; ++dj) {
            const int jj = J + dj;
            const bool jok = iok && jj >= 0 && jj < H;
#pragma unroll
            for (int dk = -1; dk <= 1; ++dk) {
                const int kk = K + dk;
                preg[(dj + 1) * 3 + dk + 1] = (jok && kk >= 0 && kk < W) ? cm[jj * W + kk] : 0.f;
            }
        }
        // weights: (mm, az, c, 16 in-plane taps): KC * 2 * TC * 16 floats; source w[m][c][kz][16]
#pragma unroll
        for (int r = 0; r < NWL; ++r) {
            const int idx4 = tid + 256 * r;
            const int t4 = (idx4 & 3) * 4, cc = (idx4 >> 2) % TC, rest = (idx4 >> 2) / TC, az = rest & 1, mm = rest >> 1;
            wreg[r] = (mstep + mm < M && c0 + cc < C)
                          ? *reinterpret_cast<const float4*>(w + (((size_t)(mstep + mm) * C + c0 + cc) * 4 + kz_of[az]) * 16 + t4)
                          : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int d = 0; d < 9; ++d) Ps[buf][smm][saz][d][n] = preg[d];
#pragma unroll
        for (int r = 0; r < NWL; ++r) {
            const int idx4 = tid + 256 * r;
            const int t4 = (idx4 & 3) * 4, cc = (idx4 >> 2) % TC, rest = (idx4 >> 2) / TC, az = rest & 1, mm = rest >> 1;
            Ws[buf][mm][az][t4 + 0][cc] = wreg[r].x;
            Ws[buf][mm][az][t4 + 1][cc] = wreg[r].y;
            Ws[buf][mm][az][t4 + 2][cc] = wreg[r].z;
            Ws[buf][mm][az][t4 + 3][cc] = wreg[r].w;
        }
    };
    const int nsteps = (M + KC - 1) / KC;
    const int st_lo = slice * ss, st_hi = min(nsteps, st_lo + ss);
    issue(st_lo * KC);
    commit(0);
    __syncthreads();
    for (int st = st_lo; st < st_hi; ++st) {
        const int buf = (st - st_lo) & 1;
        const bool more = st + 1 < st_hi;
        if (more) issue((st + 1) * KC);
#pragma unroll
        for (int mm = 0; mm < KC; ++mm)
#pragma unroll
            for (int az = 0; az < 2; ++az)
#pragma unroll
                for (int py = 0; py < 2; ++py) {
                    const int ky = py == 0 ? (ta == 0 ? 1 : 3) : (ta == 0 ? 0 : 2);
                    const int dj = py == 0 ? (ta == 0 ? 0 : -1) : (ta == 0 ? 1 : 0);
#pragma unroll
                    for (int px = 0; px < 2; ++px) {
                        const int kx = px == 0 ? (tb == 0 ? 1 : 3) : (tb == 0 ? 0 : 2);
                        const int dk = px == 0 ? (tb == 0 ? 0 : -1) : (tb == 0 ? 1 : 0);
                        float a[RT], bb[2];
#pragma unroll
                        for (int r = 0; r < RT; ++r) a[r] = Ws[buf][mm][az][ky * 4 + kx][row0 + 16 * r + li];
#pragma unroll
                        for (int s = 0; s < 2; ++s) bb[s] = Ps[buf][mm][az][(dj + 1) * 3 + dk + 1][col0 + 16 * s + li];
#pragma unroll
                        for (int r = 0; r < RT; ++r)
#pragma unroll
                            for (int s = 0; s < 2; ++s) acc[py * 2 + px][r][s] = mfma4_3(a[r], bb[s], acc[py * 2 + px][r][s]);
                    }
                }
        if (more) commit(buf ^ 1);
        __syncthreads();
    }
    const int H2 = 2 * H, W2 = 2 * W;
    const size_t fvol = (size_t)8 * vol;
    float* fo = fine + (size_t)b * C * fvol;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int PP = p0 + col0 + 16 * s + li;
        if (PP >= vol) continue;
        const int II = PP / HW, JJ = (PP % HW) / W, KK = PP % W;
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int c = c0 + row0 + 16 * r + 4 * lk + qq;
                if (c >= C) continue;
                float* dst = fo + (size_t)c * fvol + ((size_t)(2 * II + pz) * H2 + 2 * JJ) * W2 + 2 * KK;
                *reinterpret_cast<float2*>(dst) = make_float2(acc[0][r][s][qq], acc[1][r][s][qq]);
                *reinterpret_cast<float2*>(dst + W2) = make_float2(acc[2][r][s][qq], acc[3][r][s][qq]);
            }
    }
}

// out[i] = part[0][i] + part[1][i] + ... in slice order (deterministic)
__global__ void __launch_bounds__(256) conv3d_ksum_kernel(const float* __restrict__ part, float* __restrict__ out, size_t n, int S, size_t stride) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float v = part[i];
        for (int k = 1; k < S; ++k) v += part[(size_t)k * stride + i];
        out[i] = v;
    }
}

// Split of the contraction over workgroups for the deep, narrow layers of the generator (128 -> 128 channels on 4^3 positions is ONE position tile: two
// workgroups walked 8192 products each in 250 us -- gpurun_out/t14_kt, the 128^3 step spent 1.7 of its 3.4 ms in launches of <= 32 workgroups): `units`
// K-steps go to S slices so that the launch has ~512 workgroups, >= 4 steps per slice; returns the steps per slice (S = ceil(units / steps))
static int c3_split(int64_t wgs, int units, int* S) {
    *S = 1;
    if (wgs >= 256 || units < 8) return units;
    int s = (int)std::min<int64_t>((512 + wgs - 1) / wgs, units / 4);
    if (s < 2) return units;
    const int per = (units + s - 1) / s;
    *S = (units + per - 1) / per;
    return per;
}

static int c3_check(int64_t B, int64_t C, int64_t M, int64_t D, int64_t H, int64_t W) {
    if (B < 1 || C < 1 || M < 1 || D < 1 || H < 1 || W < 1) return DN_E_BADARG;
    if (8 * D * H * W >= (1ll << 31) || B * ((D * H * W + 63) / 64) >= (1ll << 31)) return DN_E_UNSUPPORTED;
    return 0;
}

}  // namespace dn

using namespace dn;

extern "C" int64_t dn_conv3d_k4s2_workspace_bytes(int32_t up, int64_t B, int64_t C, int64_t M, int64_t D, int64_t H, int64_t W) {
    if (c3_check(B, C, M, D, H, W)) return DN_E_BADARG;
    const int64_t tiles = B * ((D * H * W + C3_TN - 1) / C3_TN);
    int S;
    if (up) {
        c3_split(tiles * ((C + (C <= 32 ? 31 : 63)) / (C <= 32 ? 32 : 64)) * 2, (int)((M + 1) / 2), &S);
        return S > 1 ? (int64_t)S * B * C * 8 * D * H * W * (int64_t)sizeof(float) : 0;
    }
    c3_split(tiles * ((M + (M <= 32 ? 31 : 63)) / (M <= 32 ? 32 : 64)), (int)C, &S);
    return S > 1 ? (int64_t)S * B * M * D * H * W * (int64_t)sizeof(float) : 0;
}

extern "C" int dn_conv3d_k4s2_down_ws(const float* fine, const float* w, float* coarse, int64_t B, int64_t C, int64_t M, int64_t D, int64_t H,
                                      int64_t W, void* workspace, int64_t workspace_bytes, void* stream) {
    if (int rc = c3_check(B, C, M, D, H, W)) return rc;
    if (!fine || !w || !coarse) return DN_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const unsigned tiles = (unsigned)(B * ((D * H * W + C3_TN - 1) / C3_TN));
    const unsigned mtiles = (unsigned)(M <= 32 ? (M + 31) / 32 : (M + 63) / 64);
    int S = 1, cs = (int)C;
    const size_t n = (size_t)(B * M * D * H * W);
    if (workspace) {
        cs = c3_split((int64_t)tiles * mtiles, (int)C, &S);
        if (S > 1 && workspace_bytes < (int64_t)((size_t)S * n * sizeof(float))) return DN_E_WORKSPACE;
    }
    float* dst = S > 1 ? reinterpret_cast<float*>(workspace) : coarse;
    if (M <= 32) {
        hipLaunchKernelGGL((conv3d_k4s2_down_kernel<32>), dim3(tiles, mtiles, S), dim3(256), 0, s, fine, w, dst, (int)B, (int)C, (int)M, (int)D, (int)H, (int)W, cs, n);
    } else {
        hipLaunchKernelGGL((conv3d_k4s2_down_kernel<64>), dim3(tiles, mtiles, S), dim3(256), 0, s, fine, w, dst, (int)B, (int)C, (int)M, (int)D, (int)H, (int)W, cs, n);
    }
    DN_LAUNCH_CHECK();
    if (S > 1) {
        hipLaunchKernelGGL(conv3d_ksum_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, s, dst, coarse, n, S, n);
        DN_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int dn_conv3d_k4s2_up_ws(const float* coarse, const float* w, float* fine, int64_t B, int64_t C, int64_t M, int64_t D, int64_t H,
                                    int64_t W, void* workspace, int64_t workspace_bytes, void* stream) {
    if (int rc = c3_check(B, C, M, D, H, W)) return rc;
    if (!fine || !w || !coarse) return DN_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const unsigned tiles = (unsigned)(B * ((D * H * W + C3_TN - 1) / C3_TN));
    const unsigned ctiles = (unsigned)(C <= 32 ? (C + 31) / 32 : (C + 63) / 64);
    const int nsteps = (int)((M + 1) / 2);
    int S = 1, ss = nsteps;
    const size_t n = (size_t)(B * C * 8 * D * H * W);
    if (workspace) {
        ss = c3_split((int64_t)tiles * ctiles * 2, nsteps, &S);
        if (S > 1 && workspace_bytes < (int64_t)((size_t)S * n * sizeof(float))) return DN_E_WORKSPACE;
    }
    float* dst = S > 1 ? reinterpret_cast<float*>(workspace) : fine;
    if (C <= 32) {
        hipLaunchKernelGGL((conv3d_k4s2_up_kernel<32>), dim3(tiles, ctiles, 2 * S), dim3(256), 0, s, coarse, w, dst, (int)B, (int)C, (int)M, (int)D, (int)H, (int)W, ss, n);
    } else {
        hipLaunchKernelGGL((conv3d_k4s2_up_kernel<64>), dim3(tiles, ctiles, 2 * S), dim3(256), 0, s, coarse, w, dst, (int)B, (int)C, (int)M, (int)D, (int)H, (int)W, ss, n);
    }
    DN_LAUNCH_CHECK();
    if (S > 1) {
        hipLaunchKernelGGL(conv3d_ksum_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, s, dst, fine, n, S, n);
        DN_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int dn_conv3d_k4s2_down(const float* fine, const float* w, float* coarse, int64_t B, int64_t C, int64_t M, int64_t D, int64_t H,
                                   int64_t W, void* stream) {
    return dn_conv3d_k4s2_down_ws(fine, w, coarse, B, C, M, D, H, W, nullptr, 0, stream);
}

extern "C" int dn_conv3d_k4s2_up(const float* coarse, const float* w, float* fine, int64_t B, int64_t C, int64_t M, int64_t D, int64_t H,
                                 int64_t W, void* stream) {
    return dn_conv3d_k4s2_up_ws(coarse, w, fine, B, C, M, D, H, W, nullptr, 0, stream);
}
