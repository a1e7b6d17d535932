// Fused output block of the 2-D U-Net generator: Upsample(x2, nearest) -> ZeroPad2d((1,0,1,0)) -> Conv2d(C -> 1, 4x4,
// padding 1, bias) -> Sigmoid  (reference DiffNet/networks/unets.py:68-74, `self.final`), forward and both backward
// passes.  torch materialises the upsampled (B, C, 2h, 2w) tensor and its padded copy (4.3 GB each at C = 64, 512^2,
// B = 64) and runs a C -> 1 convolution on it; that block is ~60 % of a U-Net training step on MI355X
// (profiles/r1_unet_step_*.csv).  Here the low-resolution input is read once.
//
// With U[y][x] = in[y >> 1][x >> 1] (zero outside), the block is  z[Y][X] = b + sum_c sum_{ky,kx} W[c][ky][kx] U[c][Y+ky-2][X+kx-2].
// Output row Y = 2i + a reads low-resolution rows  a = 0: {i-1: ky 0,1; i: ky 2,3},  a = 1: {i-1: ky 0; i: ky 1,2; i+1: ky 3},
// so the 2 x 2 outputs of low-resolution pixel (i, j) need the 3 x 3 neighbourhood of `in` and 25 instead of 64 MACs per
// channel (taps that fall on the same low-resolution pixel are pre-summed: `weff`, 25 floats per channel).  Transposed:
// in[i] feeds output rows 2i-1 .. 2i+3 with row-tap sets {3}, {2,3}, {1,2}, {0,1}, {0}  ->  5 x 5 backward weights.
#include "dn_common.h"

namespace dn {

// forward effective weights [C][25]: phase (0,0) 2x2 | (0,1) 2x3 | (1,0) 3x2 | (1,1) 3x3, neighbour offsets -1..+1
// backward effective weights [C][25]: [dy][dx], gradient rows 2i-1+dy, columns 2j-1+dx
__global__ void upconv_weff_kernel(const float* __restrict__ w, float* __restrict__ wf, float* __restrict__ wb, int C) {
    // one thread per effective weight: ids [0, 25 C) forward, [25 C, 50 C) backward
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= 50 * C) return;
    const bool fwd = id < 25 * C;
    const int c = (id % (25 * C)) / 25, o = id % 25;
    const float* W = w + c * 16;
    int ly, hy, lx, hx;
    if (fwd) {
        // phase (a, b) and group (gy, gx) of entry o: (0,0) 2x2 | (0,1) 2x3 | (1,0) 3x2 | (1,1) 3x3
        int a, b, gy, gx;
        if (o < 4) { a = 0; b = 0; gy = o / 2; gx = o % 2; }
        else if (o < 10) { a = 0; b = 1; gy = (o - 4) / 3; gx = (o - 4) % 3; }
        else if (o < 16) { a = 1; b = 0; gy = (o - 10) / 2; gx = (o - 10) % 2; }
        else { a = 1; b = 1; gy = (o - 16) / 3; gx = (o - 16) % 3; }
        // tap range of group g in phase a:  a = 0: {0,1},{2,3};  a = 1: {0},{1,2},{3}
        auto lo = [](int a_, int g) { return a_ == 0 ? 2 * g : (g == 0 ? 0 : (g == 1 ? 1 : 3)); };
        auto hi = [](int a_, int g) { return a_ == 0 ? 2 * g + 1 : (g == 0 ? 0 : (g == 1 ? 2 : 3)); };
        ly = lo(a, gy); hy = hi(a, gy); lx = lo(b, gx); hx = hi(b, gx);
    } else {
        // gradient row 2i-1+dy: tap sets {3},{2,3},{1,2},{0,1},{0}
        const int blo[5] = {3, 2, 1, 0, 0}, bhi[5] = {3, 3, 2, 1, 0};
        ly = blo[o / 5]; hy = bhi[o / 5]; lx = blo[o % 5]; hx = bhi[o % 5];
    }
    float s = 0.f;
    for (int ky = ly; ky <= hy; ++ky)
        for (int kx = lx; kx <= hx; ++kx) s += W[ky * 4 + kx];
    (fwd ? wf : wb)[c * 25 + o] = s;
}

__device__ __forceinline__ float sigmoidf(float z) { return 1.f / (1.f + __expf(-z)); }

// One thread per low-resolution pixel: 3 x 3 neighbourhood per channel from global memory (neighbouring lanes share
// the lines: L1 hits), weights through scalar loads (uniform index), 2 x 2 outputs.
__global__ void __launch_bounds__(256) upconv_fwd_kernel(const float* __restrict__ in, const float* __restrict__ wf,
                                                         const float* __restrict__ bias_ptr, float* __restrict__ out, int C, int h, int w,
                                                         int act) {
    const float bias = bias_ptr ? bias_ptr[0] : 0.f;
    const int j = blockIdx.x * 64 + (threadIdx.x & 63), i = blockIdx.y * 4 + (threadIdx.x >> 6), b = blockIdx.z;
    if (j >= w || i >= h) return;
    const size_t plane = (size_t)h * w;
    const float* ib = in + (size_t)b * C * plane;
    const int im = i > 0 ? i - 1 : i, ip = i < h - 1 ? i + 1 : i, jm = j > 0 ? j - 1 : j, jp = j < w - 1 ? j + 1 : j;
    const float zt = i > 0 ? 1.f : 0.f, zb = i < h - 1 ? 1.f : 0.f, zl = j > 0 ? 1.f : 0.f, zr = j < w - 1 ? 1.f : 0.f;   // zero padding
    const unsigned o00 = im * w + jm, o01 = im * w + j, o02 = im * w + jp, o10 = i * w + jm, o11 = i * w + j, o12 = i * w + jp,
                   o20 = ip * w + jm, o21 = ip * w + j, o22 = ip * w + jp;
    float z00 = bias, z01 = bias, z10 = bias, z11 = bias;
#pragma unroll 1                                   // unrolling over channels (2 / 4) was measured much slower (2-D: 124 -> 481 / 610 us at B=16, 64 x 256^2)
    for (int c = 0; c < C; ++c) {
        const float* p = ib + (size_t)c * plane;
        const float* k = wf + c * 25;
        const float v00 = p[o00] * (zt * zl), v01 = p[o01] * zt, v02 = p[o02] * (zt * zr);
        const float v10 = p[o10] * zl, v11 = p[o11], v12 = p[o12] * zr;
        const float v20 = p[o20] * (zb * zl), v21 = p[o21] * zb, v22 = p[o22] * (zb * zr);
        z00 += k[0] * v00 + k[1] * v01 + k[2] * v10 + k[3] * v11;
        z01 += k[4] * v00 + k[5] * v01 + k[6] * v02 + k[7] * v10 + k[8] * v11 + k[9] * v12;
        z10 += k[10] * v00 + k[11] * v01 + k[12] * v10 + k[13] * v11 + k[14] * v20 + k[15] * v21;
        z11 += k[16] * v00 + k[17] * v01 + k[18] * v02 + k[19] * v10 + k[20] * v11 + k[21] * v12 + k[22] * v20 + k[23] * v21 + k[24] * v22;
    }
    if (act) { z00 = sigmoidf(z00); z01 = sigmoidf(z01); z10 = sigmoidf(z10); z11 = sigmoidf(z11); }
    float* ob = out + (size_t)b * 4 * plane + (size_t)(2 * i) * (2 * w) + 2 * j;
    *reinterpret_cast<float2*>(ob) = make_float2(z00, z01);
    *reinterpret_cast<float2*>(ob + 2 * w) = make_float2(z10, z11);
}

// gz = gout * y (1 - y) (sigmoid) or gout, zero outside the image: the 5 x 5 patch around the high-resolution pixel (2 i - 1, 2 j - 1).
// Round 4: branch-free -- every load goes to a clamped (always legal) address and out-of-image values are zeroed with an opaque bit mask
// afterwards, the sigmoid factor is read in ONE wave-uniform branch around all 25 loads.  The round-2 form (a bounds branch and an `act`
// branch around each load) made every load wait for the one before it (a load inside a branch is drained where the branch joins):
// 50 dependent memory round trips per pixel -- the weight gradient of the U-Net's output block took 361 us for 285 MB.
__device__ __forceinline__ void gz_patch(const float* __restrict__ g, const float* __restrict__ y, int i, int j, int H, int W, int act, bool live,
                                         float (&G)[25]) {
    unsigned off[25], msk[25];
#pragma unroll
    for (int dy = 0; dy < 5; ++dy)
#pragma unroll
        for (int dx = 0; dx < 5; ++dx) {
            const int Y = 2 * i - 1 + dy, X = 2 * j - 1 + dx;
            off[dy * 5 + dx] = (unsigned)min(max(Y, 0), H - 1) * (unsigned)W + (unsigned)min(max(X, 0), W - 1);
            unsigned m = (live && Y >= 0 && Y < H && X >= 0 && X < W) ? 0xffffffffu : 0u;
            asm volatile("" : "+v"(m));                   // opaque: the compiler must not turn the mask back into a branch around the load
            msk[dy * 5 + dx] = m;
        }
#pragma unroll
    for (int t = 0; t < 25; ++t) G[t] = g[off[t]];
    if (act) {
        float Yv[25];
#pragma unroll
        for (int t = 0; t < 25; ++t) Yv[t] = y[off[t]];
#pragma unroll
        for (int t = 0; t < 25; ++t) G[t] = G[t] * Yv[t] * (1.f - Yv[t]);
    }
#pragma unroll
    for (int t = 0; t < 25; ++t) G[t] = __uint_as_float(__float_as_uint(G[t]) & msk[t]);
}

// grad wrt the low-resolution input: one thread per pixel, the 5 x 5 gradient patch in registers, loop over channels.
__global__ void __launch_bounds__(256) upconv_bwd_data_kernel(const float* __restrict__ gout, const float* __restrict__ y,
                                                              const float* __restrict__ wb, float* __restrict__ gin, int C, int h, int w,
                                                              int act) {
    const int j = blockIdx.x * 64 + (threadIdx.x & 63), i = blockIdx.y * 4 + (threadIdx.x >> 6), b = blockIdx.z;
    if (j >= w || i >= h) return;
    const int H = 2 * h, W = 2 * w;
    const float* gb = gout + (size_t)b * H * W;
    const float* yb = y ? y + (size_t)b * H * W : nullptr;
    float G[25];
    gz_patch(gb, yb, i, j, H, W, act, true, G);
    const size_t plane = (size_t)h * w;
    float* ob = gin + (size_t)b * C * plane + (size_t)i * w + j;
    for (int c = 0; c < C; ++c) {
        const float* k = wb + c * 25;
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 25; ++t) s = fmaf(k[t], G[t], s);
        ob[(size_t)c * plane] = s;
    }
}

// grad wrt the 4 x 4 x C weights and the bias.  gW[c][ky][kx] = sum_p in[c][p] * S[ky][kx][p] with S the 2 x 2 box sum of gz at
// rows 2i+2-ky .. +1, columns 2j+2-kx .. +1: a (C x pixels) x (pixels x 16) product.  A workgroup walks its share of the pixels
// 128 at a time: 128 threads compute the 16 box sums of one pixel each into LDS, all stage `in` for 64 channels, then the four
// waves accumulate one 16-channel x 16-tap tile each on the matrix cores (v_mfma_f32_16x16x4_f32, exact fp32).
// Per-workgroup partials are summed in index order by upconv_wsum_kernel.
constexpr int UC_CCH = 64;      // channels staged per pass
constexpr int UC_TP = 128;      // pixels per tile (44 KB of LDS per workgroup: three workgroups per CU)
__global__ void __launch_bounds__(256) upconv_bwd_weight_kernel(const float* __restrict__ in, const float* __restrict__ gout,
                                                                const float* __restrict__ y, float* __restrict__ part, int B, int C, int h,
                                                                int w, int act, int tiles_per_wg, int vec16) {
    // row strides 16 / 132 floats: the MFMA operand reads (16 lanes along a row, 4 lane groups along k) hit distinct banks
    __shared__ __attribute__((aligned(16))) float S[UC_TP][16];      // [pixel][tap]
    __shared__ __attribute__((aligned(16))) float V[UC_CCH][UC_TP + 4];   // [channel][pixel]
    __shared__ double red[8];
    const int tid = threadIdx.x;
    const int H = 2 * h, W = 2 * w;
    const size_t plane = (size_t)h * w;
    const long npix = (long)B * h * w;
    const int nc_pass = (C + UC_CCH - 1) / UC_CCH;
    // staging roles: pixel pp = tid % 128, channel parity half = tid / 128; accumulation on the matrix cores
    // (v_mfma_f32_16x16x4_f32, exact fp32): wave wv owns channels 16 wv .. 16 wv + 15, A[row li][k lk] = V, B[k lk][col li] = S
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int pp = tid & (UC_TP - 1), half = tid >> 7;
    const int lane = tid & 63, wv = tid >> 6, li = lane & 15, lk = lane >> 4;
    float bias_acc = 0.f;
    const size_t nwg = gridDim.x;
    float* pw = part + blockIdx.x;             // partials are stored [output][workgroup]
    for (int pass = 0; pass < nc_pass; ++pass) {
        const int c0 = pass * UC_CCH;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < tiles_per_wg; ++t) {
            const long p = ((long)blockIdx.x * tiles_per_wg + t) * UC_TP + pp;
            const bool ok = p < npix;
            int b = 0, i = 0, j = 0;
            if (ok) { b = (int)(p / plane); const int r = (int)(p % plane); i = r / w; j = r % w; }
            __syncthreads();
            if (half == 0) {   // box sums of this thread's pixel
                float G[25];
                const float* gb = gout + (size_t)b * H * W;
                const float* yb = y ? y + (size_t)b * H * W : gb;
                gz_patch(gb, yb, i, j, H, W, act, ok, G);
                // rows 2i+2-ky, 2i+3-ky  ->  patch rows dy = 3-ky, 4-ky
#pragma unroll
                for (int ky = 0; ky < 4; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 4; ++kx) {
                        const int dy = 3 - ky, dx = 3 - kx;
                        S[pp][ky * 4 + kx] = (G[dy * 5 + dx] + G[dy * 5 + dx + 1]) + (G[(dy + 1) * 5 + dx] + G[(dy + 1) * 5 + dx + 1]);
                    }
                if (pass == 0) bias_acc += (G[1 * 5 + 1] + G[1 * 5 + 2]) + (G[2 * 5 + 1] + G[2 * 5 + 2]);   // rows 2i, 2i+1
            }
            if (vec16) {
                // planes of a multiple of 128 pixels (the U-Net's shapes): a tile is 128 CONSECUTIVE floats of every channel plane -- 16 bytes per lane,
                // 8 loads per thread and tile instead of 32 (round 4: the launch was bound by its vector-memory instruction count, 360 -> see profiles)
                const long p0 = ((long)blockIdx.x * tiles_per_wg + t) * UC_TP;
                const bool tok = p0 < npix;
                const float* ib0 = in + (tok ? (size_t)(p0 / (long)plane) * C * plane + (size_t)(p0 % (long)plane) : (size_t)0);
#pragma unroll
                for (int kq = 0; kq < (UC_CCH * UC_TP / 4) / 256; ++kq) {
                    const int idx = tid + 256 * kq, c = idx >> 5, q = idx & 31;
                    unsigned mk = (tok && c0 + c < C) ? 0xffffffffu : 0u;
                    asm volatile("" : "+v"(mk));
                    const float4 v = *reinterpret_cast<const float4*>(ib0 + (size_t)min(c0 + c, C - 1) * plane + 4 * q);
                    *reinterpret_cast<float4*>(&V[c][4 * q]) = make_float4(__uint_as_float(__float_as_uint(v.x) & mk), __uint_as_float(__float_as_uint(v.y) & mk),
                                                                          __uint_as_float(__float_as_uint(v.z) & mk), __uint_as_float(__float_as_uint(v.w) & mk));
                }
            } else {
                const float* ib = in + (size_t)b * C * plane + (size_t)i * w + j;
                // (unconditional loads of clamped channels, zeroed by an opaque mask: a load in a branch waits for the one before it, see gz_patch)
#pragma unroll 8
                for (int c = half; c < UC_CCH; c += 2) {
                    unsigned mk = (ok && c0 + c < C) ? 0xffffffffu : 0u;
                    asm volatile("" : "+v"(mk));
                    V[c][pp] = __uint_as_float(__float_as_uint(ib[(size_t)min(c0 + c, C - 1) * plane]) & mk);
                }
            }
            __syncthreads();
#pragma unroll 8
            for (int q0 = 0; q0 < UC_TP; q0 += 4)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(V[16 * wv + li][q0 + lk], S[q0 + lk][li], acc, 0, 0, 0);
        }
        // C/D layout of the 16 x 16 tile: column (tap) = lane & 15, row (channel) = 4 (lane >> 4) + register
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ch = c0 + 16 * wv + 4 * lk + k;
            if (ch < C) pw[(size_t)(ch * 16 + li) * nwg] = acc[k];
        }
    }
    const double bs = block_sum((double)bias_acc, red, tid, 256);
    if (tid == 0) pw[(size_t)(C * 16) * nwg] = (float)bs;
}

// Sum of the per-workgroup partials, one wave per output: lanes stride over the workgroups (partials are stored
// [output][workgroup], so the reads are coalesced), fp64 accumulation, fixed butterfly order.
__global__ void __launch_bounds__(256) upconv_wsum_kernel(const float* __restrict__ part, float* __restrict__ gw, float* __restrict__ gbias,
                                                          int nwg, int n) {
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (k > n) return;                          // k == n: the bias entry
    const float* pk = part + (size_t)k * nwg;
    double s = 0.0;
    for (int g = lane; g < nwg; g += 64) s += (double)pk[g];
    s = wave_sum(s);
    if (lane == 0) {
        if (k < n) gw[k] = (float)s;
        else if (gbias) *gbias = (float)s;
    }
}

static int uc_plan(int64_t npix, int& nwg, int& tiles_per_wg) {
    const int64_t tiles = (npix + UC_TP - 1) / UC_TP;
    int64_t want = 3072;                       // 3 resident workgroups per CU x 4 rounds
    if (want > tiles) want = tiles;
    tiles_per_wg = (int)((tiles + want - 1) / want);
    nwg = (int)((tiles + tiles_per_wg - 1) / tiles_per_wg);
    return 0;
}

}  // namespace dn

using namespace dn;

static int uc_check(int64_t B, int64_t C, int64_t h, int64_t w) {
    if (B < 1 || C < 1 || h < 1 || w < 1 || B > 65535 || C > 4096 || h > 32768 || w > 32768) return DN_E_BADARG;
    if (B * C * h * w >= (1ll << 40)) return DN_E_UNSUPPORTED;
    return 0;
}

extern "C" int64_t dn_upconv_out_workspace_bytes(int64_t B, int64_t C, int64_t h, int64_t w) {
    if (uc_check(B, C, h, w)) return DN_E_BADARG;
    int nwg, tpw;
    uc_plan(B * h * w, nwg, tpw);
    return (int64_t)sizeof(float) * (2 * C * 25 + (int64_t)nwg * (C * 16 + 1));
}

extern "C" int dn_upconv_out_fwd(const float* in, const float* weight, const float* bias, float* out, int64_t B, int64_t C, int64_t h,
                                 int64_t w, int act, void* workspace, int64_t workspace_bytes, void* stream) {
    if (int rc = uc_check(B, C, h, w)) return rc;
    if (!in || !weight || !out || !workspace || workspace_bytes < dn_upconv_out_workspace_bytes(B, C, h, w)) return DN_E_WORKSPACE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float* wf = static_cast<float*>(workspace);
    float* wb = wf + C * 25;
    hipLaunchKernelGGL(upconv_weff_kernel, dim3((unsigned)((50 * C + 255) / 256)), dim3(256), 0, s, weight, wf, wb, (int)C);
    hipLaunchKernelGGL(upconv_fwd_kernel, dim3((unsigned)((w + 63) / 64), (unsigned)((h + 3) / 4), (unsigned)B), dim3(256), 0, s, in, wf, bias,
                       out, (int)C, (int)h, (int)w, act);
    DN_LAUNCH_CHECK();
    return 0;
}

extern "C" int dn_upconv_out_bwd(const float* in, const float* weight, const float* out, const float* grad_out, float* grad_in,
                                 float* grad_weight, float* grad_bias, int64_t B, int64_t C, int64_t h, int64_t w, int act, void* workspace,
                                 int64_t workspace_bytes, void* stream) {
    if (int rc = uc_check(B, C, h, w)) return rc;
    if (!weight || !grad_out || (act && !out)) return DN_E_BADARG;
    if (!workspace || workspace_bytes < dn_upconv_out_workspace_bytes(B, C, h, w)) return DN_E_WORKSPACE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float* wf = static_cast<float*>(workspace);
    float* wb = wf + C * 25;
    float* part = wb + C * 25;
    if (grad_in) {
        hipLaunchKernelGGL(upconv_weff_kernel, dim3((unsigned)((50 * C + 255) / 256)), dim3(256), 0, s, weight, wf, wb, (int)C);
        hipLaunchKernelGGL(upconv_bwd_data_kernel, dim3((unsigned)((w + 63) / 64), (unsigned)((h + 3) / 4), (unsigned)B), dim3(256), 0, s, grad_out,
                           act ? out : nullptr, wb, grad_in, (int)C, (int)h, (int)w, act);
    }
    if (grad_weight) {
        if (!in) return DN_E_BADARG;
        int nwg, tpw;
        uc_plan(B * h * w, nwg, tpw);
        const int vec16 = ((long long)h * w) % UC_TP == 0 && (reinterpret_cast<uintptr_t>(in) % 16) == 0;
        hipLaunchKernelGGL(upconv_bwd_weight_kernel, dim3((unsigned)nwg), dim3(256), 0, s, in, grad_out, act ? out : nullptr, part, (int)B, (int)C,
                           (int)h, (int)w, act, tpw, vec16);
        const int n = (int)C * 16;
        hipLaunchKernelGGL(upconv_wsum_kernel, dim3((unsigned)((n + 1 + 3) / 4)), dim3(256), 0, s, part, grad_weight, grad_bias, nwg, n);
    }
    DN_LAUNCH_CHECK();
    return 0;
}
