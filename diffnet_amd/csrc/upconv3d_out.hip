// Fused output block of the 3-D generator: Upsample(x2, nearest) -> Conv3d(C -> 1, 3x3x3, padding 1, bias) -> Sigmoid
// (reference DiffNet/networks/wgan3d.py:88-92, `self.final`), forward and both backward passes, without the upsampled
// (B, C, 2d, 2h, 2w) tensor (268 MB per sample at C = 32, 128^3).  On MI355X MIOpen's weight gradient of that convolution
// takes 321 ms at 128^3 (87 % of a training step, profiles/r1_gen3d_step.txt); here the three passes read the
// low-resolution input once each.
//
// z[Z][Y][X] = b + sum_c sum_k W[c][kz][ky][kx] U[c][Z+kz-1][Y+ky-1][X+kx-1],  U[z] = in[z >> 1], zero outside.
// Output plane Z = 2i + a reads  a = 0: {i-1: kz 0; i: kz 1,2},  a = 1: {i: kz 0,1; i+1: kz 2}: every phase sees 2 x 2 x 2
// low-resolution voxels (64 pre-summed weights per channel, `wf`); transposed, in[i] feeds planes 2i-1 .. 2i+2 with tap
// sets {2}, {1,2}, {0,1}, {0} (4 x 4 x 4 backward weights, `wb`).
#include "dn_common.h"

namespace dn {

// wf[c][phase = a*4+b*2+e][gz*4+gy*2+gx]   (gz = 0: lower plane of the phase's pair, 1: upper)
// wb[c][dz*16+dy*4+dx]                      (gradient voxel 2i-1+dz, 2j-1+dy, 2k-1+dx)
__global__ void upconv3d_weff_kernel(const float* __restrict__ w, float* __restrict__ wf, float* __restrict__ wb, int C) {
    // one thread per effective weight: ids [0, 64 C) forward, [64 C, 128 C) backward
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= 128 * C) return;
    const int c = (id % (64 * C)) / 64, o = id % 64;
    const float* W = w + c * 27;
    int lz, hz, ly, hy, lx, hx;
    if (id < 64 * C) {
        // tap range of group g in phase a:  a = 0: g0 = {0}, g1 = {1,2};  a = 1: g0 = {0,1}, g1 = {2}
        auto lo = [](int a, int g) { return a == 0 ? (g == 0 ? 0 : 1) : (g == 0 ? 0 : 2); };
        auto hi = [](int a, int g) { return a == 0 ? (g == 0 ? 0 : 2) : (g == 0 ? 1 : 2); };
        const int ph = o / 8, g = o % 8;
        const int a = ph >> 2, b = (ph >> 1) & 1, e = ph & 1, gz = g >> 2, gy = (g >> 1) & 1, gx = g & 1;
        lz = lo(a, gz); hz = hi(a, gz); ly = lo(b, gy); hy = hi(b, gy); lx = lo(e, gx); hx = hi(e, gx);
    } else {
        const int blo[4] = {2, 1, 0, 0}, bhi[4] = {2, 2, 1, 0};
        const int dz = o / 16, dy = (o / 4) % 4, dx = o % 4;
        lz = blo[dz]; hz = bhi[dz]; ly = blo[dy]; hy = bhi[dy]; lx = blo[dx]; hx = bhi[dx];
    }
    float s = 0.f;
    for (int kz = lz; kz <= hz; ++kz)
        for (int ky = ly; ky <= hy; ++ky)
            for (int kx = lx; kx <= hx; ++kx) s += W[(kz * 3 + ky) * 3 + kx];
    (id < 64 * C ? wf : wb)[c * 64 + o] = s;
}

__device__ __forceinline__ float sigmoid3(float z) { return 1.f / (1.f + __expf(-z)); }

// One thread per low-resolution voxel: its 3 x 3 x 3 neighbourhood per channel (zero outside), 8 outputs.
__global__ void __launch_bounds__(256) upconv3d_fwd_kernel(const float* __restrict__ in, const float* __restrict__ wf,
                                                           const float* __restrict__ bias_ptr, float* __restrict__ out, int C, int d, int h,
                                                           int w, int act) {
    const int k = blockIdx.x * 64 + (threadIdx.x & 63), j = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int i = blockIdx.z % d, b = blockIdx.z / d;
    if (k >= w || j >= h) return;
    const size_t vol = (size_t)d * h * w;
    const float* ib = in + (size_t)b * C * vol;
    const float bias = bias_ptr ? bias_ptr[0] : 0.f;
    unsigned off[27];
    float msk[27];
#pragma unroll
    for (int dz = 0; dz < 3; ++dz)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int z = i + dz - 1, y = j + dy - 1, x = k + dx - 1;
                const bool okv = z >= 0 && z < d && y >= 0 && y < h && x >= 0 && x < w;
                const int zc = min(max(z, 0), d - 1), yc = min(max(y, 0), h - 1), xc = min(max(x, 0), w - 1);
                off[(dz * 3 + dy) * 3 + dx] = (unsigned)((zc * h + yc) * w + xc);
                msk[(dz * 3 + dy) * 3 + dx] = okv ? 1.f : 0.f;
            }
    float z8[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) z8[p] = bias;
#pragma unroll 1                                   // unrolling over channels (2 / 4) was measured much slower (3-D: 350 -> 380 / 4940 us at 32 x 128^3)
    for (int c = 0; c < C; ++c) {
        const float* pc = ib + (size_t)c * vol;
        const float* kf = wf + c * 64;
        float v[27];
#pragma unroll
        for (int t = 0; t < 27; ++t) v[t] = pc[off[t]] * msk[t];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    float s = z8[a * 4 + bb * 2 + e];
#pragma unroll
                    for (int gz = 0; gz < 2; ++gz)
#pragma unroll
                        for (int gy = 0; gy < 2; ++gy)
#pragma unroll
                            for (int gx = 0; gx < 2; ++gx)      // phase a pairs planes (i-1, i) for a = 0 and (i, i+1) for a = 1
                                s = fmaf(kf[(a * 4 + bb * 2 + e) * 8 + gz * 4 + gy * 2 + gx], v[((a + gz) * 3 + (bb + gy)) * 3 + (e + gx)], s);
                    z8[a * 4 + bb * 2 + e] = s;
                }
    }
    const int D2 = 2 * d, H2 = 2 * h, W2 = 2 * w;
    float* ob = out + (size_t)b * D2 * H2 * W2;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
            float o0 = z8[a * 4 + bb * 2], o1 = z8[a * 4 + bb * 2 + 1];
            if (act) { o0 = sigmoid3(o0); o1 = sigmoid3(o1); }
            *reinterpret_cast<float2*>(ob + ((size_t)(2 * i + a) * H2 + (2 * j + bb)) * W2 + 2 * k) = make_float2(o0, o1);
        }
}

// gz = gout * y (1 - y) (sigmoid) or gout over the 4 x 4 x 4 patch around the high-resolution voxel (2 i - 1, 2 j - 1, 2 k - 1), zero outside
// the volume.  Round 4: branch-free (see gz_patch in upconv_out.hip: a bounds branch and an `act` branch around each of the 64 loads made
// every load wait for the one before it) -- loads of clamped addresses, an opaque bit mask afterwards, ONE uniform branch for the sigmoid factor.
__device__ __forceinline__ void gz3_patch(const float* __restrict__ g, const float* __restrict__ y, int i, int j, int k, int D, int H, int W, int act,
                                          bool live, float (&G)[64]) {
    unsigned zoff[4], yoff[4], xoff[4];
    bool zok[4], yok[4], xok[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int Z = 2 * i - 1 + t, Y = 2 * j - 1 + t, X = 2 * k - 1 + t;
        zok[t] = Z >= 0 && Z < D; yok[t] = Y >= 0 && Y < H; xok[t] = X >= 0 && X < W;
        zoff[t] = (unsigned)min(max(Z, 0), D - 1) * (unsigned)(H * W);
        yoff[t] = (unsigned)min(max(Y, 0), H - 1) * (unsigned)W;
        xoff[t] = (unsigned)min(max(X, 0), W - 1);
    }
#pragma unroll
    for (int dz = 0; dz < 4; ++dz)
#pragma unroll
        for (int dy = 0; dy < 4; ++dy)
#pragma unroll
            for (int dx = 0; dx < 4; ++dx) G[dz * 16 + dy * 4 + dx] = g[zoff[dz] + yoff[dy] + xoff[dx]];
    if (act) {
#pragma unroll
        for (int dz = 0; dz < 4; ++dz)
#pragma unroll
            for (int dy = 0; dy < 4; ++dy)
#pragma unroll
                for (int dx = 0; dx < 4; ++dx) {
                    const float yv = y[zoff[dz] + yoff[dy] + xoff[dx]];
                    G[dz * 16 + dy * 4 + dx] *= yv * (1.f - yv);
                }
    }
#pragma unroll
    for (int dz = 0; dz < 4; ++dz)
#pragma unroll
        for (int dy = 0; dy < 4; ++dy)
#pragma unroll
            for (int dx = 0; dx < 4; ++dx) {
                unsigned m = (live && zok[dz] && yok[dy] && xok[dx]) ? 0xffffffffu : 0u;
                asm volatile("" : "+v"(m));               // opaque: no branch around the load
                G[dz * 16 + dy * 4 + dx] = __uint_as_float(__float_as_uint(G[dz * 16 + dy * 4 + dx]) & m);
            }
}

// grad wrt the low-resolution input: 4 x 4 x 4 gradient patch in registers, loop over channels.
__global__ void __launch_bounds__(256) upconv3d_bwd_data_kernel(const float* __restrict__ gout, const float* __restrict__ y,
                                                                const float* __restrict__ wb, float* __restrict__ gin, int C, int d, int h,
                                                                int w, int act) {
    const int k = blockIdx.x * 64 + (threadIdx.x & 63), j = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int i = blockIdx.z % d, b = blockIdx.z / d;
    if (k >= w || j >= h) return;
    const int D2 = 2 * d, H2 = 2 * h, W2 = 2 * w;
    const float* gb = gout + (size_t)b * D2 * H2 * W2;
    const float* yb = y ? y + (size_t)b * D2 * H2 * W2 : nullptr;
    float G[64];
    gz3_patch(gb, yb, i, j, k, D2, H2, W2, act, true, G);
    const size_t vol = (size_t)d * h * w;
    float* ob = gin + (size_t)b * C * vol + ((size_t)i * h + j) * w + k;
    for (int c = 0; c < C; ++c) {
        const float* kb = wb + c * 64;
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 64; ++t) s = fmaf(kb[t], G[t], s);
        ob[(size_t)c * vol] = s;
    }
}

// grad wrt the 27 x C weights and the bias: gW[c][tap] = sum_p in[c][p] * S[tap][p], S = 2 x 2 x 2 box sum of gz at
// 2i+1-kz .. +1 (patch planes 2-kz, 3-kz).  Tiles of 128 voxels: 128 threads build the 27 box sums (+ the plain sum for
// the bias) of one voxel each, all stage `in` for up to 32 channels, then the four waves accumulate the four 16 x 16 tiles
// of the (32 channels x 32 padded taps) block on the matrix cores (v_mfma_f32_16x16x4_f32, exact fp32).
constexpr int U3_CCH = 32, U3_TP = 128;
__global__ void __launch_bounds__(256) upconv3d_bwd_weight_kernel(const float* __restrict__ in, const float* __restrict__ gout,
                                                                  const float* __restrict__ y, float* __restrict__ part, int B, int C, int d,
                                                                  int h, int w, int act, int tiles_per_wg) {
    // row strides 48 / 132 floats: the MFMA operand reads (16 lanes along a row, 4 lane groups along k) hit distinct banks
    __shared__ float S[U3_TP][48];                                   // taps 0..26, 27..31 zero (two 16-column tiles)
    __shared__ float V[U3_CCH][U3_TP + 4];
    __shared__ double red[8];
    const int tid = threadIdx.x;
    const int D2 = 2 * d, H2 = 2 * h, W2 = 2 * w;
    const size_t vol = (size_t)d * h * w;
    const long nvox = (long)B * vol;
    const int nc_pass = (C + U3_CCH - 1) / U3_CCH;
    const int pp = tid & (U3_TP - 1), half = tid >> 7;
    // accumulation on the matrix cores (v_mfma_f32_16x16x4_f32, exact fp32): the 32 x 32 (channel x tap) block is four
    // 16 x 16 tiles, one per wave: row tile rt (channels), column tile ct (taps); A[row li][k lk] = V, B[k lk][col li] = S
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int lane = tid & 63, wv = tid >> 6, li = lane & 15, lk = lane >> 4, rt = wv & 1, ct = wv >> 1;
    const size_t nwg = gridDim.x;
    float* pw = part + blockIdx.x;                        // partials are stored [output][workgroup]
    float bias_acc = 0.f;
    for (int pass = 0; pass < nc_pass; ++pass) {
        const int c0 = pass * U3_CCH;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < tiles_per_wg; ++t) {
            const long p = ((long)blockIdx.x * tiles_per_wg + t) * U3_TP + pp;
            const bool ok = p < nvox;
            int b = 0, i = 0, j = 0, k = 0;
            if (ok) {
                b = (int)(p / vol);
                int r = (int)(p % vol);
                i = r / (h * w); r %= h * w; j = r / w; k = r % w;
            }
            __syncthreads();
            if (half == 0) {
                const float* gb = gout + (size_t)b * D2 * H2 * W2;
                const float* yb = y ? y + (size_t)b * D2 * H2 * W2 : gb;
                float G[64];
                gz3_patch(gb, yb, i, j, k, D2, H2, W2, act, ok, G);
                // pair sums along x, then y, then z: X[dz][dy][q] = G[..][q] + G[..][q+1], q = 0..2, and so on
                float Xs[4][4][3];
#pragma unroll
                for (int dz = 0; dz < 4; ++dz)
#pragma unroll
                    for (int dy = 0; dy < 4; ++dy)
#pragma unroll
                        for (int q = 0; q < 3; ++q) Xs[dz][dy][q] = G[dz * 16 + dy * 4 + q] + G[dz * 16 + dy * 4 + q + 1];
                float Ys[4][3][3];
#pragma unroll
                for (int dz = 0; dz < 4; ++dz)
#pragma unroll
                    for (int q = 0; q < 3; ++q)
#pragma unroll
                        for (int r = 0; r < 3; ++r) Ys[dz][q][r] = Xs[dz][q][r] + Xs[dz][q + 1][r];
                // box starting at patch index (2-kz, 2-ky, 2-kx)
#pragma unroll
                for (int kz = 0; kz < 3; ++kz)
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) S[pp][(kz * 3 + ky) * 3 + kx] = Ys[2 - kz][2 - ky][2 - kx] + Ys[3 - kz][2 - ky][2 - kx];
#pragma unroll
                for (int z = 27; z < 32; ++z) S[pp][z] = 0.f;
                if (pass == 0) bias_acc += Ys[1][1][1] + Ys[2][1][1];      // planes 2i, 2i+1 (patch 1, 2): every output voxel once
            }
            const float* ib = in + (size_t)b * C * vol + ((size_t)i * h + j) * w + k;
#pragma unroll 8
            for (int c = half; c < U3_CCH; c += 2) {      // (unconditional loads of clamped channels + opaque mask: see gz3_patch)
                unsigned mk = (ok && c0 + c < C) ? 0xffffffffu : 0u;
                asm volatile("" : "+v"(mk));
                V[c][pp] = __uint_as_float(__float_as_uint(ib[(size_t)min(c0 + c, C - 1) * vol]) & mk);
            }
            __syncthreads();
#pragma unroll 8
            for (int q0 = 0; q0 < U3_TP; q0 += 4)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(V[16 * rt + li][q0 + lk], S[q0 + lk][16 * ct + li], acc, 0, 0, 0);
        }
        // C/D layout of the 16 x 16 tile: column (tap) = lane & 15, row (channel) = 4 (lane >> 4) + register
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ch = c0 + 16 * rt + 4 * lk + q, tap = 16 * ct + li;
            if (ch < C && tap < 27) pw[(size_t)(ch * 27 + tap) * nwg] = acc[q];
        }
    }
    const double bs = block_sum((double)bias_acc, red, tid, 256);
    if (tid == 0) pw[(size_t)(C * 27) * nwg] = (float)bs;
}

// Sum of the per-workgroup partials, one wave per output (see upconv_out.hip).
__global__ void __launch_bounds__(256) upconv3d_wsum_kernel(const float* __restrict__ part, float* __restrict__ gw, float* __restrict__ gbias,
                                                            int nwg, int n) {
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (k > n) return;
    const float* pk = part + (size_t)k * nwg;
    double s = 0.0;
    for (int g = lane; g < nwg; g += 64) s += (double)pk[g];
    s = wave_sum(s);
    if (lane == 0) {
        if (k < n) gw[k] = (float)s;
        else if (gbias) *gbias = (float)s;
    }
}

static void u3_plan(int64_t nvox, int& nwg, int& tiles_per_wg) {
    const int64_t tiles = (nvox + U3_TP - 1) / U3_TP;
    int64_t want = 4096;
    if (want > tiles) want = tiles;
    tiles_per_wg = (int)((tiles + want - 1) / want);
    nwg = (int)((tiles + tiles_per_wg - 1) / tiles_per_wg);
}

}  // namespace dn

using namespace dn;

static int u3_check(int64_t B, int64_t C, int64_t d, int64_t h, int64_t w) {
    if (B < 1 || C < 1 || d < 1 || h < 1 || w < 1 || C > 4096 || d > 4096 || h > 4096 || w > 4096) return DN_E_BADARG;
    if (B * d > 65535 || B * C * d * h * w >= (1ll << 40) || d * h * w >= (1ll << 28)) return DN_E_UNSUPPORTED;   // grid.z, 32-bit offsets
    return 0;
}

extern "C" int64_t dn_upconv3d_out_workspace_bytes(int64_t B, int64_t C, int64_t d, int64_t h, int64_t w) {
    if (u3_check(B, C, d, h, w)) return DN_E_BADARG;
    int nwg, tpw;
    u3_plan(B * d * h * w, nwg, tpw);
    return (int64_t)sizeof(float) * (2 * C * 64 + (int64_t)nwg * (C * 27 + 1));
}

extern "C" int dn_upconv3d_out_fwd(const float* in, const float* weight, const float* bias, float* out, int64_t B, int64_t C, int64_t d,
                                   int64_t h, int64_t w, int act, void* workspace, int64_t workspace_bytes, void* stream) {
    if (int rc = u3_check(B, C, d, h, w)) return rc;
    if (!in || !weight || !out) return DN_E_BADARG;
    if (!workspace || workspace_bytes < dn_upconv3d_out_workspace_bytes(B, C, d, h, w)) return DN_E_WORKSPACE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float* wf = static_cast<float*>(workspace);
    float* wb = wf + C * 64;
    hipLaunchKernelGGL(upconv3d_weff_kernel, dim3((unsigned)((128 * C + 255) / 256)), dim3(256), 0, s, weight, wf, wb, (int)C);
    hipLaunchKernelGGL(upconv3d_fwd_kernel, dim3((unsigned)((w + 63) / 64), (unsigned)((h + 3) / 4), (unsigned)(B * d)), dim3(256), 0, s, in, wf,
                       bias, out, (int)C, (int)d, (int)h, (int)w, act);
    DN_LAUNCH_CHECK();
    return 0;
}

extern "C" int dn_upconv3d_out_bwd(const float* in, const float* weight, const float* out, const float* grad_out, float* grad_in,
                                   float* grad_weight, float* grad_bias, int64_t B, int64_t C, int64_t d, int64_t h, int64_t w, int act,
                                   void* workspace, int64_t workspace_bytes, void* stream) {
    if (int rc = u3_check(B, C, d, h, w)) return rc;
    if (!weight || !grad_out || (act && !out)) return DN_E_BADARG;
    if (!workspace || workspace_bytes < dn_upconv3d_out_workspace_bytes(B, C, d, h, w)) return DN_E_WORKSPACE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float* wf = static_cast<float*>(workspace);
    float* wb = wf + C * 64;
    float* part = wb + C * 64;
    if (grad_in) {
        hipLaunchKernelGGL(upconv3d_weff_kernel, dim3((unsigned)((128 * C + 255) / 256)), dim3(256), 0, s, weight, wf, wb, (int)C);
        hipLaunchKernelGGL(upconv3d_bwd_data_kernel, dim3((unsigned)((w + 63) / 64), (unsigned)((h + 3) / 4), (unsigned)(B * d)), dim3(256), 0, s,
                           grad_out, act ? out : nullptr, wb, grad_in, (int)C, (int)d, (int)h, (int)w, act);
    }
    if (grad_weight) {
        if (!in) return DN_E_BADARG;
        int nwg, tpw;
        u3_plan(B * d * h * w, nwg, tpw);
        hipLaunchKernelGGL(upconv3d_bwd_weight_kernel, dim3((unsigned)nwg), dim3(256), 0, s, in, grad_out, act ? out : nullptr, part, (int)B,
                           (int)C, (int)d, (int)h, (int)w, act, tpw);
        const int n = (int)C * 27;
        hipLaunchKernelGGL(upconv3d_wsum_kernel, dim3((unsigned)((n + 1 + 3) / 4)), dim3(256), 0, s, part, grad_weight, grad_bias, nwg, n);
    }
    DN_LAUNCH_CHECK();
    return 0;
}
