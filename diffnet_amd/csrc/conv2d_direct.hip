// Stride-1 "valid" convolutions with small kernels (k x k, k <= 7) and bias: the stem / head layers of the auto-encoder that sit
// behind an explicit ReflectionPad2d (reference DiffNet/networks/autoencoders.py:13 `nn.Conv2d(in_channels, dim*2, 7)`, :75
// `nn.Conv2d(dim*(i+1)*2, out_channels, 3)`, `nn.Conv2d(out_channels, out_channels, 7)`).  In the reference network these layers
// have 1 input or 1 output channel (1 -> 128 7x7, 128 -> 1 3x3, 1 -> 1 7x7): a few MFLOP per sample, so they are plain fp32 FMA
// kernels -- forward, input gradient, weight + bias gradient with fixed-order reductions (bitwise repeatable) -- not GEMMs.
//   y[b,co,i,j]   = bias[co] + sum_{ci,ky,kx} w[co,ci,ky,kx] x[b,ci,i+ky,j+kx]
//   gx[b,ci,y,x]  = sum_{co,ky,kx} w[co,ci,ky,kx] gy[b,co,y-ky,x-kx]
//   gw[co,ci,ky,kx] = sum_{b,i,j} gy[b,co,i,j] x[b,ci,i+ky,j+kx],   gb[co] = sum gy[b,co,:,:]
#include "dn_common.h"

namespace dn {

__global__ void __launch_bounds__(256) conv2d_valid_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ y, int B, int Ci, int Co,
                                                               int H, int W, int K) {
    const int Ho = H - K + 1, Wo = W - K + 1;
    const long n = (long)B * Co * Ho * Wo;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const int j = (int)(t % Wo), i = (int)((t / Wo) % Ho), co = (int)((t / ((long)Wo * Ho)) % Co), b = (int)(t / ((long)Wo * Ho * Co));
    float acc = bias ? bias[co] : 0.f;
    const float* xb = x + (size_t)b * Ci * H * W;
    const float* wc = w + (size_t)co * Ci * K * K;
    for (int ci = 0; ci < Ci; ++ci)
        for (int ky = 0; ky < K; ++ky) {
            const float* xr = xb + ((size_t)ci * H + i + ky) * W + j;
            const float* wr = wc + ((size_t)ci * K + ky) * K;
            for (int kx = 0; kx < K; ++kx) acc = fmaf(wr[kx], xr[kx], acc);
        }
    y[t] = acc;
}

__global__ void __launch_bounds__(256) conv2d_valid_bwd_data_kernel(const float* __restrict__ gy, const float* __restrict__ w, float* __restrict__ gx,
                                                                    int B, int Ci, int Co, int H, int W, int K) {
    const int Ho = H - K + 1, Wo = W - K + 1;
    const long n = (long)B * Ci * H * W;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const int xx = (int)(t % W), yy = (int)((t / W) % H), ci = (int)((t / ((long)W * H)) % Ci), b = (int)(t / ((long)W * H * Ci));
    float acc = 0.f;
    const float* gb = gy + (size_t)b * Co * Ho * Wo;
    for (int co = 0; co < Co; ++co)
        for (int ky = 0; ky < K; ++ky) {
            const int i = yy - ky;
            if (i < 0 || i >= Ho) continue;
            const float* gr = gb + ((size_t)co * Ho + i) * Wo;
            const float* wr = w + (((size_t)co * Ci + ci) * K + ky) * K;
            for (int kx = 0; kx < K; ++kx) {
                const int j = xx - kx;
                if (j >= 0 && j < Wo) acc = fmaf(wr[kx], gr[j], acc);
            }
        }
    gx[t] = acc;
}

// one workgroup per weight element (and one per bias element after them): fixed-order sum over (b, i, j)
__global__ void __launch_bounds__(256) conv2d_valid_bwd_weight_kernel(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ gw,
                                                                      float* __restrict__ gbias, int B, int Ci, int Co, int H, int W, int K) {
    __shared__ double red[8];
    const int Ho = H - K + 1, Wo = W - K + 1;
    const int nw = Co * Ci * K * K;
    const int e = blockIdx.x, tid = threadIdx.x;
    const long npos = (long)B * Ho * Wo;
    double s = 0.0;
    if (e < nw) {
        const int kx = e % K, ky = (e / K) % K, ci = (e / (K * K)) % Ci, co = e / (K * K * Ci);
        for (long p = tid; p < npos; p += 256) {
            const int j = (int)(p % Wo), i = (int)((p / Wo) % Ho), b = (int)(p / ((long)Wo * Ho));
            s += (double)gy[(((size_t)b * Co + co) * Ho + i) * Wo + j] * (double)x[(((size_t)b * Ci + ci) * H + i + ky) * W + j + kx];
        }
        s = block_sum(s, red, tid, 256);
        if (tid == 0) gw[e] = (float)s;
    } else {
        const int co = e - nw;
        for (long p = tid; p < npos; p += 256) {
            const int ij = (int)(p % ((long)Wo * Ho)), b = (int)(p / ((long)Wo * Ho));
            s += (double)gy[((size_t)b * Co + co) * Ho * Wo + ij];
        }
        s = block_sum(s, red, tid, 256);
        if (tid == 0) gbias[co] = (float)s;
    }
}

static int cd_check(int64_t B, int64_t Ci, int64_t Co, int64_t H, int64_t W, int64_t K) {
    if (B < 1 || Ci < 1 || Co < 1 || K < 1 || H < K || W < K) return DN_E_BADARG;
    if (K > 7 || B * Ci * H * W >= (1ll << 38) || B * Co * H * W >= (1ll << 38) || Co * Ci * K * K + Co >= (1ll << 31)) return DN_E_UNSUPPORTED;
    return 0;
}

}  // namespace dn

using namespace dn;

extern "C" int dn_conv2d_valid_fwd(const float* x, const float* w, const float* bias, float* y, int64_t B, int64_t Ci, int64_t Co, int64_t H,
                                   int64_t W, int64_t K, void* stream) {
    if (int rc = cd_check(B, Ci, Co, H, W, K)) return rc;
    if (!x || !w || !y) return DN_E_BADARG;
    const long n = (long)(B * Co * (H - K + 1) * (W - K + 1));
    hipLaunchKernelGGL(conv2d_valid_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, w, bias, y,
                       (int)B, (int)Ci, (int)Co, (int)H, (int)W, (int)K);
    DN_LAUNCH_CHECK();
    return 0;
}

extern "C" int dn_conv2d_valid_bwd_data(const float* gy, const float* w, float* gx, int64_t B, int64_t Ci, int64_t Co, int64_t H, int64_t W, int64_t K,
                                        void* stream) {
    if (int rc = cd_check(B, Ci, Co, H, W, K)) return rc;
    if (!gy || !w || !gx) return DN_E_BADARG;
    const long n = (long)(B * Ci * H * W);
    hipLaunchKernelGGL(conv2d_valid_bwd_data_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), gy, w, gx,
                       (int)B, (int)Ci, (int)Co, (int)H, (int)W, (int)K);
    DN_LAUNCH_CHECK();
    return 0;
}

extern "C" int dn_conv2d_valid_bwd_weight(const float* x, const float* gy, float* gw, float* gbias, int64_t B, int64_t Ci, int64_t Co, int64_t H,
                                          int64_t W, int64_t K, void* stream) {
    if (int rc = cd_check(B, Ci, Co, H, W, K)) return rc;
    if (!x || !gy || !gw) return DN_E_BADARG;
    const unsigned nblk = (unsigned)(Co * Ci * K * K + (gbias ? Co : 0));
    hipLaunchKernelGGL(conv2d_valid_bwd_weight_kernel, dim3(nblk), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, gy, gw, gbias, (int)B,
                       (int)Ci, (int)Co, (int)H, (int)W, (int)K);
    DN_LAUNCH_CHECK();
    return 0;
}
