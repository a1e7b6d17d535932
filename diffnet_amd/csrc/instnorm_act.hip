// Fused InstanceNorm (affine = False, biased variance, eps) + LeakyReLU / ReLU, forward and backward, for the
// generator blocks of SURVEY.md 8(a) rows a16-a18 (reference: `nn.InstanceNorm{2,3}d(c)` followed by
// `nn.LeakyReLU(0.2)` / `nn.ReLU` in DiffNet/networks/unets.py:13-45, autoencoders.py:7-70, wgan3d.py:23-55).
// torch runs these as a normalisation kernel plus an activation kernel each way; here a workgroup owns one
// (sample, channel) instance - or one of `nchunk` slices of it when there are too few instances to fill 256 CUs:
//   pass 1  sum / sum of squares in fp64 (fixed order: lane-strided, wave butterfly, waves in index order,
//           slices in index order -> bitwise repeatable),
//   pass 2  normalise, activate, store (the second read of the slice is an L2 hit for the sizes the networks use).
//   y = act((x - mean) * rstd),  act(t) = t > 0 ? t : slope * t        (slope 0 = ReLU, 1 = identity)
// Backward, with g = gy * act'(xhat):  gx = rstd * (g - mean(g) - xhat * mean(g * xhat)).
#include "dn_common.h"

namespace dn {

constexpr int IN_MAXCHUNK = 64;
constexpr long IN_MINCHUNK = 8192;      // elements; below this a slice is not worth a second launch

struct Slice { long lo, hi; bool vec; };

__device__ __forceinline__ Slice slice_of(const void* a, const void* b, const void* c, long S, int nchunk) {
    Slice s;
    long len = (S + nchunk - 1) / nchunk;
    len = (len + 3) & ~3l;
    s.lo = (long)blockIdx.y * len;
    s.hi = s.lo + len < S ? s.lo + len : S;
    if (s.lo > S) s.lo = S;
    s.vec = (S % 4 == 0) &&
            ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) % 16 == 0);
    return s;
}

template <int BLOCK>
__device__ __forceinline__ void fwd_sums(const float* xi, const Slice sl, double& s, double& q) {
    s = 0.0; q = 0.0;
    if (sl.vec) {
        const float4* x4 = reinterpret_cast<const float4*>(xi);
        for (long i = sl.lo / 4 + threadIdx.x; i < sl.hi / 4; i += BLOCK) {
            const float4 v = x4[i];
            s += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
            q += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
        }
    } else {
        for (long i = sl.lo + threadIdx.x; i < sl.hi; i += BLOCK) { const float v = xi[i]; s += v; q += (double)v * v; }
    }
}

__device__ __forceinline__ float act(float t, float slope) { return t > 0.f ? t : slope * t; }

template <int BLOCK>
__device__ __forceinline__ void fwd_apply(const float* xi, float* yi, const Slice sl, float m, float r, float slope) {
    if (sl.vec) {
        const float4* x4 = reinterpret_cast<const float4*>(xi);
        float4* y4 = reinterpret_cast<float4*>(yi);
        for (long i = sl.lo / 4 + threadIdx.x; i < sl.hi / 4; i += BLOCK) {
            float4 v = x4[i];
            v.x = act((v.x - m) * r, slope); v.y = act((v.y - m) * r, slope);
            v.z = act((v.z - m) * r, slope); v.w = act((v.w - m) * r, slope);
            y4[i] = v;
        }
    } else {
        for (long i = sl.lo + threadIdx.x; i < sl.hi; i += BLOCK) yi[i] = act((xi[i] - m) * r, slope);
    }
}

__device__ __forceinline__ void fwd_stats(double s, double q, long S, float eps, float& m, float& r) {
    const double mu = s / (double)S;
    double var = q / (double)S - mu * mu;
    var = var < 0.0 ? 0.0 : var;
    m = (float)mu;
    r = (float)(1.0 / sqrt(var + (double)eps));
}

// nchunk == 1: one launch does both passes.
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK) in_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ mean,
                                                        float* __restrict__ rstd, long S, float eps, float slope) {
    __shared__ double red[BLOCK / 64 + 1];
    __shared__ float stat[2];
    const long inst = blockIdx.x;
    const float* xi = x + inst * S;
    float* yi = y + inst * S;
    const Slice sl = slice_of(xi, yi, nullptr, S, 1);
    double s, q;
    fwd_sums<BLOCK>(xi, sl, s, q);
    s = block_sum(s, red, threadIdx.x, BLOCK);
    q = block_sum(q, red, threadIdx.x, BLOCK);
    if (threadIdx.x == 0) {
        fwd_stats(s, q, S, eps, stat[0], stat[1]);
        mean[inst] = stat[0];
        rstd[inst] = stat[1];
    }
    __syncthreads();
    fwd_apply<BLOCK>(xi, yi, sl, stat[0], stat[1], slope);
}

// nchunk > 1: partial sums per (instance, slice) ...
__global__ void __launch_bounds__(256) in_fwd_partial_kernel(const float* __restrict__ x, double* __restrict__ part, long S, int nchunk) {
    __shared__ double red[5];
    const long inst = blockIdx.x;
    const float* xi = x + inst * S;
    const Slice sl = slice_of(xi, nullptr, nullptr, S, nchunk);
    double s, q;
    fwd_sums<256>(xi, sl, s, q);
    s = block_sum(s, red, threadIdx.x, 256);
    q = block_sum(q, red, threadIdx.x, 256);
    if (threadIdx.x == 0) {
        part[(inst * nchunk + blockIdx.y) * 2 + 0] = s;
        part[(inst * nchunk + blockIdx.y) * 2 + 1] = q;
    }
}

// ... then every slice sums the instance's partials in index order and normalises its part.
__global__ void __launch_bounds__(256) in_fwd_apply_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ mean,
                                                           float* __restrict__ rstd, const double* __restrict__ part, long S, int nchunk,
                                                           float eps, float slope) {
    __shared__ float stat[2];
    const long inst = blockIdx.x;
    const float* xi = x + inst * S;
    float* yi = y + inst * S;
    if (threadIdx.x == 0) {
        double s = 0.0, q = 0.0;
        for (int c = 0; c < nchunk; ++c) { s += part[(inst * nchunk + c) * 2]; q += part[(inst * nchunk + c) * 2 + 1]; }
        fwd_stats(s, q, S, eps, stat[0], stat[1]);
        if (blockIdx.y == 0) { mean[inst] = stat[0]; rstd[inst] = stat[1]; }
    }
    __syncthreads();
    fwd_apply<256>(xi, yi, slice_of(xi, yi, nullptr, S, nchunk), stat[0], stat[1], slope);
}

template <int BLOCK>
__device__ __forceinline__ void bwd_sums(const float* xi, const float* gi, const Slice sl, float m, float r, float slope, double& sg,
                                         double& sgx) {
    sg = 0.0; sgx = 0.0;
    if (sl.vec) {
        const float4* x4 = reinterpret_cast<const float4*>(xi);
        const float4* g4 = reinterpret_cast<const float4*>(gi);
        for (long i = sl.lo / 4 + threadIdx.x; i < sl.hi / 4; i += BLOCK) {
            const float4 xv = x4[i], gv = g4[i];
            const float h0 = (xv.x - m) * r, h1 = (xv.y - m) * r, h2 = (xv.z - m) * r, h3 = (xv.w - m) * r;
            const float g0 = gv.x * (h0 > 0.f ? 1.f : slope), g1 = gv.y * (h1 > 0.f ? 1.f : slope);
            const float g2 = gv.z * (h2 > 0.f ? 1.f : slope), g3 = gv.w * (h3 > 0.f ? 1.f : slope);
            sg += ((double)g0 + (double)g1) + ((double)g2 + (double)g3);
            sgx += ((double)g0 * h0 + (double)g1 * h1) + ((double)g2 * h2 + (double)g3 * h3);
        }
    } else {
        for (long i = sl.lo + threadIdx.x; i < sl.hi; i += BLOCK) {
            const float h = (xi[i] - m) * r;
            const float g = gi[i] * (h > 0.f ? 1.f : slope);
            sg += g;
            sgx += (double)g * h;
        }
    }
}

template <int BLOCK>
__device__ __forceinline__ void bwd_apply(const float* xi, const float* gi, float* oi, const Slice sl, float m, float r, float slope,
                                          float mg, float mgx) {
    if (sl.vec) {
        const float4* x4 = reinterpret_cast<const float4*>(xi);
        const float4* g4 = reinterpret_cast<const float4*>(gi);
        float4* o4 = reinterpret_cast<float4*>(oi);
        for (long i = sl.lo / 4 + threadIdx.x; i < sl.hi / 4; i += BLOCK) {
            const float4 xv = x4[i], gv = g4[i];
            const float h0 = (xv.x - m) * r, h1 = (xv.y - m) * r, h2 = (xv.z - m) * r, h3 = (xv.w - m) * r;
            float4 o;
            o.x = r * (gv.x * (h0 > 0.f ? 1.f : slope) - mg - h0 * mgx);
            o.y = r * (gv.y * (h1 > 0.f ? 1.f : slope) - mg - h1 * mgx);
            o.z = r * (gv.z * (h2 > 0.f ? 1.f : slope) - mg - h2 * mgx);
            o.w = r * (gv.w * (h3 > 0.f ? 1.f : slope) - mg - h3 * mgx);
            o4[i] = o;
        }
    } else {
        for (long i = sl.lo + threadIdx.x; i < sl.hi; i += BLOCK) {
            const float h = (xi[i] - m) * r;
            oi[i] = r * (gi[i] * (h > 0.f ? 1.f : slope) - mg - h * mgx);
        }
    }
}

template <int BLOCK>
__global__ void __launch_bounds__(BLOCK) in_bwd_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, const float* __restrict__ gy, float* __restrict__ gx,
                                                        long S, float slope, long C, long gbs) {
    __shared__ double red[BLOCK / 64 + 1];
    __shared__ float stat[2];
    const long inst = blockIdx.x;
    const float *xi = x + inst * S, *gi = gy + (inst / C) * gbs + (inst % C) * S;      // grad_y may be a channel slice of a wider tensor
    float* oi = gx + inst * S;
    const float m = mean[inst], r = rstd[inst];
    const Slice sl = slice_of(xi, gi, oi, S, 1);
    double sg, sgx;
    bwd_sums<BLOCK>(xi, gi, sl, m, r, slope, sg, sgx);
    sg = block_sum(sg, red, threadIdx.x, BLOCK);
    sgx = block_sum(sgx, red, threadIdx.x, BLOCK);
    if (threadIdx.x == 0) { stat[0] = (float)(sg / (double)S); stat[1] = (float)(sgx / (double)S); }
    __syncthreads();
    bwd_apply<BLOCK>(xi, gi, oi, sl, m, r, slope, stat[0], stat[1]);
}

__global__ void __launch_bounds__(256) in_bwd_partial_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, const float* __restrict__ gy,
                                                             double* __restrict__ part, long S, int nchunk, float slope, long C, long gbs) {
    __shared__ double red[5];
    const long inst = blockIdx.x;
    const float *xi = x + inst * S, *gi = gy + (inst / C) * gbs + (inst % C) * S;
    double sg, sgx;
    bwd_sums<256>(xi, gi, slice_of(xi, gi, nullptr, S, nchunk), mean[inst], rstd[inst], slope, sg, sgx);
    sg = block_sum(sg, red, threadIdx.x, 256);
    sgx = block_sum(sgx, red, threadIdx.x, 256);
    if (threadIdx.x == 0) {
        part[(inst * nchunk + blockIdx.y) * 2 + 0] = sg;
        part[(inst * nchunk + blockIdx.y) * 2 + 1] = sgx;
    }
}

__global__ void __launch_bounds__(256) in_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const float* __restrict__ gy, float* __restrict__ gx,
                                                           const double* __restrict__ part, long S, int nchunk, float slope, long C, long gbs) {
    __shared__ float stat[2];
    const long inst = blockIdx.x;
    const float *xi = x + inst * S, *gi = gy + (inst / C) * gbs + (inst % C) * S;
    float* oi = gx + inst * S;
    if (threadIdx.x == 0) {
        double sg = 0.0, sgx = 0.0;
        for (int c = 0; c < nchunk; ++c) { sg += part[(inst * nchunk + c) * 2]; sgx += part[(inst * nchunk + c) * 2 + 1]; }
        stat[0] = (float)(sg / (double)S);
        stat[1] = (float)(sgx / (double)S);
    }
    __syncthreads();
    bwd_apply<256>(xi, gi, oi, slice_of(xi, gi, oi, S, nchunk), mean[inst], rstd[inst], slope, stat[0], stat[1]);
}

// Slices per instance: enough workgroups for ~8 per CU, each slice at least IN_MINCHUNK elements.
static int plan_chunks(int64_t n_inst, int64_t S) {
    if (n_inst >= 2048 || S < 2 * IN_MINCHUNK) return 1;
    int64_t want = (2048 + n_inst - 1) / n_inst;
    const int64_t most = S / IN_MINCHUNK;
    if (want > most) want = most;
    if (want > IN_MAXCHUNK) want = IN_MAXCHUNK;
    return want < 1 ? 1 : (int)want;
}

}  // namespace dn

extern "C" int64_t dn_instnorm_workspace_bytes(int64_t n_inst, int64_t spatial) {
    if (n_inst < 1 || spatial < 1) return DN_E_BADARG;
    const int nchunk = dn::plan_chunks(n_inst, spatial);
    return nchunk > 1 ? n_inst * nchunk * 2 * (int64_t)sizeof(double) : 0;
}

static int in_check(int64_t n_inst, int64_t spatial, void* workspace, int64_t workspace_bytes, int& nchunk) {
    if (n_inst < 1 || spatial < 1 || n_inst > 2147483647ll) return DN_E_BADARG;
    nchunk = dn::plan_chunks(n_inst, spatial);
    if (nchunk > 1 && (!workspace || workspace_bytes < n_inst * nchunk * 2 * (int64_t)sizeof(double))) return DN_E_WORKSPACE;
    return 0;
}

extern "C" int dn_instnorm_act_fwd(const float* x, float* y, float* mean, float* rstd, int64_t n_inst, int64_t spatial, float eps,
                                   float slope, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!x || !y || !mean || !rstd) return DN_E_BADARG;
    int nchunk;
    if (int rc = in_check(n_inst, spatial, workspace, workspace_bytes, nchunk)) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const long S = (long)spatial;
    if (nchunk == 1) {
        if (S <= 1024)
            hipLaunchKernelGGL(dn::in_fwd_kernel<64>, dim3((unsigned)n_inst), dim3(64), 0, st, x, y, mean, rstd, S, eps, slope);
        else
            hipLaunchKernelGGL(dn::in_fwd_kernel<256>, dim3((unsigned)n_inst), dim3(256), 0, st, x, y, mean, rstd, S, eps, slope);
    } else {
        double* part = static_cast<double*>(workspace);
        hipLaunchKernelGGL(dn::in_fwd_partial_kernel, dim3((unsigned)n_inst, nchunk), dim3(256), 0, st, x, part, S, nchunk);
        hipLaunchKernelGGL(dn::in_fwd_apply_kernel, dim3((unsigned)n_inst, nchunk), dim3(256), 0, st, x, y, mean, rstd, part, S, nchunk, eps,
                           slope);
    }
    DN_LAUNCH_CHECK();
    return 0;
}

extern "C" int dn_instnorm_act_bwd(const float* x, const float* mean, const float* rstd, const float* grad_y, float* grad_x,
                                   int64_t n_inst, int64_t spatial, float slope, int64_t channels, int64_t grad_y_batch_stride,
                                   void* workspace, int64_t workspace_bytes, void* stream) {
    if (!x || !mean || !rstd || !grad_y || !grad_x) return DN_E_BADARG;
    if (channels < 1 || n_inst % channels != 0) return DN_E_BADARG;
    const long C = (long)channels, gbs = grad_y_batch_stride > 0 ? (long)grad_y_batch_stride : (long)(channels * spatial);
    if (gbs < channels * spatial) return DN_E_BADARG;
    int nchunk;
    if (int rc = in_check(n_inst, spatial, workspace, workspace_bytes, nchunk)) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const long S = (long)spatial;
    if (nchunk == 1) {
        if (S <= 1024)
            hipLaunchKernelGGL(dn::in_bwd_kernel<64>, dim3((unsigned)n_inst), dim3(64), 0, st, x, mean, rstd, grad_y, grad_x, S, slope, C, gbs);
        else
            hipLaunchKernelGGL(dn::in_bwd_kernel<256>, dim3((unsigned)n_inst), dim3(256), 0, st, x, mean, rstd, grad_y, grad_x, S, slope, C, gbs);
    } else {
        double* part = static_cast<double*>(workspace);
        hipLaunchKernelGGL(dn::in_bwd_partial_kernel, dim3((unsigned)n_inst, nchunk), dim3(256), 0, st, x, mean, rstd, grad_y, part, S, nchunk,
                           slope, C, gbs);
        hipLaunchKernelGGL(dn::in_bwd_apply_kernel, dim3((unsigned)n_inst, nchunk), dim3(256), 0, st, x, mean, rstd, grad_y, grad_x, part, S,
                           nchunk, slope, C, gbs);
    }
    DN_LAUNCH_CHECK();
    return 0;
}
