// DiffNetFDM stencil derivatives (reference DiffNet/DiffNetFDM.py:6-60 kernels, :63-119 boundary
// correction matrices, :158-199 operators).  The reference computes conv2d(g_padded, 3x3 kernel) followed by a dense
// N x N "correction" matmul that only rewrites the two boundary columns (or rows):
//     d[edge] <- a * d[edge] + b * d[edge -/+ 1]      (a,b) = (4,-1) first derivative, (0,1) second derivative.
// Here stencil and boundary fix-up are one pass; the N x N matmul disappears.  The adjoint is in gather form.
//   g: (B,1,ny+2,nx+2) replicate-padded input, out: (B,1,ny,nx);  axis 0: fix columns x = 0, nx-1; axis 1: rows.
#include <algorithm>

#include "dn_common.h"

namespace dn {

struct FdmParams {
    float k[9];      // 3x3 correlation kernel, row-major (y, x)
    float a, b;
    int axis, ny, nx;
};

__device__ __forceinline__ float fdm_conv(const float* __restrict__ g, const FdmParams& p, int j, int i) {
    const int W = p.nx + 2;
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) s = fmaf(p.k[r * 3 + c], g[(size_t)(j + r) * W + i + c], s);
    return s;
}

__global__ void __launch_bounds__(256) fdm_fwd_kernel(const float* __restrict__ g, float* __restrict__ out, const FdmParams p, int batch) {
    const size_t n = (size_t)p.ny * p.nx;
    const size_t total = n * batch;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(idx / n);
        const int j = (int)((idx - (size_t)b * n) / p.nx), i = (int)(idx % p.nx);
        const float* gb = g + (size_t)b * (p.ny + 2) * (p.nx + 2);
        float d = fdm_conv(gb, p, j, i);
        const int pos = p.axis == 0 ? i : j, last = (p.axis == 0 ? p.nx : p.ny) - 1;
        if (last >= 1 && (pos == 0 || pos == last)) {
            const int q = pos == 0 ? 1 : last - 1;
            const float dn = p.axis == 0 ? fdm_conv(gb, p, j, q) : fdm_conv(gb, p, q, i);
            d = p.a * d + p.b * dn;
        }
        out[idx] = d;
    }
}

// cotangent of the un-corrected conv output at (j, i): undo the boundary combination
__device__ __forceinline__ float fdm_gd(const float* __restrict__ go, const FdmParams& p, int j, int i) {
    if (j < 0 || j >= p.ny || i < 0 || i >= p.nx) return 0.f;
    const int pos = p.axis == 0 ? i : j, last = (p.axis == 0 ? p.nx : p.ny) - 1;
    float v = go[(size_t)j * p.nx + i];
    if (last < 1) return v;
    if (pos == 0 || pos == last) v *= p.a;
    if (pos == 1) v += p.b * (p.axis == 0 ? go[(size_t)j * p.nx + 0] : go[(size_t)0 * p.nx + i]);
    if (pos == last - 1) v += p.b * (p.axis == 0 ? go[(size_t)j * p.nx + last] : go[(size_t)last * p.nx + i]);
    return v;
}

__global__ void __launch_bounds__(256) fdm_bwd_kernel(const float* __restrict__ go, float* __restrict__ gg, const FdmParams p, int batch) {
    const int W = p.nx + 2, H = p.ny + 2;
    const size_t n = (size_t)H * W;
    const size_t total = n * batch;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(idx / n);
        const int jj = (int)((idx - (size_t)b * n) / W), ii = (int)(idx % W);
        const float* gob = go + (size_t)b * p.ny * p.nx;
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) s = fmaf(p.k[r * 3 + c], fdm_gd(gob, p, jj - r, ii - c), s);
        gg[idx] = s;
    }
}

// ---- fused form: replicate padding folded into the stencil (indices clamped), input and output on the same ny x nx grid -------
__device__ __forceinline__ float fdm_conv_clamped(const float* __restrict__ u, const FdmParams& p, int j, int i) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int jj = min(max(j + r - 1, 0), p.ny - 1);
#pragma unroll
        for (int c = 0; c < 3; ++c) s = fmaf(p.k[r * 3 + c], u[(size_t)jj * p.nx + min(max(i + c - 1, 0), p.nx - 1)], s);
    }
    return s;
}

__global__ void __launch_bounds__(256) fdm_fused_fwd_kernel(const float* __restrict__ u, float* __restrict__ out, const FdmParams p, int batch) {
    const size_t n = (size_t)p.ny * p.nx;
    const size_t total = n * batch;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(idx / n);
        const int j = (int)((idx - (size_t)b * n) / p.nx), i = (int)(idx % p.nx);
        const float* ub = u + (size_t)b * n;
        float d = fdm_conv_clamped(ub, p, j, i);
        const int pos = p.axis == 0 ? i : j, last = (p.axis == 0 ? p.nx : p.ny) - 1;
        if (last >= 1 && (pos == 0 || pos == last)) {
            const int q = pos == 0 ? 1 : last - 1;
            const float dn = p.axis == 0 ? fdm_conv_clamped(ub, p, j, q) : fdm_conv_clamped(ub, p, q, i);
            d = p.a * d + p.b * dn;
        }
        out[idx] = d;
    }
}

// cotangent of padded cell (jj, ii) of the (ny + 2) x (nx + 2) replicate-padded grid (the body of fdm_bwd_kernel)
__device__ __forceinline__ float fdm_gpad(const float* __restrict__ gob, const FdmParams& p, int jj, int ii) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) s = fmaf(p.k[r * 3 + c], fdm_gd(gob, p, jj - r, ii - c), s);
    return s;
}

// adjoint of pad + stencil + fix-up: node (j, i) collects the padded cells that replicate it (1, 2 or 4 of them), fixed order
__global__ void __launch_bounds__(256) fdm_fused_bwd_kernel(const float* __restrict__ go, float* __restrict__ gu, const FdmParams p, int batch) {
    const size_t n = (size_t)p.ny * p.nx;
    const size_t total = n * batch;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(idx / n);
        const int j = (int)((idx - (size_t)b * n) / p.nx), i = (int)(idx % p.nx);
        const float* gob = go + (size_t)b * n;
        const int j0 = j == 0 ? 0 : j + 1, j1 = j == p.ny - 1 ? p.ny + 1 : j + 1;
        const int i0 = i == 0 ? 0 : i + 1, i1 = i == p.nx - 1 ? p.nx + 1 : i + 1;
        float s = 0.f;
        for (int jj = j0; jj <= j1; ++jj)
            for (int ii = i0; ii <= i1; ++ii) s += fdm_gpad(gob, p, jj, ii);
        gu[idx] = s;
    }
}

static int fdm_params(FdmParams& p, const float* k9, int axis, float a, float b, int ny, int nx) {
    if (!k9 || (axis != 0 && axis != 1) || ny < 1 || nx < 1) return DN_E_BADARG;
    for (int i = 0; i < 9; ++i) p.k[i] = k9[i];
    p.a = a; p.b = b; p.axis = axis; p.ny = ny; p.nx = nx;
    return 0;
}

}  // namespace dn

using namespace dn;

extern "C" int dn_fdm_stencil_fwd(const float* g_padded, float* out, int32_t batch, int32_t ny, int32_t nx, const float* kernel9,
                                  int32_t axis, float a, float b, void* stream) {
    FdmParams p;
    int rc = fdm_params(p, kernel9, axis, a, b, ny, nx);
    if (rc) return rc;
    if (!g_padded || !out || batch < 1) return DN_E_BADARG;
    const size_t total = (size_t)ny * nx * batch;
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(fdm_fwd_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), g_padded, out, p, batch);
    DN_LAUNCH_CHECK();
    return 0;
}

extern "C" int dn_fdm_stencil_bwd(const float* grad_out, float* grad_g_padded, int32_t batch, int32_t ny, int32_t nx,
                                  const float* kernel9, int32_t axis, float a, float b, void* stream) {
    FdmParams p;
    int rc = fdm_params(p, kernel9, axis, a, b, ny, nx);
    if (rc) return rc;
    if (!grad_out || !grad_g_padded || batch < 1) return DN_E_BADARG;
    const size_t total = (size_t)(ny + 2) * (nx + 2) * batch;
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(fdm_bwd_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), grad_out, grad_g_padded, p, batch);
    DN_LAUNCH_CHECK();
    return 0;
}

extern "C" int dn_fdm_fused_fwd(const float* u, float* out, int32_t batch, int32_t ny, int32_t nx, const float* kernel9, int32_t axis, float a,
                                float b, void* stream) {
    FdmParams p;
    int rc = fdm_params(p, kernel9, axis, a, b, ny, nx);
    if (rc) return rc;
    if (!u || !out || batch < 1) return DN_E_BADARG;
    const size_t total = (size_t)ny * nx * batch;
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(fdm_fused_fwd_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), u, out, p, batch);
    DN_LAUNCH_CHECK();
    return 0;
}

extern "C" int dn_fdm_fused_bwd(const float* grad_out, float* grad_u, int32_t batch, int32_t ny, int32_t nx, const float* kernel9, int32_t axis,
                                float a, float b, void* stream) {
    FdmParams p;
    int rc = fdm_params(p, kernel9, axis, a, b, ny, nx);
    if (rc) return rc;
    if (!grad_out || !grad_u || batch < 1) return DN_E_BADARG;
    const size_t total = (size_t)ny * nx * batch;
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(fdm_fused_bwd_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), grad_out, grad_u, p, batch);
    DN_LAUNCH_CHECK();
    return 0;
}
