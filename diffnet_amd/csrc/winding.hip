// Winding-number inside/outside field of the IBN scripts:
// reference `compute_winding_nodes`, IBN/poisson-2d/parametric/IBN_2D.py:89-104 -- a Python list comprehension over
// every grid column, each a (B, Npts, Ny) broadcast reduction.  Here: one thread per grid node, the boundary
// points and normals staged through LDS in chunks and broadcast-read by the whole workgroup; O(B * Nnodes * Npts)
// flops, reads points once per workgroup, writes each node once.
//   out[b, 0, ix, iy] = sum_p ((p - q) . n_p) / (4 pi (|p_x - q_x| + |p_y - q_y|))^3 ,  q = nodes[:, iy, ix]
// (the reference's denominator is the L1 distance, its `area` argument is unused, and its output is (Nx, Ny)-ordered:
// reproduced as is).
#include "dn_common.h"

namespace dn {

__global__ void __launch_bounds__(256) winding_kernel(const float* __restrict__ pts, const float* __restrict__ nrm,
                                                      const float* __restrict__ nodes, float* __restrict__ out, int npts, int ny,
                                                      int nx) {
    __shared__ float4 sp[256];                     // (px, py, nx, ny) per boundary point
    const int b = blockIdx.y;
    const int o = blockIdx.x * blockDim.x + threadIdx.x;      // flattened (ix, iy) output index
    const int nn = nx * ny;
    const int ix = o / ny, iy = o - ix * ny;
    float qx = 0.f, qy = 0.f;
    if (o < nn) {
        qx = nodes[(size_t)iy * nx + ix];
        qy = nodes[(size_t)nn + (size_t)iy * nx + ix];
    }
    const float k = 4.0f * 3.14159265358979323846f;
    float acc = 0.f;
    for (int p0 = 0; p0 < npts; p0 += 256) {
        const int p = p0 + threadIdx.x;
        if (p < npts) {
            const size_t i = ((size_t)b * npts + p) * 2;
            sp[threadIdx.x] = make_float4(pts[i], pts[i + 1], nrm[i], nrm[i + 1]);
        }
        __syncthreads();
        const int cnt = min(256, npts - p0);
        for (int j = 0; j < cnt; ++j) {
            const float4 v = sp[j];
            const float dx = v.x - qx, dy = v.y - qy;
            const float den = k * (fabsf(dx) + fabsf(dy));
            // v_rcp_f32 (1 ulp) instead of the IEEE division sequence (~10 instructions per pair: 2.2x the kernel time)
            acc = fmaf(fmaf(dx, v.z, dy * v.w), __builtin_amdgcn_rcpf(den * den * den), acc);
        }
        __syncthreads();
    }
    if (o < nn) out[(size_t)b * nn + o] = acc;
}

}  // namespace dn

extern "C" int dn_winding_nodes(const float* points, const float* normals, const float* nodes, float* out, int32_t batch,
                                int32_t npts, int32_t ny, int32_t nx, void* stream) {
    if (!points || !normals || !nodes || !out || batch < 1 || npts < 1 || ny < 1 || nx < 1 || batch > 65535) return DN_E_BADARG;
    const int nn = nx * ny;
    hipLaunchKernelGGL(dn::winding_kernel, dim3((nn + 255) / 256, batch), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), points,
                       normals, nodes, out, npts, ny, nx);
    DN_LAUNCH_CHECK();
    return 0;
}
