// Shared device helpers for libdiffnet_hip (gfx950 / CDNA4 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/diffnet_hip.h"

#define DN_WAVE 64

#define DN_LAUNCH_CHECK()                         \
    do {                                          \
        hipError_t e__ = hipGetLastError();       \
        if (e__ != hipSuccess) return (int)e__;   \
    } while (0)

namespace dn {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = DN_WAVE / 2; off > 0; off >>= 1) v += __shfl_down(v, off, DN_WAVE);
    return v;
}

// Deterministic block sum of a double; result valid in thread 0.  `scratch` holds >= blockDim/64 doubles.
__device__ __forceinline__ double block_sum(double v, double* scratch, int tid, int nthreads) {
    v = wave_sum(v);
    const int lane = tid & (DN_WAVE - 1), wave = tid / DN_WAVE;
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    double r = 0.0;
    if (tid == 0) {
        const int nw = (nthreads + DN_WAVE - 1) / DN_WAVE;
        for (int w = 0; w < nw; ++w) r += scratch[w];
    }
    __syncthreads();
    return r;
}

template <int N>
struct VecT;
template <>
struct VecT<1> { using type = float; };
template <>
struct VecT<2> { using type = float2; };
template <>
struct VecT<4> { using type = float4; };

// Load N consecutive floats starting at p[i0]; entries with index >= limit read as `fill`.
// VEC: the caller guarantees 4*N-byte alignment of &p[i0] (vector load when fully in range).
template <int N, bool VEC>
__device__ __forceinline__ void load_run(const float* __restrict__ p, int64_t base, int i0, int limit, float fill,
                                         float (&dst)[N]) {
    if constexpr (VEC && (N == 2 || N == 4)) {
        if (i0 + N <= limit) {
            using V = typename VecT<N>::type;
            const V v = *reinterpret_cast<const V*>(p + base + i0);
            if constexpr (N == 2) { dst[0] = v.x; dst[1] = v.y; }
            else { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w; }
            return;
        }
    }
#pragma unroll
    for (int k = 0; k < N; ++k) dst[k] = (i0 + k < limit) ? p[base + i0 + k] : fill;
}

template <int N, bool VEC>
__device__ __forceinline__ void store_run(float* __restrict__ p, int64_t base, int i0, int limit, const float (&src)[N]) {
    if constexpr (VEC && (N == 2 || N == 4)) {
        if (i0 + N <= limit) {
            using V = typename VecT<N>::type;
            V v;
            if constexpr (N == 2) { v.x = src[0]; v.y = src[1]; }
            else { v.x = src[0]; v.y = src[1]; v.z = src[2]; v.w = src[3]; }
            *reinterpret_cast<V*>(p + base + i0) = v;
            return;
        }
    }
#pragma unroll
    for (int k = 0; k < N; ++k)
        if (i0 + k < limit) p[base + i0 + k] = src[k];
}

}  // namespace dn
