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

// Tuning switches (dn_config_set / DN_<KEY> at load time, dn_api.hip): value of a switch, nullptr when unset.  A plain table
// lookup -- nothing on the launch path touches the process environment.
enum ConfigKey : int { CFG_PLAN2D = 0, CFG_PLAN3D, CFG_PLAN_FSDT, CFG_Q1_RULE_KERNEL, CFG_GPE_GATHER, CFG_GPE_TILED, CFG_Q1_3D_T16, CFG_Q1_3D_E1SUM, CFG_FSDT_GENERIC, CFG_Q1_3D_E1, CFG_HANDOVER_SPIN_LIMIT, CFG_CONV2D_V1, CFG_CONV_WRW_WGS, CFG_Q1_3D_N2, CFG_FSDT_FORM, CFG_COUNT };
const char* config(ConfigKey k);

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

// Two block sums in one pass (one LDS exchange, two barriers instead of four); results valid in thread 0.  `scratch` holds
// >= 2 * blockDim / 64 doubles.  Same order of additions as two block_sum calls.
__device__ __forceinline__ void block_sum2(double& a, double& b, double* scratch, int tid, int nthreads) {
    a = wave_sum(a);
    b = wave_sum(b);
    const int lane = tid & (DN_WAVE - 1), wave = tid / DN_WAVE;
    const int nw = (nthreads + DN_WAVE - 1) / DN_WAVE;
    if (lane == 0) { scratch[wave] = a; scratch[nw + wave] = b; }
    __syncthreads();
    double ra = 0.0, rb = 0.0;
    if (tid == 0) {
        for (int w = 0; w < nw; ++w) { ra += scratch[w]; rb += scratch[nw + w]; }
    }
    __syncthreads();
    a = ra; b = rb;
}

template <int N>
struct VecT;
template <>
struct VecT<1> { using type = float; };
template <>
struct VecT<2> { using type = float2; };
template <>
struct VecT<4> { using type = float4; };

// Typed access at a 32-bit BYTE offset from a wave-uniform base pointer.  Spelling the address as base + zext(u32 bytes)
// lets the compiler pick `global_load/store ... v_off, s[base:base+1]` (SGPR base + 32-bit VGPR offset); indexing
// `base[u32_index]` does not, because index * sizeof(T) may exceed 32 bits, and costs a 64-bit VALU add per access.
// Callers guarantee in-sample byte offsets < 2^32 (dn_poisson_apply rejects samples of >= 2^30 nodes).
#ifndef DN_NT_STORES
#define DN_NT_STORES 0     // per translation unit: 1 in the 2-D closed-form kernel (poisson2d_q1_cf.hip); measured neutral to 3 % slower in the 3-D kernel
#endif
template <typename V, typename T>
__device__ __forceinline__ V ld_at(const T* __restrict__ base, unsigned index) {
#ifdef DN_NT_LOADS          // experiment: non-temporal loads for the streamed fields (profiles/r2_ab2d_nt.txt)
    const char* a = reinterpret_cast<const char*>(base) + (index * (unsigned)sizeof(T));
    if constexpr (sizeof(V) == 16) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(a));
        return __builtin_bit_cast(V, t);
    } else if constexpr (sizeof(V) == 8) {
        typedef float v2f __attribute__((ext_vector_type(2)));
        const v2f t = __builtin_nontemporal_load(reinterpret_cast<const v2f*>(a));
        return __builtin_bit_cast(V, t);
    } else {
        return __builtin_nontemporal_load(reinterpret_cast<const V*>(a));
    }
#else
    return *reinterpret_cast<const V*>(reinterpret_cast<const char*>(base) + (index * (unsigned)sizeof(T)));
#endif
}
template <typename V, typename T>
__device__ __forceinline__ void st_at(T* __restrict__ base, unsigned index, const V& v) {
#if DN_NT_STORES            // the operators' outputs are written once and not read again by the launch: non-temporal stores (2-D bench
                            // kernel 54.3 -> 52.8 us, profiles/r2_ab2d_nt.txt; non-temporal LOADS are 20 % slower: halo rows and shared nodes are re-read)
    char* a = reinterpret_cast<char*>(base) + (index * (unsigned)sizeof(T));
    if constexpr (sizeof(V) == 16) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(__builtin_bit_cast(v4f, v), reinterpret_cast<v4f*>(a));
    } else if constexpr (sizeof(V) == 8) {
        typedef float v2f __attribute__((ext_vector_type(2)));
        __builtin_nontemporal_store(__builtin_bit_cast(v2f, v), reinterpret_cast<v2f*>(a));
    } else {
        __builtin_nontemporal_store(v, reinterpret_cast<V*>(a));
    }
#else
    *reinterpret_cast<V*>(reinterpret_cast<char*>(base) + (index * (unsigned)sizeof(T))) = v;
#endif
}

// Branch-free row-segment load: NW consecutive nodes starting at x0 plus the node x0+NW shared with the next
// thread, from a per-sample base pointer (wave-uniform => SGPR) and a 32-bit in-sample offset (=> the
// `global_load saddr + voffset` form, no 64-bit VALU address arithmetic).  Indices are clamped into [0, nx)
// so every lane issues the same instructions (no exec-masked regions, no wait between loads); clamped
// duplicates only ever feed elements that are skipped.
// VEC: nx % NW == 0 and NW-element aligned rows (checked by the host), so the vector access is aligned.
template <int NW, bool VEC, typename T>
__device__ __forceinline__ void load_seg(const T* __restrict__ base, unsigned rowoff, int x0, int nx, T (&dst)[NW + 1]) {
    if constexpr (VEC && (NW == 2 || NW == 4)) {
        const unsigned xl = (unsigned)min(x0, nx - NW);
        if constexpr (sizeof(T) == 4) {
            using V = typename VecT<NW>::type;
            const V v = ld_at<V>(base, rowoff + xl);
            const T* vf = reinterpret_cast<const T*>(&v);
#pragma unroll
            for (int k = 0; k < NW; ++k) dst[k] = vf[k];
        } else {
            static_assert(sizeof(T) == 1, "load_seg: 1- or 4-byte elements");
            if constexpr (NW == 4) {
                const uint32_t w = ld_at<uint32_t>(base, rowoff + xl);
#pragma unroll
                for (int k = 0; k < 4; ++k) dst[k] = (T)((w >> (8 * k)) & 0xffu);
            } else {
                const uint16_t w = ld_at<uint16_t>(base, rowoff + xl);
                dst[0] = (T)(w & 0xffu);
                dst[1] = (T)(w >> 8);
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < NW; ++k) dst[k] = ld_at<T>(base, rowoff + (unsigned)min(x0 + k, nx - 1));
    }
    dst[NW] = ld_at<T>(base, rowoff + (unsigned)min(x0 + NW, nx - 1));
}

// load_seg for a field that is streamed once (coefficients): the aligned vector part as a NON-TEMPORAL load, the shared node as usual
template <int NW, bool VEC, typename T>
__device__ __forceinline__ void load_seg_stream(const T* __restrict__ base, unsigned rowoff, int x0, int nx, T (&dst)[NW + 1]) {
    if constexpr (VEC && NW == 4 && sizeof(T) == 4) {
        typedef T v4t __attribute__((ext_vector_type(4)));
        const unsigned xl = (unsigned)min(x0, nx - NW);
        const v4t v = __builtin_nontemporal_load(reinterpret_cast<const v4t*>(reinterpret_cast<const char*>(base) + (rowoff + xl) * 4u));
        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
        dst[NW] = ld_at<T>(base, rowoff + (unsigned)min(x0 + NW, nx - 1));
    } else if constexpr (VEC && NW == 4 && sizeof(T) == 1) {       // byte masks: the four own nodes as one non-temporal dword
        const unsigned xl = (unsigned)min(x0, nx - NW);
        const uint32_t w = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(base) + (rowoff + xl)));
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[k] = (T)((w >> (8 * k)) & 0xffu);
        dst[NW] = ld_at<T>(base, rowoff + (unsigned)min(x0 + NW, nx - 1));
    } else {
        load_seg<NW, VEC>(base, rowoff, x0, nx, dst);
    }
}

// lane l <- lane l + 1 (whole wave), lane 63 keeps `last`
__device__ __forceinline__ unsigned dpp_from_right_u32(unsigned v, unsigned last) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)last, (int)v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
}

// load_seg for NW = 4 aligned rows where the shared node x0 + 4 is the RIGHT NEIGHBOUR LANE's first node: one vector load per
// lane; the extra value comes over DPP (wave_shl:1) and only lane 63 of each wave loads it from memory (its neighbour is in
// the next wave).  A per-lane dword / byte load at a stride of 16 bytes touches 16 cache lines per wave-instruction; this form
// issues it for one lane.  Requires consecutive lanes = consecutive x0 (x0 = 4 * lane + const within the wave).
template <typename T>
__device__ __forceinline__ void load_seg4_dpp(const T* __restrict__ base, unsigned rowoff, int x0, int nx, T (&dst)[5]) {
    const unsigned xl = (unsigned)min(x0, nx - 4);
    // the node right of the wave's last lane: ONE wave-uniform address -> a scalar (SMEM) load, no divergent branch
    const unsigned eidx = (unsigned)__builtin_amdgcn_readfirstlane((int)rowoff) +
                          (unsigned)min(__builtin_amdgcn_readfirstlane(x0) + 256, nx - 1);
    if constexpr (sizeof(T) == 4) {
        const float4 v = ld_at<float4>(base, rowoff + xl);
        const unsigned bits = __float_as_uint(reinterpret_cast<const float&>(v.x));
        const T lastv = base[eidx];
        const float last = reinterpret_cast<const float&>(lastv);
        const unsigned e = dpp_from_right_u32(bits, __float_as_uint(last));
        const float vf[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[k] = reinterpret_cast<const T&>(vf[k]);
        const float ef = __uint_as_float(e);
        dst[4] = reinterpret_cast<const T&>(ef);
    } else {
        static_assert(sizeof(T) == 1, "load_seg4_dpp: 1- or 4-byte elements");
        const uint32_t w = ld_at<uint32_t>(base, rowoff + xl);
        const uint32_t last = (uint32_t)base[eidx];
        const uint32_t e = dpp_from_right_u32(w, last);
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[k] = (T)((w >> (8 * k)) & 0xffu);
        dst[4] = (T)(e & 0xffu);
    }
}

// load_seg4_dpp with the lane exchange over ds_bpermute (__shfl_down: 2-3 cycles of a shared SIMD where a DPP move costs 30-36, see
// DESIGN.md section 4) and, optionally, the 16-byte row vector as a NON-TEMPORAL load.  In this form every byte of a row is requested
// once per wave (plus one dword through the scalar cache for the wave's last lane): the precondition for non-temporal loads to pay
// (tools/march_probe.py, profiles/r3_march_probe.txt: the access pattern alone 51.4 -> 47.9 us once nothing is re-read).
template <bool NT, typename T>
__device__ __forceinline__ void load_seg4_shfl(const T* __restrict__ base, unsigned rowoff, int x0, int nx, T (&dst)[5]) {
    const unsigned xl = (unsigned)min(x0, nx - 4);
    const unsigned eidx = (unsigned)__builtin_amdgcn_readfirstlane((int)rowoff) +
                          (unsigned)min(__builtin_amdgcn_readfirstlane(x0) + 256, nx - 1);
    const bool lastlane = (threadIdx.x & 63u) == 63u;
    if constexpr (sizeof(T) == 4) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        const v4f* a = reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(base) + (rowoff + xl) * 4u);
        const v4f v = NT ? __builtin_nontemporal_load(a) : *a;
        const T lastv = base[eidx];
        float e = __shfl_down(v.x, 1, 64);
        // the exchange must run with every lane active: pinned here, or the compiler sinks it into the `!lastlane` side of the select
        // below, where lane 63 -- the source of lane 62 -- is masked off and ds_bpermute returns 0 for it
        asm volatile("" : "+v"(e));
        const float vf[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[k] = reinterpret_cast<const T&>(vf[k]);
        const float ef = lastlane ? reinterpret_cast<const float&>(lastv) : e;
        dst[4] = reinterpret_cast<const T&>(ef);
    } else {
        static_assert(sizeof(T) == 1, "load_seg4_shfl: 1- or 4-byte elements");
        const uint32_t w = ld_at<uint32_t>(base, rowoff + xl);
        const uint32_t last = (uint32_t)base[eidx];
        uint32_t sh = (uint32_t)__shfl_down((int)w, 1, 64);
        asm volatile("" : "+v"(sh));          // as above: the exchange stays outside the divergent select
        const uint32_t e = lastlane ? last : sh;
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[k] = (T)((w >> (8 * k)) & 0xffu);
        dst[4] = (T)(e & 0xffu);
    }
}

// Exactly NW consecutive nodes starting at x0 (clamped like load_seg, no shared +1 node).
template <int NW, bool VEC, typename T>
__device__ __forceinline__ void load_own(const T* __restrict__ base, unsigned rowoff, int x0, int nx, T (&dst)[NW]) {
    if constexpr (VEC && (NW == 2 || NW == 4)) {
        const unsigned xl = (unsigned)min(x0, nx - NW);
        if constexpr (sizeof(T) == 4) {
            using V = typename VecT<NW>::type;
            const V v = ld_at<V>(base, rowoff + xl);
            const T* vf = reinterpret_cast<const T*>(&v);
#pragma unroll
            for (int k = 0; k < NW; ++k) dst[k] = vf[k];
        } else {
            if constexpr (NW == 4) {
                const uint32_t w = ld_at<uint32_t>(base, rowoff + xl);
#pragma unroll
                for (int k = 0; k < 4; ++k) dst[k] = (T)((w >> (8 * k)) & 0xffu);
            } else {
                const uint16_t w = ld_at<uint16_t>(base, rowoff + xl);
                dst[0] = (T)(w & 0xffu);
                dst[1] = (T)(w >> 8);
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < NW; ++k) dst[k] = ld_at<T>(base, rowoff + (unsigned)min(x0 + k, nx - 1));
    }
}

// Store NW consecutive floats at x0 (entries >= nx dropped).  VEC as above.
template <int NW, bool VEC>
__device__ __forceinline__ void store_seg(float* __restrict__ base, unsigned rowoff, int x0, int nx, const float (&src)[NW]) {
    if constexpr (VEC && (NW == 2 || NW == 4)) {
        if (x0 + NW <= nx) {                          // (== x0 < nx where rows are a multiple of NW nodes wide; rows of 4 k + 1 nodes: the thread column past the last full one stores nothing)
            using V = typename VecT<NW>::type;
            V v;
            float* vf = reinterpret_cast<float*>(&v);
#pragma unroll
            for (int k = 0; k < NW; ++k) vf[k] = src[k];
            st_at<V>(base, rowoff + (unsigned)x0, v);
        }
    } else {
#pragma unroll
        for (int k = 0; k < NW; ++k)
            if (x0 + k < nx) st_at<float>(base, rowoff + (unsigned)(x0 + k), src[k]);
    }
}

}  // namespace dn
