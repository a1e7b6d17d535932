// Micro-benchmark: what one vector-memory / LDS instruction costs in SIMD issue time when it sits in a VALU-bound stream (the
// situation of the 3-D FEM kernel).  Body = the generated ~240-instruction VALU DAG (valu_dag_body.h) + NM memory instructions of
// kind K per iteration; values loaded in iteration i are consumed in iteration i + 1 (no latency exposure).  All global accesses
// hit L1/L2 (a 64 KB array).  Reported: extra SIMD-cycles per memory instruction vs K = 0, at 1 / 2 / 5 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "valu_dag_body.h"
#ifndef NLOOP
#define NLOOP NI
#endif
#define NM 8

template <int K>
__global__ void __launch_bounds__(256) k(float* out, const float* __restrict__ src, float* __restrict__ dst, int iters, float c0, float c1, float c2,
                                         float c3) {
    __shared__ float lds[4096];
    float r[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) r[i] = threadIdx.x * 0.001f + i;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i;
    __syncthreads();
    float pend[NM * 4], acc[NM];
#pragma unroll
    for (int i = 0; i < NM; ++i) acc[i] = 0.f;
#pragma unroll
    for (int i = 0; i < NM * 4; ++i) pend[i] = 0.f;
    const unsigned lane_off = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        // consume what the previous iteration requested
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            acc[m] += pend[4 * m];
            if (K == 2 || K == 4) acc[m] += pend[4 * m + 1] + pend[4 * m + 2] + pend[4 * m + 3];
            if (K == 7) acc[m] += pend[4 * m + 1];
        }
        const unsigned base = (unsigned)(it & 15) * 1024u;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const unsigned idx = base + (unsigned)m * 256u + lane_off;        // floats
            if (K == 1) pend[4 * m] = src[idx & 16383u];
            if (K == 2) { const float4 v = reinterpret_cast<const float4*>(src)[(idx & 4095u)]; pend[4 * m] = v.x; pend[4 * m + 1] = v.y; pend[4 * m + 2] = v.z; pend[4 * m + 3] = v.w; }
            if (K == 3) pend[4 * m] = lds[idx & 4095u];
            if (K == 4) { const float4 v = reinterpret_cast<const float4*>(lds)[idx & 1023u]; pend[4 * m] = v.x; pend[4 * m + 1] = v.y; pend[4 * m + 2] = v.z; pend[4 * m + 3] = v.w; }
            if (K == 5) lds[idx & 4095u] = acc[m] + r[m];
            if (K == 6) dst[(size_t)blockIdx.x * 16384u + (idx & 16383u)] = acc[m] + r[m];
            if (K == 7) { pend[4 * m] = lds[idx & 4095u]; pend[4 * m + 1] = lds[(idx + 16u) & 4095u]; }
            if (K == 8) pend[4 * m] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((int)(((lane_off + 16u) & 63u) << 2), __builtin_bit_cast(int, r[m + 8])));
            if (K == 10) { unsigned t; asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "=v"(t) : "v"(r[m + 8])); pend[4 * m] = __builtin_bit_cast(float, t); }
            if (K == 11) pend[4 * m] = __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, r[m + 8]) >> 8) & 0xffu);
            if (K == 12) pend[4 * m] = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, r[m + 8]), 0x8000 | 0x4e));
            if (K == 13) pend[4 * m] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, r[m + 8]), 0xb1, 0xf, 0xf, true));
            if (K == 15) pend[4 * m] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r[m + 8]), 5));
            if (K == 9) pend[4 * m] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, r[m + 8]), 0x111, 0xf, 0xf, true));
        }
        BODY
#pragma unroll
        for (int i = 0; i < NR; ++i) asm volatile("" : "+v"(r[i]));
#pragma unroll
        for (int i = 0; i < NM; ++i) asm volatile("" : "+v"(acc[i]));
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NR; ++i) s += r[i];
#pragma unroll
    for (int i = 0; i < NM; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int K>
static double run(int wg_per_cu, float* out, float* src, float* dst, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    dim3 grid(256 * wg_per_cu), block(256);
    float ms = 0, best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<K>, grid, block, 0, 0, out, src, dst, iters, 0.999f, 1.001f, 0.5f, 0.25f);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    return best;
}

int main() {
    float *out, *src, *dst;
    (void)hipMalloc(&out, 256 * 256 * 32 * sizeof(float));
    (void)hipMalloc(&src, 65536 * sizeof(float));
    (void)hipMemset(src, 0, 65536 * sizeof(float));
    (void)hipMalloc(&dst, (size_t)256 * 8 * 16384 * sizeof(float));
    const int iters = 2000;
    const char* names[16] = {"none", "global_load_dword (L1 hit)", "global_load_dwordx4 (L1 hit)", "ds_read_b32", "ds_read_b128", "ds_write_b32",
                             "global_store_dword", "ds_read2_b32", "ds_bpermute_b32", "v_mov_dpp row_shr:1", "v_mov_b32_sdwa BYTE_1", "v_bfe_u32 (no sdwa)", "ds_swizzle_b32", "v_mov_dpp quad_perm", "dpp8 (n/a on gfx9)", "v_readlane_b32"};
    for (int w : {1, 2, 5}) {
        double t[16];
        t[0] = run<0>(w, out, src, dst, iters); t[1] = run<1>(w, out, src, dst, iters); t[2] = run<2>(w, out, src, dst, iters);
        t[3] = run<3>(w, out, src, dst, iters); t[4] = run<4>(w, out, src, dst, iters); t[5] = run<5>(w, out, src, dst, iters);
        t[6] = run<6>(w, out, src, dst, iters); t[7] = run<7>(w, out, src, dst, iters); t[8] = run<8>(w, out, src, dst, iters);
        t[9] = run<9>(w, out, src, dst, iters); t[10] = run<10>(w, out, src, dst, iters); t[11] = run<11>(w, out, src, dst, iters);
        t[12] = run<12>(w, out, src, dst, iters); t[13] = run<13>(w, out, src, dst, iters); t[14] = t[0]; t[15] = run<15>(w, out, src, dst, iters);
        for (int kk = 0; kk < 16; ++kk) {
            const double cyc_iter = t[kk] * 1e-3 * 2.4e9 / iters / w;       // SIMD-cycles per iteration and wave
            const double extra = (t[kk] - t[0]) * 1e-3 * 2.4e9 / iters / w / NM;
            printf("waves/SIMD=%d  %-30s %.3f ms  %.0f SIMD-cycles per wave-iteration  extra per memory instruction = %.1f cycles\n", w, names[kk], t[kk],
                   cyc_iter, extra);
        }
    }
    return 0;
}
