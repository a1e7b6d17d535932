// Micro-benchmark: the same instruction streams as valu_sgpr.hip, on float2 ext-vectors (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32):
// what does a packed fp32 instruction cost a SIMD, with and without SGPR operands, at 1..4 waves per SIMD?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f pkfma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f pkfma(float a, v2f b, v2f c) { return __builtin_elementwise_fma((v2f)(a), b, c); }
#define fmaf pkfma
#include BODYFILE
#ifndef NLOOP
#define NLOOP NI
#endif
__global__ void __launch_bounds__(256) k(float* out, int iters, float c0, float c1, float c2, float c3, float c4, float c5, float c6, float c7) {
    v2f r[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) r[i] = (v2f){threadIdx.x * 0.001f + i, threadIdx.x * 0.002f - i};
    for (int it = 0; it < iters; ++it) {
        BODY
#pragma unroll
        for (int i = 0; i < NR; ++i) asm volatile("" : "+v"(r[i]));
    }
    v2f s = 0.f;
#pragma unroll
    for (int i = 0; i < NR; ++i) s += r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
int main() {
    float* out;
    (void)hipMalloc(&out, 256 * 256 * 32 * sizeof(float));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 2000;
    for (int w : {1, 2, 3}) {
        float ms = 0, best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256 * w), dim3(256), 0, 0, out, iters, 0.999f, 1.001f, 0.5f, 0.25f, 0.75f, 1.5f, 0.9f, 1.1f);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best) best = ms;
        }
        printf("packed %s loopVALU=%d waves/SIMD=%d  %.3f ms  SIMD-cycles per packed wave-instr (at 2.4 GHz) = %.2f\n", BODYFILE, NLOOP, w, best, best * 1e-3 * 2.4e9 / ((double)iters * NLOOP * w));
    }
    return 0;
}
