// Micro-benchmark: do waves that sit at DIFFERENT program counters still pair on a SIMD?  Four different 256-instruction fp32 bodies;
// MODE 0: every workgroup runs body 0 (the waves of a SIMD walk the same code), MODE 1: workgroup w runs body w & 3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "body_div.h"
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
    float r[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) r[i] = threadIdx.x * 0.001f + i;
    const int sel = MODE ? (blockIdx.x >> 8) & 3 : 0;          // 256 workgroups per "layer" of residency: co-resident workgroups differ
    for (int it = 0; it < iters; ++it) {
        if (sel == 0) { BODY0 }
        else if (sel == 1) { BODY1 }
        else if (sel == 2) { BODY2 }
        else { BODY3 }
#pragma unroll
        for (int i = 0; i < NR; ++i) asm volatile("" : "+v"(r[i]));
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NR; ++i) s += r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
static void run(float* out) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 2000;
    for (int w : {1, 2, 4, 8}) {
        float ms = 0, best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k<MODE>, dim3(256 * w), dim3(256), 0, 0, out, iters);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best) best = ms;
        }
        printf("mode %d (%s) waves/SIMD=%d  %.3f ms  SIMD-cycles per wave-instr (~200 VALU per body, 2.4 GHz) = %.2f\n", MODE,
               MODE ? "co-resident workgroups run different code" : "all workgroups run the same code", w, best, best * 1e-3 * 2.4e9 / ((double)iters * 200 * w));
    }
}
int main() {
    float* out;
    (void)hipMalloc(&out, 256 * 256 * 32 * sizeof(float));
    run<0>(out);
    run<1>(out);
    return 0;
}
