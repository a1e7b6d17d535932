// Micro-benchmark: issue cost of fp32 VALU instructions by operand pattern (VGPR / SGPR sources) vs waves per SIMD.
// Motivation: the 3-D Q1 kernel sustains ~4.5 SIMD-cycles per VALU wave-instruction although x = fma(x, s, s) streams reach
// 2.4 (valu_bench.hip); this measures fma(v,v,v), fma(s,v,v), mul(v,v), add(v,v) with independent accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
    float x[8], y[8], z[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x + i; y[i] = 1.0f + 1e-7f * (threadIdx.x + i); z[i] = 0.25f * i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) x[i] = fmaf(x[i], y[i], z[i]);            // 3 VGPR sources
                else if (MODE == 1) x[i] = fmaf(a, y[i], x[i]);          // SGPR, VGPR, VGPR (v_fmac)
                else if (MODE == 2) x[i] = x[i] * y[i];                  // mul v, v
                else if (MODE == 3) x[i] = x[i] + z[i];                  // add v, v
                else if (MODE == 4) x[i] = fmaf(x[i], a, b);             // 1 VGPR source
                else if (MODE == 5) x[i] = fmaf(x[i], y[(i + 1) & 7], z[(i + 3) & 7]);   // 3 VGPR sources, shuffled registers
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i] + y[i] + z[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(float* out, const char* name) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int w = 1; w <= 8; w *= 2) {
        dim3 grid(256 * w), block(256);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, out, iters, 1.0000001f, 1e-9f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s waves/SIMD=%d  %.3f ms  SIMD-cycles per wave-instr (at 2.4 GHz) = %.2f\n", name, w, ms, ms * 1e-3 * 2.4e9 / ((double)iters * 32 * w));
    }
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 256 * 8 * sizeof(float));
    run<4>(out, "fma(v, s, s)  1 VGPR src");
    run<1>(out, "fma(s, v, v)  2 VGPR src");
    run<0>(out, "fma(v, v, v)  3 VGPR src");
    run<5>(out, "fma(v, v', v'') 3 VGPR mixed");
    run<2>(out, "mul(v, v)");
    run<3>(out, "add(v, v)");
    return 0;
}
