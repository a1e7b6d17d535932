// Micro-benchmark: sustained fp32 VALU rate of gfx950 vs waves per SIMD (tuning aid, not part of the library).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int PK>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 y0 = {x0, x1}, y1 = {x2, x3}, y2 = {x4, x5}, y3 = {x6, x7};
    v2 av = {a, a}, bv = {b, b};
    for (int i = 0; i < iters; ++i) {
        if (PK) {
#pragma unroll
            for (int r = 0; r < 8; ++r) { y0 = y0 * av + bv; y1 = y1 * av + bv; y2 = y2 * av + bv; y3 = y3 * av + bv; }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
                x4 = fmaf(x4, a, b); x5 = fmaf(x5, a, b); x6 = fmaf(x6, a, b); x7 = fmaf(x7, a, b);
            }
        }
    }
    if (PK) out[blockIdx.x * blockDim.x + threadIdx.x] = y0.x + y0.y + y1.x + y1.y + y2.x + y2.y + y3.x + y3.y;
    else out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 256 * 32 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int pk = 0; pk < 2; ++pk)
        for (int wg_per_cu = 1; wg_per_cu <= 8; wg_per_cu *= 2) {   // 256-thread WGs: 1 WG/CU = 1 wave/SIMD
            dim3 grid(256 * wg_per_cu), block(256);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (pk) hipLaunchKernelGGL(k<1>, grid, block, 0, 0, out, iters, 1.0001f, 0.5f);
                else hipLaunchKernelGGL(k<0>, grid, block, 0, 0, out, iters, 1.0001f, 0.5f);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double instr_per_wave = (double)iters * 32;                      // VALU instructions
            const double waves = (double)grid.x * 4;
            const double flops = waves * instr_per_wave * 64 * 2 * (pk ? 2 : 1);
            const double cyc_per_instr = ms * 1e-3 * 2.4e9 / (instr_per_wave * wg_per_cu);   // per SIMD, at 2.4 GHz
            printf("pk=%d waves/SIMD=%d  %.3f ms  %.1f TFLOP/s  SIMD-cycles per wave-instr (at 2.4GHz)=%.2f\n", pk, wg_per_cu, ms,
                   flops / ms * 1e-9, cyc_per_instr);
        }
    return 0;
}
