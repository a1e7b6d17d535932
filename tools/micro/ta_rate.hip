// Micro-benchmark: vector-memory instruction throughput of one CU (TA / TCP path) for the access shapes of the FEM kernels.
// Every wave issues NL independent loads per iteration from a small (L1/L2-resident) array and adds them up; no other work.
// Reported: CU-cycles per wave-level load instruction (2.4 GHz) at 8 waves per SIMD, i.e. the rate the address/tag/data path
// sustains -- the number the 3-D kernel is designed against (profiles/r2_ta_rate.txt).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define NL 8

struct __attribute__((packed, aligned(4))) F2U { float a, b; };
struct __attribute__((packed, aligned(4))) F4U { float a, b, c, d; };

// shape: 0 dword, 256 B contiguous per wave | 1 dword, 4 rows x 16 lanes (row stride 1 KB) | 2 dwordx2 at 4-B stride, 4 rows x 16 lanes
//        3 ubyte, 4 rows x 16 lanes | 4 dwordx4 contiguous (1 KB per wave) | 5 dwordx4, 16 rows x 4 lanes (64 B per row) | 6 ushort at 1-B stride, 4 rows
//        7 dwordx4 4-B aligned only (misaligned), 16 rows x 4 lanes | 8 dword store contiguous | 9 dword store 4 rows x 16 lanes
template <int SHAPE>
__global__ void __launch_bounds__(256) k(float* out, const float* __restrict__ src, float* __restrict__ dst, int iters) {
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const char* base = reinterpret_cast<const char*>(src) + wave * 4096u;
    unsigned off;
    if (SHAPE == 0 || SHAPE == 8) off = lane * 4u;
    else if (SHAPE == 1 || SHAPE == 2 || SHAPE == 9) off = (lane >> 4) * 1024u + (lane & 15u) * 4u;
    else if (SHAPE == 3 || SHAPE == 6) off = (lane >> 4) * 1024u + (lane & 15u);
    else if (SHAPE == 4) off = lane * 16u;
    else if (SHAPE == 5) off = (lane >> 2) * 1024u + (lane & 3u) * 16u;
    else off = (lane >> 2) * 1024u + (lane & 3u) * 16u + 4u;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        const unsigned o = off + (unsigned)(it & 7) * 64u;
#pragma unroll
        for (int m = 0; m < NL; ++m) {
            const char* p = base + o + (unsigned)m * 16384u;
            if (SHAPE == 0 || SHAPE == 1) acc += *reinterpret_cast<const float*>(p);
            if (SHAPE == 2) { const F2U v = *reinterpret_cast<const F2U*>(p); acc += v.a + v.b; }
            if (SHAPE == 3) acc += (float)*reinterpret_cast<const uint8_t*>(p);
            if (SHAPE == 4 || SHAPE == 5) { const float4 v = *reinterpret_cast<const float4*>(p); acc += v.x + v.y + v.z + v.w; }
            if (SHAPE == 6) acc += (float)*reinterpret_cast<const uint16_t*>(p);
            if (SHAPE == 7) { const F4U v = *reinterpret_cast<const F4U*>(p); acc += v.a + v.b + v.c + v.d; }
            if (SHAPE == 8 || SHAPE == 9) *reinterpret_cast<float*>(reinterpret_cast<char*>(dst) + (size_t)blockIdx.x * 262144u + wave * 4096u + o + (unsigned)m * 16384u) = acc + (float)m;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int SHAPE>
static void run(const char* name, float* out, float* src, float* dst) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4000;
    for (int wg_per_cu : {1, 2, 8}) {
        float ms = 0, best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k<SHAPE>, dim3(256 * wg_per_cu), dim3(256), 0, 0, out, src, dst, iters);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best) best = ms;
        }
        const double instr_per_cu = (double)iters * NL * 4 * wg_per_cu;
        printf("%-58s waves/SIMD=%d  %.3f ms  CU-cycles per wave-instruction = %.1f\n", name, wg_per_cu, best, best * 1e-3 * 2.4e9 / instr_per_cu);
    }
}

int main() {
    float *out, *src, *dst;
    (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    (void)hipMalloc(&src, 1 << 20);
    (void)hipMemset(src, 0, 1 << 20);
    (void)hipMalloc(&dst, (size_t)256 * 8 * 262144);
    run<0>("load dword, 256 B contiguous", out, src, dst);
    run<1>("load dword, 4 rows x 16 lanes", out, src, dst);
    run<2>("load dwordx2 at 4-B stride, 4 rows x 16 lanes", out, src, dst);
    run<3>("load ubyte, 4 rows x 16 lanes", out, src, dst);
    run<6>("load ushort at 1-B stride, 4 rows x 16 lanes", out, src, dst);
    run<4>("load dwordx4, 1 KB contiguous", out, src, dst);
    run<5>("load dwordx4, 16 rows x 4 lanes (64 B rows)", out, src, dst);
    run<7>("load dwordx4 4-B aligned, 16 rows x 4 lanes", out, src, dst);
    run<8>("store dword, 256 B contiguous", out, src, dst);
    run<9>("store dword, 4 rows x 16 lanes", out, src, dst);
    return 0;
}
