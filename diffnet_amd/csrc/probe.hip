// Measurement probe (no reference counterpart): a plain streaming kernel with the fused Poisson kernel's access mix -- three fp32
// arrays read once, one written once -- over caller-supplied arrays.  bench.py runs it over the SAME arrays and rotation as the
// timed launches and reports its rate as roofline.stream_ceiling: what this part delivers for 3 reads + 1 write of that size when
// nothing but the bytes has to be moved (no halo rows, no reduction, no arithmetic to speak of).
#include "dn_common.h"

namespace dn {

typedef float v4f __attribute__((ext_vector_type(4)));

template <bool NT_LD, bool NT_ST>
__device__ __forceinline__ void stream4(const v4f* a, const v4f* b, const v4f* c, v4f* out, long long i) {
    v4f x, y, z;
    if constexpr (NT_LD) {
        x = __builtin_nontemporal_load(a + i); y = __builtin_nontemporal_load(b + i); z = __builtin_nontemporal_load(c + i);
    } else {
        x = a[i]; y = b[i]; z = c[i];
    }
    const v4f r = x * y + z;
    if constexpr (NT_ST) __builtin_nontemporal_store(r, out + i);
    else out[i] = r;
}

// one 16-byte vector per thread and array
template <bool NT_LD, bool NT_ST>
__global__ void __launch_bounds__(256) stream_flat_kernel(const v4f* __restrict__ a, const v4f* __restrict__ b, const v4f* __restrict__ c,
                                                          v4f* __restrict__ out, long long n4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) stream4<NT_LD, NT_ST>(a, b, c, out, i);
}

// workgroup-contiguous blocks, U vectors per thread and array in flight (all loads before the first store), grid-stride over blocks
template <bool NT_LD, bool NT_ST, int U>
__global__ void __launch_bounds__(256) stream_blocks_kernel(const v4f* __restrict__ a, const v4f* __restrict__ b, const v4f* __restrict__ c,
                                                            v4f* __restrict__ out, long long n4) {
    const long long per = 256ll * U;
    for (long long blk = blockIdx.x; blk * per < n4; blk += gridDim.x) {
        const long long i0 = blk * per + threadIdx.x;
        v4f x[U], y[U], z[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const long long i = i0 + 256ll * k < n4 ? i0 + 256ll * k : n4 - 1;
            if constexpr (NT_LD) {
                x[k] = __builtin_nontemporal_load(a + i); y[k] = __builtin_nontemporal_load(b + i); z[k] = __builtin_nontemporal_load(c + i);
            } else {
                x[k] = a[i]; y[k] = b[i]; z[k] = c[i];
            }
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const long long i = i0 + 256ll * k;
            if (i < n4) {
                const v4f r = x[k] * y[k] + z[k];
                if constexpr (NT_ST) __builtin_nontemporal_store(r, out + i);
                else out[i] = r;
            }
        }
    }
}

}  // namespace dn

using namespace dn;

// mode: bit 0 = non-temporal stores, bit 1 = non-temporal loads, bits 2.. = form (0: one vector per thread; 1: blocks of 4 vectors per
// thread, one block per workgroup; 2: the same, 2048 persistent workgroups striding over the blocks)
extern "C" int dn_probe_stream(const float* a, const float* b, const float* c, float* out, int64_t n, int32_t mode, void* stream) {
    if (!a || !b || !c || !out || n < 4 || (n & 3) || mode < 0 || (mode >> 2) > 2) return DN_E_BADARG;
    for (const void* q : {(const void*)a, (const void*)b, (const void*)c, (const void*)out})
        if (reinterpret_cast<uintptr_t>(q) & 15) return DN_E_BADARG;
    const long long n4 = n / 4;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const v4f *pa = reinterpret_cast<const v4f*>(a), *pb = reinterpret_cast<const v4f*>(b), *pc = reinterpret_cast<const v4f*>(c);
    v4f* po = reinterpret_cast<v4f*>(out);
    const int form = mode >> 2;
    const bool nts = mode & 1, ntl = mode & 2;
    const long long flat_wg = (n4 + 255) / 256, blk_wg = (n4 + 1023) / 1024;
    if (flat_wg >= (1ll << 31)) return DN_E_UNSUPPORTED;
#define DN_PROBE(LD, ST)                                                                                                                  \
    do {                                                                                                                                  \
        if (form == 0) hipLaunchKernelGGL((stream_flat_kernel<LD, ST>), dim3((unsigned)flat_wg), dim3(256), 0, s, pa, pb, pc, po, n4);       \
        else hipLaunchKernelGGL((stream_blocks_kernel<LD, ST, 4>), dim3((unsigned)(form == 1 ? blk_wg : (blk_wg < 2048 ? blk_wg : 2048))),  \
                                dim3(256), 0, s, pa, pb, pc, po, n4);                                                                     \
    } while (0)
    if (ntl && nts) DN_PROBE(true, true);
    else if (ntl) DN_PROBE(true, false);
    else if (nts) DN_PROBE(false, true);
    else DN_PROBE(false, false);
#undef DN_PROBE
    DN_LAUNCH_CHECK();
    return 0;
}
