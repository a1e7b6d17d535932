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

// The marching access pattern of the fused 2-D kernel without its arithmetic: a workgroup of 128 threads owns a strip of R node rows of
// one 512-wide sample and walks it row by row -- per row one 16-byte load per thread and input array (+ optionally the dword of the node
// shared with the right neighbour), a dependent (load -> use -> store) step, one 16-byte store -- with D rows requested ahead of the
// row being consumed.  What rate does this structure reach for a given (R, D), i.e. waves per SIMD and rows in flight per wave?
// SHARED: 0 no shared node, 1 a dword load per lane, 2 the neighbouring lane's first value (ds_bpermute) + a dword load by lane 63 only
template <int D, int SHARED, bool NT_ST, bool NT_LD>
__global__ void __launch_bounds__(128) march_probe_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                                          float* __restrict__ out, int ny, int R, int halo) {
    const int tid = threadIdx.x, strip = blockIdx.y, smp = blockIdx.z;
    const long long base = (long long)smp * ny * 512;
    const float *pa = a + base, *pb = b + base, *pc = c + base;
    float* po = out + base;
    const int y0 = strip * R, y1 = min(y0 + R, ny);
    const int r0 = max(y0 - halo, 0), r1 = min(y1 + halo, ny);          // rows read: the strip's own + `halo` rows on either side
    const unsigned x0 = 4u * tid, xs = min(4u * tid + 4u, 511u);
    v4f ra[D], rb[D], rc[D];
    float sa[D], sb[D], sc[D];
    auto issue = [&](int slot, int y) {
        const unsigned off = (unsigned)min(y, r1 - 1) * 512u;      // rows past the strip's last one re-read that row (cache hits), as the real kernel does
        if constexpr (NT_LD) {
            ra[slot] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(pa + off + x0));
            rb[slot] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(pb + off + x0));
            rc[slot] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(pc + off + x0));
        } else {
            ra[slot] = *reinterpret_cast<const v4f*>(pa + off + x0);
            rb[slot] = *reinterpret_cast<const v4f*>(pb + off + x0);
            rc[slot] = *reinterpret_cast<const v4f*>(pc + off + x0);
        }
        if constexpr (SHARED == 1) { sa[slot] = pa[off + xs]; sb[slot] = pb[off + xs]; sc[slot] = pc[off + xs]; }
        if constexpr (SHARED == 2) {
            sa[slot] = sb[slot] = sc[slot] = 0.f;
            if ((tid & 63) == 63) { sa[slot] = pa[off + xs]; sb[slot] = pb[off + xs]; sc[slot] = pc[off + xs]; }
        }
    };
#pragma unroll
    for (int d = 0; d < D; ++d) issue(d, r0 + d);
    v4f carry = {0.f, 0.f, 0.f, 0.f};
    for (int y = r0; y < r1; y += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (y + d < r1) {
                v4f r = ra[d] * rb[d] + rc[d] + carry;
                if constexpr (SHARED == 1) r.w += sa[d] * sb[d] + sc[d];
                if constexpr (SHARED == 2) {
                    const bool last = (tid & 63) == 63;
                    const float na = __shfl_down(ra[d].x, 1, 64), nb = __shfl_down(rb[d].x, 1, 64), nc = __shfl_down(rc[d].x, 1, 64);
                    r.w += (last ? sa[d] : na) * (last ? sb[d] : nb) + (last ? sc[d] : nc);
                }
                carry = r * 0.5f;
                issue(d, y + d + D);
                if (y + d >= y0 && y + d < y1) {
                    if constexpr (NT_ST) __builtin_nontemporal_store(r, reinterpret_cast<v4f*>(po + (unsigned)(y + d) * 512u + x0));
                    else *reinterpret_cast<v4f*>(po + (unsigned)(y + d) * 512u + x0) = r;
                }
            }
        }
    }
}

// Paired form: rows (2k, 2k + 1) -- one 4 KB block of a 512-wide fp32 array -- are requested TOGETHER and consumed together (two rows per
// trip), optionally with the next pair already in flight (AHEAD = 2 pairs).
template <int AHEAD, bool NT_ST, bool NT_LD>
__global__ void __launch_bounds__(128) march_pairs_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                                          float* __restrict__ out, int ny, int R) {
    const int tid = threadIdx.x, strip = blockIdx.y, smp = blockIdx.z;
    const long long base = (long long)smp * ny * 512;
    const float *pa = a + base, *pb = b + base, *pc = c + base;
    float* po = out + base;
    const int y0 = strip * R, y1 = min(y0 + R, ny);
    const unsigned x0 = 4u * tid;
    v4f ra[AHEAD][2], rb[AHEAD][2], rc[AHEAD][2];
    auto ld = [&](const float* p, unsigned off) {
        if constexpr (NT_LD) return __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p + off + x0));
        else return *reinterpret_cast<const v4f*>(p + off + x0);
    };
    auto issue = [&](int slot, int y) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const unsigned off = (unsigned)min(y + k, y1 - 1) * 512u;
            ra[slot][k] = ld(pa, off); rb[slot][k] = ld(pb, off); rc[slot][k] = ld(pc, off);
        }
    };
#pragma unroll
    for (int s = 0; s < AHEAD; ++s) issue(s, y0 + 2 * s);
    v4f carry = {0.f, 0.f, 0.f, 0.f};
    for (int y = y0; y < y1; y += 2 * AHEAD) {
#pragma unroll
        for (int s = 0; s < AHEAD; ++s) {
            const int yy = y + 2 * s;
            if (yy < y1) {
                v4f r0 = ra[s][0] * rb[s][0] + rc[s][0] + carry;
                v4f r1 = ra[s][1] * rb[s][1] + rc[s][1] + r0 * 0.5f;
                carry = r1 * 0.5f;
                issue(s, yy + 2 * AHEAD);
                if constexpr (NT_ST) {
                    __builtin_nontemporal_store(r0, reinterpret_cast<v4f*>(po + (unsigned)yy * 512u + x0));
                    if (yy + 1 < y1) __builtin_nontemporal_store(r1, reinterpret_cast<v4f*>(po + (unsigned)(yy + 1) * 512u + x0));
                } else {
                    *reinterpret_cast<v4f*>(po + (unsigned)yy * 512u + x0) = r0;
                    if (yy + 1 < y1) *reinterpret_cast<v4f*>(po + (unsigned)(yy + 1) * 512u + x0) = r1;
                }
            }
        }
    }
}

// Tile form: a workgroup of NT threads owns TR node rows of one 512-wide sample, requests ALL the rows it needs -- its own and one halo
// row on either side, three arrays -- at once (every load in flight before the first use), parks them in LDS, then walks the rows out of
// LDS (a dependent chain on LDS latency, not on HBM latency) and stores its TR rows.  Many short-lived workgroups, like a flat stream.
template <int NT, int TR, bool NT_LD, bool XCD>
__global__ void __launch_bounds__(NT) tile_probe_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                                        float* __restrict__ out, int ny, int tiles) {
    constexpr int ROWS = TR + 2;
    constexpr int V = ROWS * 128 / NT + ((ROWS * 128) % NT ? 1 : 0);       // 16-byte vectors per thread and array
    extern __shared__ v4f lds[];                                           // [3][ROWS][128]
    unsigned lid = blockIdx.x;
    if constexpr (XCD) {        // consecutive tiles (which share halo rows) on the same XCD
        const unsigned nwg = gridDim.x, xcd = lid & 7u, idx = lid >> 3, base = nwg >> 3, rem = nwg & 7u;
        lid = xcd * base + min(xcd, rem) + idx;
    }
    const int tile = (int)(lid % (unsigned)tiles), smp = (int)(lid / (unsigned)tiles);
    const long long sbase = (long long)smp * ny * 512;
    const float *pa = a + sbase, *pb = b + sbase, *pc = c + sbase;
    float* po = out + sbase;
    const int y0 = tile * TR;
    const int tid = threadIdx.x;
    v4f ra[V], rb[V], rc[V];
#pragma unroll
    for (int k = 0; k < V; ++k) {
        const int i = tid + k * NT, row = min(i >> 7, ROWS - 1), col = i & 127;
        const unsigned off = (unsigned)min(max(y0 - 1 + row, 0), ny - 1) * 512u + 4u * col;
        if constexpr (NT_LD) {
            ra[k] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(pa + off));
            rb[k] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(pb + off));
            rc[k] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(pc + off));
        } else {
            ra[k] = *reinterpret_cast<const v4f*>(pa + off);
            rb[k] = *reinterpret_cast<const v4f*>(pb + off);
            rc[k] = *reinterpret_cast<const v4f*>(pc + off);
        }
    }
#pragma unroll
    for (int k = 0; k < V; ++k) {
        const int i = tid + k * NT;
        if (i < ROWS * 128) { lds[i] = ra[k]; lds[ROWS * 128 + i] = rb[k]; lds[2 * ROWS * 128 + i] = rc[k]; }
    }
    __syncthreads();
    // walk: NT / 128 row groups, each thread a 4-node column segment; TR rows are split among the row groups
    const int col = tid & 127, grp = tid >> 7;
    constexpr int G = NT / 128;
    v4f carry = {0.f, 0.f, 0.f, 0.f};
    for (int r = grp; r < TR; r += G) {
        v4f acc = carry;
#pragma unroll
        for (int d = 0; d < 3; ++d) {       // the row and its two neighbours (what an element layer needs)
            const int i = (r + d) * 128 + col;
            acc += lds[i] * lds[ROWS * 128 + i] + lds[2 * ROWS * 128 + i];
        }
        carry = acc * 0.25f;
        if (y0 + r < ny) __builtin_nontemporal_store(acc, reinterpret_cast<v4f*>(po + (unsigned)(y0 + r) * 512u + 4u * col));
    }
}

}  // namespace dn

using namespace dn;

// tile probe: threads 256 | 512, rows_per_tile 4 | 8 | 16, flags: bit 3 non-temporal loads, bit 6 XCD-aware tile order
extern "C" int dn_probe_tile(const float* a, const float* b, const float* c, float* out, int32_t B, int32_t ny, int32_t rows_per_tile,
                             int32_t threads, int32_t flags, void* stream) {
    if (!a || !b || !c || !out || B < 1 || ny < 2) return DN_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int tiles = (ny + rows_per_tile - 1) / rows_per_tile;
    const dim3 grid((unsigned)(tiles * B));
    const size_t lds = (size_t)3 * (rows_per_tile + 2) * 128 * 16;
    const bool nt = flags & 8, xcd = flags & 64;
#define DN_TILE(NT, TR)                                                                                                                  \
    do {                                                                                                                                 \
        if (nt && xcd) hipLaunchKernelGGL((tile_probe_kernel<NT, TR, true, true>), grid, dim3(NT), lds, s, a, b, c, out, ny, tiles);      \
        else if (nt) hipLaunchKernelGGL((tile_probe_kernel<NT, TR, true, false>), grid, dim3(NT), lds, s, a, b, c, out, ny, tiles);       \
        else if (xcd) hipLaunchKernelGGL((tile_probe_kernel<NT, TR, false, true>), grid, dim3(NT), lds, s, a, b, c, out, ny, tiles);      \
        else hipLaunchKernelGGL((tile_probe_kernel<NT, TR, false, false>), grid, dim3(NT), lds, s, a, b, c, out, ny, tiles);              \
    } while (0)
    if (threads == 256 && rows_per_tile == 4) DN_TILE(256, 4);
    else if (threads == 256 && rows_per_tile == 8) DN_TILE(256, 8);
    else if (threads == 512 && rows_per_tile == 8) DN_TILE(512, 8);
    else if (threads == 512 && rows_per_tile == 16) DN_TILE(512, 16);
    else if (threads == 256 && rows_per_tile == 16) DN_TILE(256, 16);
    else return DN_E_UNSUPPORTED;
#undef DN_TILE
    DN_LAUNCH_CHECK();
    return 0;
}

// rows_ahead: 1..4 rows requested ahead per wave; flags: bit 0 read one halo row on either side of a strip, bit 1 also the dword of the
// shared node, bit 2 non-temporal stores, bit 3 non-temporal vector loads, bit 4 shared node from the neighbouring lane (+ a load by lane 63).
// Arrays (B, ny, 512) fp32.
extern "C" int dn_probe_march(const float* a, const float* b, const float* c, float* out, int32_t B, int32_t ny, int32_t R, int32_t rows_ahead,
                              int32_t flags, void* stream) {
    if (!a || !b || !c || !out || B < 1 || ny < 2 || R < 1 || rows_ahead < 1 || rows_ahead > 4) return DN_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid(1, (ny + R - 1) / R, B), block(128);
    const int halo = flags & 1;
    if (flags & 32) {            // paired rows (no halo, no shared node): rows_ahead = pairs in flight (1 or 2)
        if (R & 1) return DN_E_BADARG;
#define DN_PAIRS(A)                                                                                                              \
    do {                                                                                                                         \
        if ((flags & 4) && (flags & 8)) hipLaunchKernelGGL((march_pairs_kernel<A, true, true>), grid, block, 0, s, a, b, c, out, ny, R);   \
        else if (flags & 4) hipLaunchKernelGGL((march_pairs_kernel<A, true, false>), grid, block, 0, s, a, b, c, out, ny, R);              \
        else if (flags & 8) hipLaunchKernelGGL((march_pairs_kernel<A, false, true>), grid, block, 0, s, a, b, c, out, ny, R);              \
        else hipLaunchKernelGGL((march_pairs_kernel<A, false, false>), grid, block, 0, s, a, b, c, out, ny, R);                            \
    } while (0)
        if (rows_ahead >= 2) DN_PAIRS(2);
        else DN_PAIRS(1);
#undef DN_PAIRS
        DN_LAUNCH_CHECK();
        return 0;
    }
#define DN_MARCH2(D, SH)                                                                                                             \
    do {                                                                                                                             \
        if ((flags & 4) && (flags & 8)) hipLaunchKernelGGL((march_probe_kernel<D, SH, true, true>), grid, block, 0, s, a, b, c, out, ny, R, halo);   \
        else if (flags & 4) hipLaunchKernelGGL((march_probe_kernel<D, SH, true, false>), grid, block, 0, s, a, b, c, out, ny, R, halo);              \
        else if (flags & 8) hipLaunchKernelGGL((march_probe_kernel<D, SH, false, true>), grid, block, 0, s, a, b, c, out, ny, R, halo);              \
        else hipLaunchKernelGGL((march_probe_kernel<D, SH, false, false>), grid, block, 0, s, a, b, c, out, ny, R, halo);                            \
    } while (0)
#define DN_MARCH(D)                                                          \
    do {                                                                     \
        if (flags & 16) DN_MARCH2(D, 2);                                     \
        else if (flags & 2) DN_MARCH2(D, 1);                                 \
        else DN_MARCH2(D, 0);                                                \
    } while (0)
    switch (rows_ahead) {
        case 1: DN_MARCH(1); break;
        case 2: DN_MARCH(2); break;
        case 3: DN_MARCH(3); break;
        default: DN_MARCH(4); break;
    }
#undef DN_MARCH2
#undef DN_MARCH
    DN_LAUNCH_CHECK();
    return 0;
}

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
