// Shared by the two FSDT kernels (fsdt.hip: element form; fsdt_st.hip: assembled-stencil form): kernel parameters and the
// deterministic in-kernel reduction of the three sums of squares.
#pragma once
#include "poisson_common.h"

namespace dn {

struct FsdtParams {
    float b[4][4], dx[4][4], dy[4][4];     // 1-D tables at the Gauss points (derivatives scaled by 2/h)
    float w2[4][4];                        // w[jg] * w[ig] * wscale
    float D11, D12, D22, D66, A44, A55, q;
    const float* fld[3];                   // w, phi_x, phi_y
    const float* in_scale;                 // optional 3 device floats: field k is scaled as it is loaded
    const float* in_num;                   // optional 3 + 3 device floats: field k is scaled by in_num[k] / in_den[k] (0 where in_den[k] <= 0)
    const float* in_den;
    float* norms;                          // optional 3 device floats: sqrt of the three sums of squares, written by the last workgroup
    const void* mask;
    int mask_is_u8, mask_batched;
    const float* bcf[3];
    int bcf_batched[3];
    float bcv[3];
    float* out[3];
    double* part;                          // [3][nblocks] partial sums of squares
    unsigned* counter;
    double* sumsq;                         // 3 doubles
    int nx, ny, nelx, nely, rows_per_strip, want_sums, spin_limit;
    int defer_sums;                        // != 0: the launch stores its per-workgroup partials (and their count) and leaves the reduction to its consumer;
                                           // the value is the pair's ticket, left in the workspace header for the consumer to check
    int den_ticket;                        // consumer: the ticket it expects there (a mismatch -- another reducing launch used the workspace in between -- gives NaN)
    const unsigned* den_counter;           // consumer: header of the producer's workspace (word 4: its number of workgroups) ...
    const double* den_part;                // ... and its partials [3][nblocks]
};

constexpr int FSDT_WS_NBLOCKS_WORD = 4;    // word of the workspace header in which a deferring launch leaves its number of workgroups
constexpr int FSDT_WS_TICKET_WORD = 5;     // ... and its ticket (0 after any launch that reduced in the kernel)

// Deterministic in-kernel final reduction of three scalars (same protocol as finish_sums in poisson_common.h).
__device__ __forceinline__ void finish_sums3(const FsdtParams& p, const float (&sq)[3], int tid, int nthreads, double* red, int* flag) {
    const int nblocks = gridDim.x * gridDim.y * gridDim.z;
    const int blk = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    double s[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) s[k] = block_sum((double)sq[k], red, tid, nthreads);
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) __hip_atomic_store(&p.part[(size_t)k * nblocks + blk], s[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int nshard = nblocks < DN_NSHARD ? nblocks : DN_NSHARD;
        const int shard = blk % nshard;
        const unsigned in_shard = (unsigned)((nblocks - shard + nshard - 1) / nshard);
        unsigned* sc = p.counter + 16 * (1 + shard);
        int last = 0;
        const unsigned prev = __hip_atomic_fetch_add(sc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == in_shard - 1) {
            __hip_atomic_store(sc, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned prev2 = __hip_atomic_fetch_add(p.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = (prev2 == (unsigned)(nshard - 1)) ? 1 : 0;
        }
        *flag = last;
    }
    __syncthreads();
    if (*flag) {
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        // on the critical path of the whole launch: the partials of all three sums are requested eight at a time before any is added
        // (one L2 round trip per 24 loads instead of per load; same per-thread order of additions: bitwise the same sums)
        double e3[3] = {0.0, 0.0, 0.0};
        for (int i0 = tid; i0 < nblocks; i0 += nthreads * 8) {
            double v[3][8];
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int i = i0 + j * nthreads;
                    v[k][j] = __hip_atomic_load(&p.part[(size_t)k * nblocks + (i < nblocks ? i : 0)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int j = 0; j < 8; ++j) e3[k] += (i0 + j * nthreads < nblocks) ? v[k][j] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double e = block_sum(e3[k], red, tid, nthreads);
            if (tid == 0) {
                if (p.sumsq) p.sumsq[k] = e;
                if (p.norms) p.norms[k] = (float)sqrt(e);
            }
        }
        if (tid == 0) {
            __hip_atomic_store(p.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            p.counter[FSDT_WS_TICKET_WORD] = 0u;          // the partials of an earlier deferring launch in this workspace are gone
        }
    }
}

// defer_sums: the workgroup's three partials and nothing else (no arrival counter, no wait: a kernel boundary orders them before the consumer)
__device__ __forceinline__ void store_partials3(const FsdtParams& p, const float (&sq)[3], int tid, int nthreads, double* red) {
    const int nblocks = gridDim.x * gridDim.y * gridDim.z;
    const int blk = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double s = block_sum((double)sq[k], red, tid, nthreads);
        if (tid == 0) p.part[(size_t)k * nblocks + blk] = s;
    }
    if (tid == 0 && blk == 0) {
        p.counter[FSDT_WS_NBLOCKS_WORD] = (unsigned)nblocks;
        p.counter[FSDT_WS_TICKET_WORD] = (unsigned)p.defer_sums;
    }
}

// Consumer of a deferring launch: every workgroup forms the producer's three sums from its partials in the order finish_sums3 uses (thread-strided,
// then the block sum: bitwise the same numbers) and returns their square roots; workgroup 0 writes the producer's sumsq / norms where asked.
// Every thread of the workgroup must call it (block sums).
__device__ __forceinline__ void den_from_partials(const FsdtParams& p, int tid, int nthreads, double* red, double* bc3, float (&den)[3]) {
    const int nb = (int)p.den_counter[FSDT_WS_NBLOCKS_WORD];
    const bool stale = p.den_counter[FSDT_WS_TICKET_WORD] != (unsigned)p.den_ticket;      // not the partials this call was paired with: never silent
    // at the start of EVERY workgroup of the consumer: the partials are requested eight per sum at a time before any is added (one L2 round trip per 24
    // loads; a load-add loop cost the B = 8 launch 29 us), and the three block sums share one LDS exchange (wave sums, then the waves in order: the
    // additions of block_sum)
    double e3[3] = {0.0, 0.0, 0.0};
    for (int i0 = tid; i0 < nb; i0 += nthreads * 8) {
        double v[3][8];
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = i0 + j * nthreads;
                v[k][j] = p.den_part[(size_t)k * nb + (i < nb ? i : 0)];
            }
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) e3[k] += (i0 + j * nthreads < nb) ? v[k][j] : 0.0;
    }
    const int lane = tid & (DN_WAVE - 1), wave = tid / DN_WAVE, nw = (nthreads + DN_WAVE - 1) / DN_WAVE;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double s = wave_sum(e3[k]);
        if (lane == 0) red[k * nw + wave] = s;          // red: >= 3 * nthreads / 64 doubles
    }
    __syncthreads();
    if (tid == 0) {
        const bool first = blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            double e = 0.0;
            for (int w = 0; w < nw; ++w) e += red[k * nw + w];
            bc3[k] = e;
            if (first) {
                if (p.sumsq) p.sumsq[k] = stale ? __builtin_nan("") : e;
                if (p.norms) p.norms[k] = stale ? __builtin_nanf("") : (float)sqrt(e);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 3; ++k) den[k] = stale ? __builtin_nanf("") : (float)sqrt(bc3[k]);
}

// fsdt_st.hip: the assembled-stencil form
int fsdt_st_launch(const dn_mesh* m, float wscale, FsdtParams& pp, hipStream_t s);
int64_t fsdt_st_workgroups(const dn_mesh* m);

}  // namespace dn
