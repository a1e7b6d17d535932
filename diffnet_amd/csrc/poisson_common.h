// Shared device-side pieces of the fused Poisson kernels: kernel parameter block, per-sample base pointers,
// Dirichlet handling and the in-kernel deterministic final reduction.
#pragma once
#include "dn_common.h"
#include "poisson_elem.h"

namespace dn {

struct DirichletDev {
    const void* mask;
    const float* field;
    float value;
    int mask_is_u8, mask_batched, field_batched;
    int kind;              // DN_MASK_* of a present condition, -1 when absent
    int box_faces;         // DN_MASK_BOX
    int row_bytes;         // DN_MASK_BITS: bytes per node row
};

struct PoissonParams {
    ElemTab T;
    const float* u;
    const float* nu;
    const float* f;
    const float* fgp;
    int nu_batched, f_batched;
    int f_is_load;         // `f` is the assembled load vector (dn_poisson_args.f_is_load)
    DirichletDev bc[2];
    float out_scale;
    float* out;
    double* part_energy;   // per-workgroup partial sums (workspace)
    double* part_sumsq;
    unsigned* counter;     // arrival counter of the in-kernel final reduction (self-resetting)
    double* energy;        // final scalars (may be null)
    double* sumsq;
    float* energy_f32;     // optional (float)(energy * energy_scale)
    double energy_scale;
    int nx, ny, nz;        // nodes
    int nelx, nely, nelz;  // elements
    int rows_per_strip;    // element layers per strip along the marched axis
    int nstrips;           // strips per sample along the marched axis (the whole mesh, whatever the launch selects)
    int strip_sel;         // 0: the launch covers every strip; 1: only the first and the last one; 2: all but those (dn_poisson_args.strip_select)
    int acc_sums;          // the final scalars are ADDED to what energy / sumsq hold (the second launch of a split evaluation)
    int defer_sums;        // the launch only writes its per-workgroup partial sums; dn_poisson_finish_sums adds them up (on any stream ordered after it)
    int want_sums;
    // dn_poisson_args.fold_prev: the partial sums an earlier launch left (defer_sums) are added up by this launch's first workgroup
    const double* fold_pe;
    const double* fold_ps;
    double* fold_energy;
    double* fold_sumsq;
    float* fold_energy_f32;
    double fold_scale;
    int fold_n, fold_acc;
    int spin_limit;        // bound of the chained strips' LDS hand-over polls (0: the kernels' default; "HANDOVER_SPIN_LIMIT": test hook for the error path)
};

// strip index of the idx-th launched strip (split evaluations launch a subset of the strips: PoissonParams::strip_sel)
__device__ __forceinline__ int selected_strip(const PoissonParams& p, int idx) {
    return p.strip_sel == 1 ? (idx == 0 ? 0 : p.nstrips - 1) : (p.strip_sel == 2 ? idx + 1 : idx);
}

// Per-sample base pointers (wave-uniform): all in-kernel indexing is a 32-bit offset from these.
struct SampleBases {
    const float* u;
    const float* nu;
    const float* f;
    float* out;
    const void* mask[2];
    const float* field[2];
};

__device__ __forceinline__ SampleBases sample_bases(const PoissonParams& p, int b, int64_t nps) {
    SampleBases s;
    s.u = p.u + (int64_t)b * nps;
    s.nu = p.nu ? p.nu + (p.nu_batched ? (int64_t)b * nps : 0) : nullptr;
    s.f = p.f ? p.f + (p.f_batched ? (int64_t)b * nps : 0) : nullptr;
    s.out = p.out ? p.out + (int64_t)b * nps : nullptr;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const DirichletDev& d = p.bc[k];
        const int64_t mo = d.mask_batched ? (int64_t)b * nps : 0;
        if (d.kind == DN_MASK_BITS)        // bit-packed: rows of row_bytes bytes, ny * nz rows per sample
            s.mask[k] = reinterpret_cast<const uint8_t*>(d.mask) + (d.mask_batched ? (int64_t)b * ((int64_t)p.ny * p.nz) * d.row_bytes : 0);
        else
            s.mask[k] = d.mask ? (d.mask_is_u8 ? (const void*)(reinterpret_cast<const uint8_t*>(d.mask) + mo)
                                               : (const void*)(reinterpret_cast<const float*>(d.mask) + mo))
                               : nullptr;
        s.field[k] = d.field ? d.field + (d.field_batched ? (int64_t)b * nps : 0) : nullptr;
    }
    return s;
}

// Dirichlet conditions for one row segment (nodes x0..x0+NW): all mask / value loads are issued first, then
// u <- where(mask > 0.5, value, u) is applied with selects.  Returns the bit set of fixed nodes.
template <int NW, bool VEC>
__device__ __forceinline__ unsigned load_apply_bc(const PoissonParams& p, const SampleBases& sb, unsigned rowoff, int x0,
                                                  float (&u)[NW + 1]) {
    uint8_t m8[2][NW + 1];
    float mf[2][NW + 1], fv[2][NW + 1];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        if (sb.mask[k] != nullptr) {
            if (p.bc[k].mask_is_u8) load_seg<NW, VEC>(reinterpret_cast<const uint8_t*>(sb.mask[k]), rowoff, x0, p.nx, m8[k]);
            else load_seg<NW, VEC>(reinterpret_cast<const float*>(sb.mask[k]), rowoff, x0, p.nx, mf[k]);
            if (sb.field[k]) load_seg<NW, VEC>(sb.field[k], rowoff, x0, p.nx, fv[k]);
        }
    }
    unsigned bits = 0u;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        if (sb.mask[k] != nullptr) {
            unsigned kb = 0u;
            if (p.bc[k].mask_is_u8) {
#pragma unroll
                for (int n = 0; n <= NW; ++n) kb |= (m8[k][n] != 0) ? (1u << n) : 0u;
            } else {
#pragma unroll
                for (int n = 0; n <= NW; ++n) kb |= (mf[k][n] > 0.5f) ? (1u << n) : 0u;
            }
            if (sb.field[k]) {
#pragma unroll
                for (int n = 0; n <= NW; ++n) u[n] = (kb & (1u << n)) ? fv[k][n] : u[n];
            } else {
                const float val = p.bc[k].value;
#pragma unroll
                for (int n = 0; n <= NW; ++n) u[n] = (kb & (1u << n)) ? val : u[n];
            }
            bits |= kb;
        }
    }
    return bits;
}

// Block-level reduction of the two scalars + arrival of this workgroup at the in-kernel final reduction.
// Two-level arrival (DN_NSHARD shard counters on separate 64-B lines, then one top counter) keeps the
// same-address atomic fan-in at ~nblocks/64 + 64 instead of nblocks (one address retires only ~88 atomics/us:
// MI355X_MICROARCH.md "fanin").  The workgroup that arrives last sums all per-workgroup partials in index
// order (=> deterministic whatever the arrival order); counters are reset by their last arriver, so the
// workspace is ready for the next launch.  Protocol (cdna_hip_programming.md, Guideline 16): partials are
// stored write-through (sc1) and drained before the arrival atomic; the last arriver does an agent-scope
// acquire and reads the partials with sc1 loads.
#define DN_NSHARD 64
__device__ __forceinline__ void finish_sums(const PoissonParams& p, float e1, float e2, float sq, int tid, int nthreads,
                                            double* red, int* flag, double escale = 1.0) {
    const int nblocks = gridDim.x * gridDim.y * gridDim.z;
    const int blk = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    double es = escale * ((double)p.T.c * (double)e1 - (double)e2), ss = (double)sq;
    block_sum2(es, ss, red, tid, nthreads);
    if (p.defer_sums) {                          // no arrival protocol, no tail: the kernel boundary publishes the partials
        if (tid == 0) { p.part_energy[blk] = es; p.part_sumsq[blk] = ss; }
        return;
    }
    if (tid == 0) {
        // write-through (sc1) 8-byte stores + drain instead of an agent-scope release fence: a release is a
        // `buffer_wbl2` of the whole XCD L2, i.e. every workgroup would wait for everybody's freshly written
        // output lines to be flushed (measured: +5..30 us per workgroup at 8k workgroups).
        __hip_atomic_store(&p.part_energy[blk], es, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&p.part_sumsq[blk], ss, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int last = 0;
        if (nblocks <= DN_NSHARD) {              // few workgroups: one counter, one atomic round trip on the launch's critical path
            const unsigned prev = __hip_atomic_fetch_add(p.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = (prev == (unsigned)(nblocks - 1)) ? 1 : 0;
        } else {
            const int nshard = DN_NSHARD;
            const int shard = blk % nshard;
            const unsigned in_shard = (unsigned)((nblocks - shard + nshard - 1) / nshard);
            unsigned* sc = p.counter + 16 * (1 + shard);              // shard counters: one per 64-B line
            const unsigned prev = __hip_atomic_fetch_add(sc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev == in_shard - 1) {
                __hip_atomic_store(sc, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned prev2 = __hip_atomic_fetch_add(p.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = (prev2 == (unsigned)(nshard - 1)) ? 1 : 0;
            }
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
        // the last arriver's loop sits on the critical path of the whole launch (every other workgroup has finished): sixteen partials of
        // each sum are requested before the first is added (one L2 round trip per 16 partials instead of per 1; 2048 workgroups x 128
        // threads: 1 trip instead of 16).  Same per-thread order of additions as the plain loop: bitwise the same sums.
        double e = 0.0, s = 0.0;
        constexpr int NB_ = 16;                       // partials of each sum in flight per thread
        for (int i0 = tid; i0 < nblocks; i0 += nthreads * NB_) {
            double ve[NB_], vs[NB_];
#pragma unroll
            for (int k = 0; k < NB_; ++k) {
                const int i = i0 + k * nthreads;
                const int ic = i < nblocks ? i : 0;
                ve[k] = __hip_atomic_load(&p.part_energy[ic], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                vs[k] = __hip_atomic_load(&p.part_sumsq[ic], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int k = 0; k < NB_; ++k) {
                const bool ok = i0 + k * nthreads < nblocks;
                e += ok ? ve[k] : 0.0;
                s += ok ? vs[k] : 0.0;
            }
        }
        block_sum2(e, s, red, tid, nthreads);
        if (tid == 0) {
            if (p.acc_sums) {         // second launch of a split evaluation (host checks that both double slots exist): fixed order, first + second
                e += *p.energy;
                s += *p.sumsq;
            }
            if (p.energy) *p.energy = e;
            if (p.sumsq) *p.sumsq = s;
            if (p.energy_f32) *p.energy_f32 = (float)(e * p.energy_scale);
            __hip_atomic_store(p.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// dn_poisson_args.fold_prev: the final reduction of an EARLIER launch, done by the first workgroup of this one before its march (same
// arithmetic as poisson_finish_sums_kernel: each thread adds its partials in index order, fixed-order block sum -> bitwise reproducible).
// Sixteen partials of each sum are requested before the first addition (one memory round trip per 16).  `red`: >= 2 * waves doubles.
__device__ __forceinline__ void fold_prev_sums(const PoissonParams& p, int tid, int nthreads, double* red) {
    if (p.fold_n <= 0) return;                    // (wave-uniform)
    double e = 0.0, s = 0.0;
    constexpr int NB_ = 16;
    for (int i0 = tid; i0 < p.fold_n; i0 += nthreads * NB_) {
        double ve[NB_], vs[NB_];
#pragma unroll
        for (int k = 0; k < NB_; ++k) {
            const int i = i0 + k * nthreads;
            const int ic = i < p.fold_n ? i : 0;
            ve[k] = p.fold_pe[ic];
            vs[k] = p.fold_ps[ic];
        }
#pragma unroll
        for (int k = 0; k < NB_; ++k) {
            const bool ok = i0 + k * nthreads < p.fold_n;
            e += ok ? ve[k] : 0.0;
            s += ok ? vs[k] : 0.0;
        }
    }
    block_sum2(e, s, red, tid, nthreads);
    if (tid == 0) {
        if (p.fold_acc) { e += *p.fold_energy; s += *p.fold_sumsq; }
        if (p.fold_energy) *p.fold_energy = e;
        if (p.fold_sumsq) *p.fold_sumsq = s;
        if (p.fold_energy_f32) *p.fold_energy_f32 = (float)(e * p.fold_scale);
    }
}

// Raw (not yet interpreted) Dirichlet data of one row segment: issued early, applied later, so that the
// loads of the next row are in flight while the current layer is computed.
template <int NW>
struct BcRaw {
    uint32_t m[2][NW + 1];   // mask values: zero-extended byte (u8 masks) or float bit pattern
    float fv[2][NW + 1];     // Dirichlet value fields
};

#ifndef DN_NT_MASK
#define DN_NT_MASK 1                  // mask images (read once per launch) with non-temporal vector loads, like nu and f (section 4.0 of DESIGN.md)
#endif
template <int NW, bool VEC>
__device__ __forceinline__ void bc_issue(const PoissonParams& p, const SampleBases& sb, unsigned rowoff, int x0, BcRaw<NW>& r) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        if (sb.mask[k] != nullptr) {
            if (p.bc[k].mask_is_u8) {
                uint8_t t[NW + 1];
#if DN_NT_MASK
                load_seg_stream<NW, VEC>(reinterpret_cast<const uint8_t*>(sb.mask[k]), rowoff, x0, p.nx, t);
#else
                load_seg<NW, VEC>(reinterpret_cast<const uint8_t*>(sb.mask[k]), rowoff, x0, p.nx, t);
#endif
#pragma unroll
                for (int n = 0; n <= NW; ++n) r.m[k][n] = t[n];
            } else {
#if DN_NT_MASK
                load_seg_stream<NW, VEC>(reinterpret_cast<const uint32_t*>(sb.mask[k]), rowoff, x0, p.nx, r.m[k]);
#else
                load_seg<NW, VEC>(reinterpret_cast<const uint32_t*>(sb.mask[k]), rowoff, x0, p.nx, r.m[k]);
#endif
            }
            if (sb.field[k]) load_seg<NW, VEC>(sb.field[k], rowoff, x0, p.nx, r.fv[k]);
        }
    }
}

// u <- where(mask > 0.5, value, u) for both conditions in order; keep[n] = 0 on Dirichlet nodes, 1 elsewhere (all NW + 1 nodes).
template <int NW>
__device__ __forceinline__ void bc_apply_all(const PoissonParams& p, const SampleBases& sb, const BcRaw<NW>& r, float (&u)[NW + 1],
                                             float (&keep)[NW + 1]) {
#pragma unroll
    for (int n = 0; n <= NW; ++n) keep[n] = 1.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        if (sb.mask[k] != nullptr) {
            const bool u8 = p.bc[k].mask_is_u8 != 0;
            const bool hasf = sb.field[k] != nullptr;
            const float val = p.bc[k].value;
#pragma unroll
            for (int n = 0; n <= NW; ++n) {
                const bool set = u8 ? (r.m[k][n] != 0u) : (__uint_as_float(r.m[k][n]) > 0.5f);
                u[n] = set ? (hasf ? r.fv[k][n] : val) : u[n];
                keep[n] = set ? 0.f : keep[n];
            }
        }
    }
}

// same, keep[] for the NW nodes a thread owns
template <int NW>
__device__ __forceinline__ void bc_apply(const PoissonParams& p, const SampleBases& sb, const BcRaw<NW>& r, float (&u)[NW + 1],
                                         float (&keep)[NW]) {
    float k[NW + 1];
    bc_apply_all<NW>(p, sb, r, u, k);
#pragma unroll
    for (int n = 0; n < NW; ++n) keep[n] = k[n];
}

// Dirichlet conditions on exactly N nodes starting at x0 of one row (loads issued first, then selects).
// u <- where(mask > 0.5, value, u) for both conditions in order; keep[n] = 0 on Dirichlet nodes, 1 elsewhere.
template <int N, bool VEC>
__device__ __forceinline__ void bc_nodes(const PoissonParams& p, const SampleBases& sb, unsigned rowoff, int x0, float (&u)[N],
                                         float (&keep)[N]) {
    uint32_t m[2][N];
    float fv[2][N];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        if (sb.mask[k] != nullptr) {
            if (p.bc[k].mask_is_u8) {
                uint8_t t[N];
                load_own<N, VEC>(reinterpret_cast<const uint8_t*>(sb.mask[k]), rowoff, x0, p.nx, t);
#pragma unroll
                for (int n = 0; n < N; ++n) m[k][n] = t[n];
            } else {
                load_own<N, VEC>(reinterpret_cast<const uint32_t*>(sb.mask[k]), rowoff, x0, p.nx, m[k]);
            }
            if (sb.field[k]) load_own<N, VEC>(sb.field[k], rowoff, x0, p.nx, fv[k]);
        }
    }
#pragma unroll
    for (int n = 0; n < N; ++n) keep[n] = 1.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        if (sb.mask[k] != nullptr) {
            const bool u8 = p.bc[k].mask_is_u8 != 0;
            const bool hasf = sb.field[k] != nullptr;
            const float val = p.bc[k].value;
#pragma unroll
            for (int n = 0; n < N; ++n) {
                const bool set = u8 ? (m[k][n] != 0u) : (__uint_as_float(m[k][n]) > 0.5f);
                u[n] = set ? (hasf ? fv[k][n] : val) : u[n];
                keep[n] = set ? 0.f : keep[n];
            }
        }
    }
}

struct Geom2D { int T, E, chunks, strips, R, W = 1; bool ua = false; };     // W: strips chained per workgroup; ua: rows of 4 k + 1 nodes on the vector kernel (closed-form Q1 kernel only)
struct Geom3D { int TX, TY, E, chunks, tiles, strips, R; };

// 2-D Q1 marching kernels are compiled one translation unit per NGP (poisson2d_q1_g{2,3,4}.hip)
int launch_poisson2d_q1_g2(const PoissonParams& pp, const Geom2D& g, int batch, bool vec, hipStream_t s);
int launch_poisson2d_q1_g3(const PoissonParams& pp, const Geom2D& g, int batch, bool vec, hipStream_t s);
int launch_poisson2d_q1_g4(const PoissonParams& pp, const Geom2D& g, int batch, bool vec, hipStream_t s);
// closed-form 2-D Q1 kernel (poisson2d_q1_cf.hip): any rule, nodal forcing
int launch_poisson2d_q1_cf(const PoissonParams& pp, const Geom2D& g, int batch, bool vec, hipStream_t s);
int poisson2d_q1_cf_chain();      // strips per workgroup the library's closed-form kernel was built to chain (1: none)
// 3-D Q1 marching kernels likewise (poisson3d_q1_g{2,3,4}.hip)
int launch_poisson3d_q1_g2(const PoissonParams& pp, const Geom3D& g, int batch, bool vec, hipStream_t s);
int launch_poisson3d_q1_g3(const PoissonParams& pp, const Geom3D& g, int batch, bool vec, hipStream_t s);
int launch_poisson3d_q1_g4(const PoissonParams& pp, const Geom3D& g, int batch, bool vec, hipStream_t s);

// closed-form-in-z 3-D Q1 kernel (poisson3d_q1_cf.hip): exact 2-point rule, two elements per thread, energy from the nodal values
bool poisson3d_q1_cf_ok(const PoissonParams& pp);
int launch_poisson3d_q1_cf(const PoissonParams& pp, const Geom3D& g, int batch, hipStream_t s);

// 3-D Q2 / Q3 (poisson3d_gen.hip): element vectors + fixed-order gather assembly; its workspace lies behind the common header
static constexpr int64_t DN_WS_HEADER = 64 * (1 + 64);   // top counter + DN_NSHARD shard counters, one 64-B line each
static constexpr int DN_WS_ERRWORD = 8;                    // word 8 of the top counter's line: sticky error bits of the launches that used this workspace
                                                          // (bit 0: a bounded LDS hand-over poll of a chained-strip kernel ran out -- its results are NaN); read and
                                                          // cleared by dn_workspace_status
void gen3d_layout(const dn_mesh* m, long long& n1, long long& n2, long long& elem_floats);
int launch_poisson3d_gen(const PoissonParams& pp, const dn_mesh* m, void* workspace, int64_t workspace_bytes, hipStream_t s);

}  // namespace dn
