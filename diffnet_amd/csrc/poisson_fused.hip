// Fused Poisson operator for structured Q_P meshes on gfx950: Dirichlet masking, Gauss-point
// evaluation of u / nu / f and grad u, energy density, gradient (= weak-form residual) and
// element->node assembly in ONE pass over the nodal fields.
//
// Replaces, per call, the 5-6 x ngp_total single-channel convolutions + cat + ~8 elementwise ops +
// autograd backward of the reference loss bodies (IBN_2D.py:116-134, 12_klsum.py:53-132,
// solve_in_object_3d.py:75-102, e8_*_poisson_mms.py) -- see include/diffnet_hip.h.
//
// Mapping: the slowest spatial axis is *marched*; the remaining axes are tiled.
//   2-D: a workgroup owns a row segment of T threads x E elements and marches over R element rows;
//   3-D: a workgroup owns a (TY x TX*E) in-plane tile and marches over R element planes.
// A thread keeps the node values of the current element layer and the partially assembled
// output of the layer boundary in registers; contributions to nodes shared with the neighbouring
// thread are handed over through a double-buffered LDS slot (one barrier per layer).  Layer /
// tile seams are closed by recomputing one element layer (no atomics => bitwise reproducible).
// HBM traffic is the algorithmic minimum: each nodal field is read once (+ halo re-reads that hit
// L2), the output is written once.
#include "dn_common.h"
#include "poisson_elem.h"

namespace dn {

struct DirichletDev {
    const void* mask;
    const float* field;
    float value;
    int mask_is_u8, mask_batched, field_batched;
};

struct PoissonParams {
    ElemTab T;
    const float* u;
    const float* nu;
    const float* f;
    const float* fgp;
    int nu_batched, f_batched;
    DirichletDev bc[2];
    float out_scale;
    float* out;
    double* part_energy;   // per-workgroup partial sums (workspace)
    double* part_sumsq;
    int nx, ny, nz;        // nodes
    int nelx, nely, nelz;  // elements
    int rows_per_strip;    // element layers per strip along the marched axis
    int want_energy, want_sumsq;
};

// Dirichlet application for one node.  Returns the (possibly replaced) value; sets `fixed`.
__device__ __forceinline__ float apply_bc(const PoissonParams& p, float v, int b, int64_t node, int64_t nodes_per_sample,
                                          bool& fixed) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const DirichletDev& d = p.bc[k];
        if (d.mask != nullptr) {
            const int64_t mi = (d.mask_batched ? (int64_t)b * nodes_per_sample : 0) + node;
            const bool set = d.mask_is_u8 ? (reinterpret_cast<const uint8_t*>(d.mask)[mi] != 0)
                                          : (reinterpret_cast<const float*>(d.mask)[mi] > 0.5f);
            if (set) {
                v = d.field ? d.field[(d.field_batched ? (int64_t)b * nodes_per_sample : 0) + node] : d.value;
                fixed = true;
            }
        }
    }
    return v;
}

// =============================================================================================
// 2-D kernel.  grid = (chunks_x, strips_y, B), block = T threads.
//   P   : element degree, NGP: 1-D Gauss points, E: elements per thread along x, VEC: vector ld/st legal
// =============================================================================================
template <int P, int NGP, int E, bool VEC>
__global__ void __launch_bounds__(256) poisson2d_kernel(const PoissonParams p) {
    constexpr int NB = P + 1;
    constexpr int NW = E * P;             // nodes owned per thread per node row
    const int T = blockDim.x;
    const int tid = threadIdx.x;
    const int chunk = blockIdx.x, strip = blockIdx.y, b = blockIdx.z;
    const int q = chunk * (T - 1) + tid;  // logical thread column (chunks overlap by one thread)
    const int ex0 = q * E;                // first element of this thread
    const int x0 = ex0 * P;               // first node
    const bool col_owner = !(chunk > 0 && tid == 0);
    const int64_t nps = (int64_t)p.nx * p.ny;            // nodes per sample
    const int64_t eps = (int64_t)p.nelx * p.nely;        // elements per sample
    const int R = p.rows_per_strip;
    const int ey_own = strip * R;
    const int ey_begin = ey_own > 0 ? ey_own - 1 : 0;
    const int ey_end = min(ey_own + R, p.nely);
    const bool has_nu = p.nu != nullptr;
    const int fmode = p.f ? F_NODAL : (p.fgp ? F_GP : F_NONE);
    const bool any_bc = p.bc[0].mask != nullptr || p.bc[1].mask != nullptr;

    __shared__ float xch[2][P][256];
    __shared__ double red[8];

    float cu[NB][NW + 1], cn[NB][NW + 1], cf[NB][NW + 1];
    unsigned fixed[NB];                   // bit n: node (row r, n) is a Dirichlet node
    float acc[NB][NW + 1];
#pragma unroll
    for (int r = 0; r < NB; ++r) {
        fixed[r] = 0u;
#pragma unroll
        for (int n = 0; n <= NW; ++n) { acc[r][n] = 0.f; cn[r][n] = 1.f; cf[r][n] = 0.f; }
    }

    auto load_row = [&](int r, int yr) {
        const int64_t rowbase = (int64_t)yr * p.nx;
        {
            float tmp[NW];
            load_run<NW, VEC>(p.u, (int64_t)b * nps + rowbase, x0, p.nx, 0.f, tmp);
#pragma unroll
            for (int n = 0; n < NW; ++n) cu[r][n] = tmp[n];
            cu[r][NW] = (x0 + NW < p.nx) ? p.u[(int64_t)b * nps + rowbase + x0 + NW] : 0.f;
        }
        if (has_nu) {
            const int64_t base = (p.nu_batched ? (int64_t)b * nps : 0) + rowbase;
            float tmp[NW];
            load_run<NW, VEC>(p.nu, base, x0, p.nx, 0.f, tmp);
#pragma unroll
            for (int n = 0; n < NW; ++n) cn[r][n] = tmp[n];
            cn[r][NW] = (x0 + NW < p.nx) ? p.nu[base + x0 + NW] : 0.f;
        }
        if (fmode == F_NODAL) {
            const int64_t base = (p.f_batched ? (int64_t)b * nps : 0) + rowbase;
            float tmp[NW];
            load_run<NW, VEC>(p.f, base, x0, p.nx, 0.f, tmp);
#pragma unroll
            for (int n = 0; n < NW; ++n) cf[r][n] = tmp[n];
            cf[r][NW] = (x0 + NW < p.nx) ? p.f[base + x0 + NW] : 0.f;
        }
        unsigned bits = 0u;
        if (any_bc) {
#pragma unroll
            for (int n = 0; n <= NW; ++n) {
                if (x0 + n < p.nx) {
                    bool fx = false;
                    cu[r][n] = apply_bc(p, cu[r][n], b, rowbase + x0 + n, nps, fx);
                    bits |= fx ? (1u << n) : 0u;
                }
            }
        }
        fixed[r] = bits;
    };

    float e_acc = 0.f, sq_acc = 0.f;
    int par = 0;

    // Emit node row `yr` from acc[r] (adds the left neighbour's hand-over for n == 0).
    auto emit_row = [&](int r, int yr, bool owned_row) {
        xch[par][r % P][tid] = acc[r][NW];
        __syncthreads();
        const float left = (tid > 0) ? xch[par][r % P][tid - 1] : 0.f;
        if (owned_row && col_owner) {
            float o[NW];
#pragma unroll
            for (int n = 0; n < NW; ++n) {
                float v = acc[r][n] + (n == 0 ? left : 0.f);
                if (fixed[r] & (1u << n)) v = 0.f;
                if (x0 + n < p.nx) sq_acc = fmaf(v, v, sq_acc);
                o[n] = v * p.out_scale;
            }
            if (p.out) store_run<NW, VEC>(p.out, (int64_t)b * nps + (int64_t)yr * p.nx, x0, p.nx, o);
        }
    };

    load_row(0, ey_begin * P);
    for (int ey = ey_begin; ey < ey_end; ++ey) {
#pragma unroll
        for (int r = 1; r <= P; ++r) load_row(r, ey * P + r);
        const bool own_layer = ey >= ey_own;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (ex0 + e < p.nelx) {
                float lu[NB][NB], ln[NB][NB], lf[NB][NB], g[NB][NB];
#pragma unroll
                for (int jb = 0; jb < NB; ++jb)
#pragma unroll
                    for (int ib = 0; ib < NB; ++ib) {
                        lu[jb][ib] = cu[jb][e * P + ib];
                        ln[jb][ib] = cn[jb][e * P + ib];
                        lf[jb][ib] = cf[jb][e * P + ib];
                    }
                float fg[NGP * NGP];
                if (fmode == F_GP) {
                    const int64_t base = (p.f_batched ? (int64_t)b * eps * (NGP * NGP) : 0) + (int64_t)ey * p.nelx + ex0 + e;
#pragma unroll
                    for (int gi = 0; gi < NGP * NGP; ++gi) fg[gi] = p.fgp[base + (int64_t)gi * eps];
                }
                const float ee = elem2d<P, NGP>(p.T, has_nu, fmode, lu, ln, lf, fg, g);
                if (own_layer && col_owner) e_acc += ee;
#pragma unroll
                for (int jb = 0; jb < NB; ++jb)
#pragma unroll
                    for (int ib = 0; ib < NB; ++ib) acc[jb][e * P + ib] += g[jb][ib];
            }
        }
        // rows 0..P-1 of this element layer are complete now
#pragma unroll
        for (int r = 0; r < P; ++r) emit_row(r, ey * P + r, own_layer);
        par ^= 1;
        // shift: top row becomes the bottom row of the next layer
#pragma unroll
        for (int n = 0; n <= NW; ++n) {
            cu[0][n] = cu[P][n]; cn[0][n] = cn[P][n]; cf[0][n] = cf[P][n];
            acc[0][n] = acc[P][n];
#pragma unroll
            for (int r = 1; r <= P; ++r) acc[r][n] = 0.f;
        }
        fixed[0] = fixed[P];
    }
    // the last strip also owns the top boundary row of the domain
    if (ey_end == p.nely) emit_row(0, p.ny - 1, true);

    if (p.want_energy || p.want_sumsq) {
        const int blk = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        const double es = block_sum((double)e_acc, red, tid, T);
        const double ss = block_sum((double)sq_acc, red, tid, T);
        if (tid == 0) { p.part_energy[blk] = es; p.part_sumsq[blk] = ss; }
    }
}

// =============================================================================================
// 3-D Q1 kernel.  grid = (chunks_x * tiles_y, strips_z, B), block = (TX, TY).
// =============================================================================================
template <int NGP, int E, bool VEC>
__global__ void __launch_bounds__(256) poisson3d_q1_kernel(const PoissonParams p, const int chunks_x) {
    constexpr int NW = E;
    const int TX = blockDim.x, TY = blockDim.y;
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int tid = ty * TX + tx;
    const int chunk = blockIdx.x % chunks_x, tile = blockIdx.x / chunks_x, strip = blockIdx.y, b = blockIdx.z;
    const int q = chunk * (TX - 1) + tx;
    const int ex0 = q * E, x0 = ex0;
    const int ey = tile * (TY - 1) + ty;          // element row == lower node row of this thread
    const bool owner = !(chunk > 0 && tx == 0) && !(tile > 0 && ty == 0);
    const int64_t npl = (int64_t)p.nx * p.ny;     // nodes per plane
    const int64_t nps = npl * p.nz;
    const int64_t eps = (int64_t)p.nelx * p.nely * p.nelz;
    const int R = p.rows_per_strip;
    const int ez_own = strip * R;
    const int ez_begin = ez_own > 0 ? ez_own - 1 : 0;
    const int ez_end = min(ez_own + R, p.nelz);
    const bool has_nu = p.nu != nullptr;
    const int fmode = p.f ? F_NODAL : (p.fgp ? F_GP : F_NONE);
    const bool any_bc = p.bc[0].mask != nullptr || p.bc[1].mask != nullptr;
    const bool row_ok = ey < p.nely;              // thread has real elements
    const bool noderow_ok = ey < p.ny;            // thread's lower node row exists

    // hand-over slots: [parity][slot][thread]; slots: 0 = right (jb0,n=NW), 1..NW = up (jb1,n<NW), NW+1 = up-right
    __shared__ float xch[2][NW + 2][256];
    __shared__ double red[8];

    float cu[2][2][NW + 1], cn[2][2][NW + 1], cf[2][2][NW + 1];   // [plane kb][row jb][n]
    unsigned fixed[2];                                            // Dirichlet bits of (plane kb, row jb = 0)
    float acc[2][2][NW + 1];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        fixed[kb] = 0u;
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int n = 0; n <= NW; ++n) { acc[kb][jb][n] = 0.f; cn[kb][jb][n] = 1.f; cf[kb][jb][n] = 0.f; cu[kb][jb][n] = 0.f; }
    }

    auto load_plane = [&](int kb, int z) {
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            const int y = ey + jb;
            if (y < p.ny) {
                const int64_t rowbase = (int64_t)z * npl + (int64_t)y * p.nx;
                {
                    float tmp[NW];
                    load_run<NW, VEC>(p.u, (int64_t)b * nps + rowbase, x0, p.nx, 0.f, tmp);
#pragma unroll
                    for (int n = 0; n < NW; ++n) cu[kb][jb][n] = tmp[n];
                    cu[kb][jb][NW] = (x0 + NW < p.nx) ? p.u[(int64_t)b * nps + rowbase + x0 + NW] : 0.f;
                }
                if (has_nu) {
                    const int64_t base = (p.nu_batched ? (int64_t)b * nps : 0) + rowbase;
                    float tmp[NW];
                    load_run<NW, VEC>(p.nu, base, x0, p.nx, 0.f, tmp);
#pragma unroll
                    for (int n = 0; n < NW; ++n) cn[kb][jb][n] = tmp[n];
                    cn[kb][jb][NW] = (x0 + NW < p.nx) ? p.nu[base + x0 + NW] : 0.f;
                }
                if (fmode == F_NODAL) {
                    const int64_t base = (p.f_batched ? (int64_t)b * nps : 0) + rowbase;
                    float tmp[NW];
                    load_run<NW, VEC>(p.f, base, x0, p.nx, 0.f, tmp);
#pragma unroll
                    for (int n = 0; n < NW; ++n) cf[kb][jb][n] = tmp[n];
                    cf[kb][jb][NW] = (x0 + NW < p.nx) ? p.f[base + x0 + NW] : 0.f;
                }
                if (any_bc) {
                    unsigned bits = 0u;
#pragma unroll
                    for (int n = 0; n <= NW; ++n) {
                        if (x0 + n < p.nx) {
                            bool fx = false;
                            cu[kb][jb][n] = apply_bc(p, cu[kb][jb][n], b, rowbase + x0 + n, nps, fx);
                            bits |= fx ? (1u << n) : 0u;
                        }
                    }
                    if (jb == 0) fixed[kb] = bits;
                }
            }
        }
    };

    float e_acc = 0.f, sq_acc = 0.f;
    int par = 0;

    // Emit node plane z, row ey (jb = 0), nodes x0..x0+NW-1 from acc[0].
    auto emit_plane = [&](int z, bool owned_plane) {
        xch[par][0][tid] = acc[0][0][NW];
#pragma unroll
        for (int n = 0; n < NW; ++n) xch[par][1 + n][tid] = acc[0][1][n];
        xch[par][NW + 1][tid] = acc[0][1][NW];
        __syncthreads();
        if (owned_plane && owner && noderow_ok) {
            float o[NW];
#pragma unroll
            for (int n = 0; n < NW; ++n) {
                float v = acc[0][0][n];
                if (ty > 0) v += xch[par][1 + n][tid - TX];
                if (n == 0) {
                    if (tx > 0) v += xch[par][0][tid - 1];
                    if (tx > 0 && ty > 0) v += xch[par][NW + 1][tid - TX - 1];
                }
                if (fixed[0] & (1u << n)) v = 0.f;
                if (x0 + n < p.nx) sq_acc = fmaf(v, v, sq_acc);
                o[n] = v * p.out_scale;
            }
            if (p.out) store_run<NW, VEC>(p.out, (int64_t)b * nps + (int64_t)z * npl + (int64_t)ey * p.nx, x0, p.nx, o);
        }
        par ^= 1;
    };

    load_plane(0, ez_begin);
    for (int ez = ez_begin; ez < ez_end; ++ez) {
        load_plane(1, ez + 1);
        const bool own_layer = ez >= ez_own;
        if (row_ok) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                if (ex0 + e < p.nelx) {
                    float lu[2][2][2], ln[2][2][2], lf[2][2][2], g[2][2][2];
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                            for (int ib = 0; ib < 2; ++ib) {
                                lu[kb][jb][ib] = cu[kb][jb][e + ib];
                                ln[kb][jb][ib] = cn[kb][jb][e + ib];
                                lf[kb][jb][ib] = cf[kb][jb][e + ib];
                            }
                    float fg[NGP * NGP * NGP];
                    if (fmode == F_GP) {
                        constexpr int G = NGP * NGP * NGP;
                        const int64_t base = (p.f_batched ? (int64_t)b * eps * G : 0) +
                                             ((int64_t)ez * p.nely + ey) * p.nelx + ex0 + e;
#pragma unroll
                        for (int gi = 0; gi < G; ++gi) fg[gi] = p.fgp[base + (int64_t)gi * eps];
                    }
                    const float ee = elem3d_q1<NGP>(p.T, has_nu, fmode, lu, ln, lf, fg, g);
                    if (own_layer && owner) e_acc += ee;
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                            for (int ib = 0; ib < 2; ++ib) acc[kb][jb][e + ib] += g[kb][jb][ib];
                }
            }
        }
        emit_plane(ez, own_layer);
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int n = 0; n <= NW; ++n) {
                cu[0][jb][n] = cu[1][jb][n]; cn[0][jb][n] = cn[1][jb][n]; cf[0][jb][n] = cf[1][jb][n];
                acc[0][jb][n] = acc[1][jb][n];
                acc[1][jb][n] = 0.f;
            }
        fixed[0] = fixed[1];
    }
    if (ez_end == p.nelz) emit_plane(p.nz - 1, true);

    if (p.want_energy || p.want_sumsq) {
        const int blk = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        const double es = block_sum((double)e_acc, red, tid, TX * TY);
        const double ss = block_sum((double)sq_acc, red, tid, TX * TY);
        if (tid == 0) { p.part_energy[blk] = es; p.part_sumsq[blk] = ss; }
    }
}

// Sum the per-workgroup partials in a fixed order (deterministic) into the two output scalars.
__global__ void __launch_bounds__(256) poisson_finalize_kernel(const double* __restrict__ pe, const double* __restrict__ ps,
                                                               int n, double* energy, double* sumsq) {
    __shared__ double red[8];
    double e = 0.0, s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) { e += pe[i]; s += ps[i]; }
    e = block_sum(e, red, threadIdx.x, 256);
    s = block_sum(s, red, threadIdx.x, 256);
    if (threadIdx.x == 0) {
        if (energy) *energy = e;
        if (sumsq) *sumsq = s;
    }
}

// ---------------------------------------------------------------------------------------------
// host side: launch geometry
// ---------------------------------------------------------------------------------------------
struct Geom2D { int T, E, chunks, strips, R; };
struct Geom3D { int TX, TY, E, chunks, tiles, strips, R; };

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

static int chunks_for(int logical_threads, int T) {
    return logical_threads <= T ? 1 : ceil_div(logical_threads - 1, T - 1);
}

// Pick elements-per-thread, block width and strip height so that (a) the x extent wastes few lanes,
// (b) the launch has >= ~4 workgroups per CU when the problem allows it, (c) the marched strips are
// long enough that the one recomputed layer per strip stays a small fraction.
static Geom2D plan2d(const dn_mesh* m, int P) {
    Geom2D g;
    const int nx = m->nx, nely = (m->ny - 1) / P;
    const int maxE = (P == 1) ? 4 : (P == 2 ? 2 : 1);
    double best = -1.0;
    g.T = 64; g.E = 1;
    for (int E = 1; E <= maxE; E *= 2) {
        const int NW = E * P;
        const int Q = (nx - 1) / NW + 1;                 // logical thread columns
        for (int T = 64; T <= 256; T *= 2) {
            const int chunks = chunks_for(Q, T);
            const double util = (double)Q / ((double)chunks * T);
            // prefer wider per-thread work (fewer hand-overs, vector memory ops) at equal utilisation
            const double score = util + 0.02 * NW + (T == 256 ? 0.005 : 0.0);
            if (score > best) { best = score; g.T = T; g.E = E; g.chunks = chunks; }
        }
    }
    // strips: aim for >= 1024 workgroups in total, R in [4, 64]
    const long long wg_per_strip = (long long)g.chunks * m->batch;
    int R = 32;
    while (R > 4 && wg_per_strip * ceil_div(nely, R) < 1024) R /= 2;
    if (R > nely) R = nely;
    g.R = R < 1 ? 1 : R;
    g.strips = ceil_div(nely, g.R);
    return g;
}

static Geom3D plan3d(const dn_mesh* m) {
    Geom3D g;
    const int nx = m->nx, ny = m->ny, nelz = m->nz - 1;
    double best = -1.0;
    g.TX = 16; g.TY = 16; g.E = 1; g.chunks = 1; g.tiles = 1;
    for (int E = 1; E <= 4; E *= 2) {
        const int Q = (nx - 1) / E + 1;
        for (int TX = 8; TX <= 64; TX *= 2) {
            const int TY = 256 / TX;
            const int chunks = chunks_for(Q, TX);
            const int tiles = chunks_for(ny, TY);
            const double util = ((double)Q / ((double)chunks * TX)) * ((double)ny / ((double)tiles * TY));
            const double score = util + 0.01 * E;
            if (score > best) { best = score; g.TX = TX; g.TY = TY; g.E = E; g.chunks = chunks; g.tiles = tiles; }
        }
    }
    const long long wg_per_strip = (long long)g.chunks * g.tiles * m->batch;
    int R = 32;
    while (R > 4 && wg_per_strip * ceil_div(nelz, R) < 1024) R /= 2;
    if (R > nelz) R = nelz;
    g.R = R < 1 ? 1 : R;
    g.strips = ceil_div(nelz, g.R);
    return g;
}

static long long num_workgroups(const dn_mesh* m) {
    if (m->nsd == 2) {
        Geom2D g = plan2d(m, m->degree);
        return (long long)g.chunks * g.strips * m->batch;
    }
    Geom3D g = plan3d(m);
    return (long long)g.chunks * g.tiles * g.strips * m->batch;
}

static int validate_mesh(const dn_mesh* m) {
    if (!m) return DN_E_BADARG;
    if (m->nsd != 2 && m->nsd != 3) return DN_E_BADARG;
    if (m->degree < 1 || m->degree > 3 || m->ngp < 1 || m->ngp > 4) return DN_E_BADARG;
    if (m->batch < 1 || m->nx < 2 || m->ny < 2 || (m->nsd == 3 && m->nz < 2)) return DN_E_BADARG;
    if ((m->nx - 1) % m->degree || (m->ny - 1) % m->degree || (m->nsd == 3 && (m->nz - 1) % m->degree)) return DN_E_BADARG;
    if (m->batch > 65535) return DN_E_BADARG;
    return 0;
}

template <int P, int NGP, int E>
static void launch2d_vec(const PoissonParams& pp, const Geom2D& g, int batch, bool vec, hipStream_t s) {
    dim3 grid(g.chunks, g.strips, batch), block(g.T);
    if (vec) hipLaunchKernelGGL((poisson2d_kernel<P, NGP, E, true>), grid, block, 0, s, pp);
    else hipLaunchKernelGGL((poisson2d_kernel<P, NGP, E, false>), grid, block, 0, s, pp);
}

template <int P, int NGP>
static int launch2d_e(const PoissonParams& pp, const Geom2D& g, int batch, bool vec, hipStream_t s) {
    if constexpr (P == 1) {
        if (g.E == 4) { launch2d_vec<P, NGP, 4>(pp, g, batch, vec, s); return 0; }
    }
    if constexpr (P <= 2) {
        if (g.E == 2) { launch2d_vec<P, NGP, 2>(pp, g, batch, vec, s); return 0; }
    }
    if (g.E == 1) { launch2d_vec<P, NGP, 1>(pp, g, batch, vec && P > 1, s); return 0; }
    return DN_E_UNSUPPORTED;
}

static int launch2d(const PoissonParams& pp, const Geom2D& g, int P, int ngp, int batch, bool vec, hipStream_t s) {
    switch (P * 10 + ngp) {
        case 12: return launch2d_e<1, 2>(pp, g, batch, vec, s);
        case 13: return launch2d_e<1, 3>(pp, g, batch, vec, s);
        case 14: return launch2d_e<1, 4>(pp, g, batch, vec, s);
        case 23: return launch2d_e<2, 3>(pp, g, batch, vec, s);
        case 24: return launch2d_e<2, 4>(pp, g, batch, vec, s);
        case 33: return launch2d_e<3, 3>(pp, g, batch, vec, s);
        case 34: return launch2d_e<3, 4>(pp, g, batch, vec, s);
        default: return DN_E_UNSUPPORTED;
    }
}

template <int NGP, int E>
static void launch3d_vec(const PoissonParams& pp, const Geom3D& g, int batch, bool vec, hipStream_t s) {
    dim3 grid(g.chunks * g.tiles, g.strips, batch), block(g.TX, g.TY);
    if (vec && E > 1) hipLaunchKernelGGL((poisson3d_q1_kernel<NGP, E, (E > 1)>), grid, block, 0, s, pp, g.chunks);
    else hipLaunchKernelGGL((poisson3d_q1_kernel<NGP, E, false>), grid, block, 0, s, pp, g.chunks);
}

template <int NGP>
static int launch3d_e(const PoissonParams& pp, const Geom3D& g, int batch, bool vec, hipStream_t s) {
    switch (g.E) {
        case 1: launch3d_vec<NGP, 1>(pp, g, batch, vec, s); return 0;
        case 2: launch3d_vec<NGP, 2>(pp, g, batch, vec, s); return 0;
        case 4: launch3d_vec<NGP, 4>(pp, g, batch, vec, s); return 0;
        default: return DN_E_UNSUPPORTED;
    }
}

}  // namespace dn

using namespace dn;

extern "C" int64_t dn_poisson_workspace_bytes(const dn_mesh* mesh) {
    if (validate_mesh(mesh) != 0) return DN_E_BADARG;
    return (int64_t)(2 * sizeof(double)) * num_workgroups(mesh);
}

extern "C" int dn_poisson_apply(const dn_mesh* m, const dn_poisson_args* a, void* stream) {
    int rc = validate_mesh(m);
    if (rc) return rc;
    if (!a || !a->u) return DN_E_BADARG;
    if (a->f && a->f_gp) return DN_E_BADARG;
    if (!a->out && !a->energy && !a->sumsq) return DN_E_BADARG;
    if (m->nsd == 3 && m->degree != 1) return DN_E_UNSUPPORTED;
    const int P = m->degree;
    if (m->ngp < (P == 1 ? 2 : 3)) return DN_E_UNSUPPORTED;
    const bool want_red = a->energy || a->sumsq;
    const long long nwg = num_workgroups(m);
    if (want_red && (!a->workspace || a->workspace_bytes < (int64_t)(2 * sizeof(double)) * nwg)) return DN_E_WORKSPACE;

    PoissonParams pp;
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 4; ++j) {
            pp.T.b[i][j] = m->basis[i][j];
            pp.T.dx[i][j] = m->dbasis[i][j] * m->scale[0];
            pp.T.dy[i][j] = m->dbasis[i][j] * m->scale[1];
            pp.T.dz[i][j] = m->dbasis[i][j] * m->scale[2];
        }
        pp.T.w[i] = m->gpw[i];
        pp.T.wx[i] = m->gpw[i] * a->wscale;
    }
    for (int d = 0; d < 3; ++d) pp.T.hs[d] = 0.5f * m->scale[d];
    pp.T.alpha = a->alpha; pp.T.beta = a->beta; pp.T.c = a->c;
    pp.u = a->u; pp.nu = a->nu; pp.f = a->f; pp.fgp = a->f_gp;
    pp.nu_batched = a->nu_batched; pp.f_batched = a->f_batched;
    for (int k = 0; k < 2; ++k) {
        pp.bc[k].mask = a->bc[k].mask; pp.bc[k].field = a->bc[k].field; pp.bc[k].value = a->bc[k].value;
        pp.bc[k].mask_is_u8 = a->bc[k].mask_is_u8; pp.bc[k].mask_batched = a->bc[k].mask_batched;
        pp.bc[k].field_batched = a->bc[k].field_batched;
    }
    pp.out_scale = a->out_scale; pp.out = a->out;
    pp.part_energy = reinterpret_cast<double*>(a->workspace);
    pp.part_sumsq = pp.part_energy ? pp.part_energy + nwg : nullptr;
    pp.nx = m->nx; pp.ny = m->ny; pp.nz = m->nsd == 3 ? m->nz : 1;
    pp.nelx = (m->nx - 1) / P; pp.nely = (m->ny - 1) / P; pp.nelz = m->nsd == 3 ? (m->nz - 1) / P : 1;
    pp.want_energy = want_red ? 1 : 0; pp.want_sumsq = want_red ? 1 : 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);

    auto aligned = [](const void* p, int bytes) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) % bytes) == 0; };
    if (m->nsd == 2) {
        const Geom2D g = plan2d(m, P);
        pp.rows_per_strip = g.R;
        const int NW = g.E * P;
        const bool vec = (NW == 2 || NW == 4) && (m->nx % NW == 0) && aligned(a->u, 4 * NW) && aligned(a->nu, 4 * NW) &&
                         aligned(a->f, 4 * NW) && aligned(a->out, 4 * NW);
        rc = launch2d(pp, g, P, m->ngp, m->batch, vec, s);
    } else {
        const Geom3D g = plan3d(m);
        pp.rows_per_strip = g.R;
        const int NW = g.E;
        const bool vec = (NW == 2 || NW == 4) && (m->nx % NW == 0) && aligned(a->u, 4 * NW) && aligned(a->nu, 4 * NW) &&
                         aligned(a->f, 4 * NW) && aligned(a->out, 4 * NW);
        switch (m->ngp) {
            case 2: rc = launch3d_e<2>(pp, g, m->batch, vec, s); break;
            case 3: rc = launch3d_e<3>(pp, g, m->batch, vec, s); break;
            case 4: rc = launch3d_e<4>(pp, g, m->batch, vec, s); break;
            default: rc = DN_E_UNSUPPORTED;
        }
    }
    if (rc) return rc;
    DN_LAUNCH_CHECK();
    if (want_red) {
        hipLaunchKernelGGL(poisson_finalize_kernel, dim3(1), dim3(256), 0, s, pp.part_energy, pp.part_sumsq, (int)nwg,
                           a->energy, a->sumsq);
        DN_LAUNCH_CHECK();
    }
    return 0;
}
