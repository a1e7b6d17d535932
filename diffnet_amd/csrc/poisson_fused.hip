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
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "poisson_common.h"

namespace dn {

// =============================================================================================
// 2-D kernel.  grid = (chunks_x, strips_y, B), block = T threads.
//   P: element degree, NGP: 1-D Gauss points, E: elements per thread along x, VEC: vector ld/st legal,
//   FGP: forcing given at Gauss points
// =============================================================================================
template <int P, int NGP, int E, bool VEC, bool FGP>
__global__ void __launch_bounds__(256) poisson2d_kernel(const PoissonParams p) {
    constexpr int NB = P + 1;
    constexpr int NW = E * P;             // nodes owned per thread per node row
    const int T = blockDim.x;
    const int tid = threadIdx.x;
    const int chunk = blockIdx.x, strip = selected_strip(p, (int)blockIdx.y), b = blockIdx.z;
    const int q = chunk * (T - 1) + tid;  // logical thread column (chunks overlap by one thread)
    const int ex0 = q * E;                // first element of this thread
    const int x0 = ex0 * P;               // first node
    const bool col_owner = !(chunk > 0 && tid == 0);
    const int64_t nps = (int64_t)p.nx * p.ny;            // nodes per sample
    const unsigned eps = (unsigned)(p.nelx * p.nely);    // elements per sample
    const SampleBases sb = sample_bases(p, b, nps);
    const float* fgp = FGP ? p.fgp + (p.f_batched ? (int64_t)b * eps * (NGP * NGP) : 0) : nullptr;
    const int R = p.rows_per_strip;
    const int ey_own = strip * R;
    const int ey_begin = ey_own > 0 ? ey_own - 1 : 0;
    const int ey_end = min(ey_own + R, p.nely);
    const bool any_bc = sb.mask[0] != nullptr || sb.mask[1] != nullptr;

    __shared__ float xch[2][P][256];
    __shared__ double red[8];
    __shared__ int last_flag;

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
        const unsigned rowoff = (unsigned)yr * (unsigned)p.nx;
        load_seg<NW, VEC>(sb.u, rowoff, x0, p.nx, cu[r]);
        if (sb.nu) load_seg<NW, VEC>(sb.nu, rowoff, x0, p.nx, cn[r]);
        if (sb.f) load_seg<NW, VEC>(sb.f, rowoff, x0, p.nx, cf[r]);
        fixed[r] = any_bc ? load_apply_bc<NW, VEC>(p, sb, rowoff, x0, cu[r]) : 0u;
    };

    float e1_acc = 0.f, e2_acc = 0.f, sq_acc = 0.f;
    int par = 0;

    // Emit node row `yr` from acc[r] (adds the left neighbour's hand-over for n == 0).
    auto emit_row = [&](int r, int yr, bool owned_row) {
        xch[par][r % P][tid] = acc[r][NW];
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // LDS-only barrier: __syncthreads() would also drain the loads in flight
        const float left = (tid > 0) ? xch[par][r % P][tid - 1] : 0.f;
        if (owned_row && col_owner) {
            float o[NW];
#pragma unroll
            for (int n = 0; n < NW; ++n) {
                float v = acc[r][n] + (n == 0 ? left : 0.f);
                v = (fixed[r] & (1u << n)) ? 0.f : v;
                sq_acc = (x0 + n < p.nx) ? fmaf(v, v, sq_acc) : sq_acc;
                o[n] = v * p.out_scale;
            }
            if (sb.out) store_seg<NW, VEC>(sb.out, (unsigned)yr * (unsigned)p.nx, x0, p.nx, o);
        }
    };

    load_row(0, ey_begin * P);
    for (int ey = ey_begin; ey < ey_end; ++ey) {
#pragma unroll
        for (int r = 1; r <= P; ++r) load_row(r, ey * P + r);
        const bool own_layer = ey >= ey_own;
        const bool count = own_layer && col_owner;
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
                if constexpr (FGP) {
                    const unsigned eo = (unsigned)ey * (unsigned)p.nelx + (unsigned)(ex0 + e);
#pragma unroll
                    for (int gi = 0; gi < NGP * NGP; ++gi) fg[gi] = fgp[eo + (unsigned)gi * eps];
                }
                float e1, e2;
                elem2d<P, NGP, FGP>(p.T, lu, ln, lf, fg, g, e1, e2);
                e1_acc += count ? e1 : 0.f;
                e2_acc += count ? e2 : 0.f;
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

    if (p.want_sums) finish_sums(p, e1_acc, e2_acc, sq_acc, tid, T, red, &last_flag);
}

}  // namespace dn

namespace dn {

// ---------------------------------------------------------------------------------------------
// host side: launch geometry
// ---------------------------------------------------------------------------------------------

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// workspace layout: [arrival counters][partial energies: nwg doubles][partial sumsq: nwg doubles]

static int chunks_for(int logical_threads, int T) {
    return logical_threads <= T ? 1 : ceil_div(logical_threads - 1, T - 1);
}

// Pick elements-per-thread, block width and strip height so that (a) the x extent wastes few lanes,
// (b) the launch has >= ~4 workgroups per CU when the problem allows it, (c) the marched strips are
// long enough that the one recomputed layer per strip stays a small fraction.
// allow_ua: the launch goes to the closed-form Q1 kernel, which also takes rows of 4 k + 1 nodes (the 2^n + 1 meshes) four elements per thread: unaligned
// 16-byte row accesses, the last node column owned by the last thread column -- the geometry of the 4 k mesh (poisson2d_q1_cf.hip, CF_UA)
static Geom2D plan2d(const dn_mesh* m, int P, bool allow_e4 = true, bool chain_ok = false, bool allow_ua = false) {
    Geom2D g;
    const int nx = m->nx, nely = (m->ny - 1) / P;
    // (from 321 nodes per row on: below that the narrower geometry costs more in strip seams than the vector accesses save -- 257^2 x 64: 26.2 against
    // 23.4 us, 321^2: 42.0 against 45.7, 513^2: 76 against 108, 1025^2 x 15: 63.5 against 85.6; tools/time_2d_sizes.py, profiles/r3_2d_widths.txt)
    const bool ua = allow_ua && P == 1 && (nx & 3) == 1 && nx >= 321;
    const int maxE = (P == 1) ? 4 : (P == 2 ? 2 : 1);
    const int minE = (P == 1) ? 2 : 1;                  // the Q1 marching kernel is built for E = 2 and 4
    double best = -1.0;
    g.T = 64; g.E = minE; g.chunks = 1;
    for (int E = minE; E <= maxE; E *= 2) {
        const int NW = E * P;
        if (ua && E != 4) continue;
        if (P == 1 && E == 4 && !ua && (nx % 4 != 0 || !allow_e4)) continue;   // E = 4 exists only with aligned vector rows (or as the CF_UA form)
        const int Q = ua ? (nx - 1) / NW : (nx - 1) / NW + 1;                   // logical thread columns
        for (int T = 64; T <= 256; T += 64) {            // whole waves: 64, 128, 192, 256 threads
            const int chunks = chunks_for(Q, T);
            const double util = (double)Q / ((double)chunks * T);
            // prefer wider per-thread work (fewer hand-overs, vector memory ops) at equal utilisation
            const double score = util + 0.02 * NW + 0.0003 * T;      // at equal utilisation: wider workgroups (fewer seam columns and dispatches)
            if (score > best) { best = score; g.T = T; g.E = E; g.chunks = chunks; }
        }
    }
    // strips: aim for >= 4 waves per SIMD in total (4096 waves), R in [4, 32]
    const long long waves_per_strip = (long long)g.chunks * m->batch * (g.T / 64);
    int R = 32;
    while (R > 4 && waves_per_strip * ceil_div(nely, R) < 4096) R /= 2;
    if (R > nely) R = nely;
    g.R = R < 1 ? 1 : R;
    g.strips = ceil_div(nely, g.R);
    // The closed-form Q1 kernel can CHAIN W neighbouring strips per workgroup (poisson2d_q1_cf.hip, "PLAN2D" "T,E,R,W"): a workgroup then
    // reads W R + 2 node rows instead of W (R + 2) and recomputes one seam layer instead of W.
    //  * Launches that fill the chip at 16-row strips (the bench workload): measured traffic goes from 1.107x to 1.033x of the algorithmic
    //    bytes, but the launch does not get faster -- its length is the 17 dependent row trips of a strip either way (the chained strips wait
    //    for their lower neighbour's last layer instead of recomputing it), and the hand-overs cost 1-2 us (profiles/r3_rotate_chain.txt:
    //    56.0 / 57.1 / 58.0 / 61.9 us for W = 1 / 2 / 4 / 8 on batches in rotation): one strip per workgroup.
    //  * SMALL launches (cfg2 at B = 1 .. 24), which must cut the rows into strips of 4-8 to get enough waves and then pay 25-50 % for the seam
    //    layers and halo rows: chained strips of 2-4 rows win 12-18 % at B <= 4 and 2-5 % up to B = 24 (profiles/r3_plan2d_small.txt).  Strip
    //    height: the smallest of 2, 4, 8 that keeps the launch within 1.5 rounds of waves.
    g.W = 1;
    g.ua = ua;
    const int cw = poisson2d_q1_cf_chain();
    if (chain_ok && !ua && P == 1 && cw > 1 && g.E == 4 && g.T == 128 && g.R < 16) {
        int Rc = 2;
        while (Rc < 16 && waves_per_strip * ceil_div(nely, Rc) > 6144) Rc *= 2;
        if (Rc < 16 && ceil_div(nely, Rc) >= cw) { g.R = Rc; g.strips = ceil_div(nely, Rc); g.W = cw; }
    }
    // (Measured and not adopted, profiles/r3_2d_graded_launch.txt: giving the workgroups that are dispatched first -- they finish 4 us before the last
    // ones, tools/timeline2d.py -- more rows, either as taller strips on a strip-major grid or as one strip fewer in the first samples: 57-61 us
    // against 56.1; a launch lasts as long as its tallest strip's row trips.)
    return g;
}

static Geom3D plan3d(const dn_mesh* m, bool allow_e2 = false) {
    Geom3D g;
    const int nx = m->nx, ny = m->ny, nelz = m->nz - 1;
    if (m->ngp == 2 && allow_e2) {
        // node-owner form with two elements per thread (poisson3d_q1n2_kernel): tiles of 32 x 16 elements, 3 workgroups per CU (<= 168 VGPRs)
        g.TX = 16; g.TY = 16; g.E = 2;
        g.chunks = chunks_for(nx / 2, 16);
        g.tiles = chunks_for(ny, 16);
        const double cap = 256.0 * 3.0;
        const long long wg_per_strip = (long long)g.chunks * g.tiles * m->batch;
        double bestc = 1e300;
        g.R = nelz < 1 ? 1 : nelz;
        for (int R = 4; R <= 64 && R <= (nelz < 4 ? 4 : nelz); ++R) {
            const int strips = ceil_div(nelz, R);
            const double rounds = std::max(1.0, std::ceil(2.0 * (double)(wg_per_strip * strips) / cap) / 2.0);
            const double layers = (strips == 1 ? nelz : R + 1) + 1.5;
            const double cost = rounds * layers * (1.0 + 0.002 * R);
            if (cost < bestc) { bestc = cost; g.R = R; }
        }
        if (g.R > nelz) g.R = nelz < 1 ? 1 : nelz;
        g.strips = ceil_div(nelz, g.R);
        return g;
    }
    if (m->ngp == 2) {
        // 2 x 2 x 2 points: one element per thread in tiles 16 threads wide (the T16 form of poisson3d_q1w_kernel: DPP hand-over
        // along x, paired loads, next plane's loads in flight).  Tile height: the multiple of 4 rows (whole waves) that wastes
        // the fewest thread rows; strip height: fewest (half-)rounds of resident workgroups x layers per workgroup -- a
        // workgroup costs its R layers + 1 recomputed seam layer + ~1.5 layers of start-up, and 256 CUs hold 5 workgroups each.
        g.TX = 16; g.E = 1;
        g.chunks = chunks_for(nx, 16);
        double best = -1.0;
        g.TY = 16;
        for (int TY = 4; TY <= 16; TY += 4) {
            const int tiles = chunks_for(ny, TY);
            const double util = (double)ny / ((double)tiles * TY) + 0.002 * TY;
            if (util > best) { best = util; g.TY = TY; g.tiles = tiles; }
        }
        const double cap = 256.0 * 5.0 * (256.0 / (16.0 * g.TY)) ;          // resident workgroups (waves per SIMD bound)
        const long long wg_per_strip = (long long)g.chunks * g.tiles * m->batch;
        double bestc = 1e300;
        g.R = nelz < 1 ? 1 : nelz;
        for (int R = 4; R <= 64 && R <= (nelz < 4 ? 4 : nelz); ++R) {
            const int strips = ceil_div(nelz, R);
            const double rounds = std::max(1.0, std::ceil(2.0 * (double)(wg_per_strip * strips) / cap) / 2.0);
            const double layers = (strips == 1 ? nelz : R + 1) + 1.5;
            const double cost = rounds * layers * (1.0 + 0.002 * R);              // ties: shorter strips (smaller tail)
            if (cost < bestc) { bestc = cost; g.R = R; }
        }
        if (g.R > nelz) g.R = nelz < 1 ? 1 : nelz;
        g.strips = ceil_div(nelz, g.R);
        return g;
    }
    const int maxE = 1;                                  // larger rules: one element per thread (register budget)
    double best = -1.0;
    g.TX = 16; g.TY = 16; g.E = 1; g.chunks = 1; g.tiles = 1;
    // Tile shape: what counts is the fraction of thread slots doing useful work -- chunks and tiles overlap by one thread
    // column / row and the last ones are partly empty.  Any TX is legal (a wave may span rows); powers of two get a small
    // bonus (measured: equal utilisation runs a few per cent faster with aligned rows).
    for (int E = 1; E <= maxE; E *= 2) {
        const int Q = (nx - 1) / E + 1;
        for (int TX = 8; TX <= 128; ++TX) {
            const int TY = 256 / TX;
            if (TX * TY < 192 || TY < 2) continue;
            const int chunks = chunks_for(Q, TX);
            const int tiles = chunks_for(ny, TY);
            const double util = ((double)Q / ((double)chunks * TX)) * ((double)ny / ((double)tiles * TY));
            const bool pow2 = (TX & (TX - 1)) == 0;
            const double score = util + 0.05 * E + (pow2 ? 0.04 + 0.0005 * TX : 0.0);
            if (score > best) { best = score; g.TX = TX; g.TY = TY; g.E = E; g.chunks = chunks; g.tiles = tiles; }
        }
    }
    // Strip height: >= 2048 workgroups when the mesh allows it, but never fewer than 8 layers per strip (one layer is
    // recomputed per strip, and a workgroup's start-up costs about as much as a layer).
    const long long wg_per_strip = (long long)g.chunks * g.tiles * m->batch;
    int R = 32;
    while (R > 8 && wg_per_strip * ceil_div(nelz, R) < 2048) R /= 2;
    if (R > nelz) R = nelz;
    g.R = R < 1 ? 1 : R;
    g.strips = ceil_div(nelz, g.R);
    return g;
}

// dn_config_set("PLAN2D", "T,E,R") / ("PLAN3D", "TX,TY,E,R") override the launch geometry (tuning experiments only).
static Geom2D plan2d_env(const dn_mesh* m, int P, bool allow_e4 = true, bool chain_ok = false, bool allow_ua = false) {
    Geom2D g = plan2d(m, P, allow_e4, chain_ok, allow_ua);
    const char* e = config(CFG_PLAN2D);
    int T, E, R, W = 0;
    if (e && sscanf(e, "%d,%d,%d,%d", &T, &E, &R, &W) >= 3 && T >= 64 && T <= 256 && (E == 1 || E == 2 || E == 4) && R >= 1 &&
        !(P == 1 && (E == 1 || (E == 4 && !g.ua && (m->nx % 4 != 0 || !allow_e4)) || (g.ua && E != 4)))) {
        const int nely = (m->ny - 1) / P;
        g.T = T; g.E = E; g.R = R > nely ? nely : R;
        g.chunks = chunks_for(g.ua ? (m->nx - 1) / 4 : (m->nx - 1) / (E * P) + 1, T);
        g.strips = ceil_div(nely, g.R);
        const int cw = poisson2d_q1_cf_chain();        // "T,E,R,W": W = 1 (or absent) one strip per workgroup, W >= 2 the library's chain length where it applies
        g.W = (W >= 2 && !g.ua && cw > 1 && P == 1 && E == 4 && T == 128 && g.R + 3 <= 64 && g.R >= 2) ? cw : 1;
    }
    return g;
}

static Geom3D plan3d_env(const dn_mesh* m, bool allow_e2 = false) {
    Geom3D g = plan3d(m, allow_e2);
    const char* e = config(CFG_PLAN3D);
    int TX, TY, E, R;
    if (e && sscanf(e, "%d,%d,%d,%d", &TX, &TY, &E, &R) == 4 && TX * TY >= 64 && (E == 1 || (E == 2 && allow_e2 && TX == 16 && TY == 16)) && R >= 1 &&
        TX * TY <= 256) {
        const int nelz = m->nz - 1;
        g.TX = TX; g.TY = TY; g.E = E; g.R = R > nelz ? nelz : R;
        g.chunks = chunks_for((m->nx - 1) / E + 1, TX);
        g.tiles = chunks_for(m->ny, TY);
        g.strips = ceil_div(nelz, g.R);
    }
    return g;
}

static long long num_workgroups(const dn_mesh* m, bool allow_e4 = true) {
    if (m->nsd == 2) {
        const Geom2D g = plan2d_env(m, m->degree, allow_e4);
        const Geom2D gu = plan2d_env(m, m->degree, allow_e4, false, true);          // (the CF_UA form of 4 k + 1 wide meshes has its own geometry)
        return std::max((long long)g.chunks * g.strips, (long long)gu.chunks * gu.strips) * m->batch;
    }
    // 3-D: the launch is either the one-element forms or, where q1n2_ok() holds, the two-element node-owner form with its own tiling and strip
    // height (and its own "PLAN3D" overrides): the partial-sum arrays are laid out for whichever has more workgroups
    const Geom3D g = plan3d_env(m, false);
    long long n = (long long)g.chunks * g.tiles * g.strips * m->batch;
    if (m->ngp == 2) {
        const Geom3D g2 = plan3d_env(m, true);
        n = std::max(n, (long long)g2.chunks * g2.tiles * g2.strips * m->batch);
    }
    return n;
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
    const bool fgp = pp.fgp != nullptr;
    constexpr bool CANVEC = (E * P == 2 || E * P == 4);
    const bool v = vec && CANVEC;
    if constexpr (P == 1) {
        (void)v;   // Q1 goes through launch_poisson2d_q1_g* (see launch2d)
    } else {
        if (v) {
            if (fgp) hipLaunchKernelGGL((poisson2d_kernel<P, NGP, E, CANVEC, true>), grid, block, 0, s, pp);
            else hipLaunchKernelGGL((poisson2d_kernel<P, NGP, E, CANVEC, false>), grid, block, 0, s, pp);
        } else {
            if (fgp) hipLaunchKernelGGL((poisson2d_kernel<P, NGP, E, false, true>), grid, block, 0, s, pp);
            else hipLaunchKernelGGL((poisson2d_kernel<P, NGP, E, false, false>), grid, block, 0, s, pp);
        }
    }
}

template <int P, int NGP>
static int launch2d_e(const PoissonParams& pp, const Geom2D& g, int batch, bool vec, hipStream_t s) {
    if constexpr (P == 1) {
        if (g.E == 4) { launch2d_vec<P, NGP, 4>(pp, g, batch, vec, s); return 0; }
    }
    if constexpr (P <= 2) {
        if (g.E == 2) { launch2d_vec<P, NGP, 2>(pp, g, batch, vec, s); return 0; }
    }
    if (g.E == 1) { launch2d_vec<P, NGP, 1>(pp, g, batch, vec, s); return 0; }
    return DN_E_UNSUPPORTED;
}

static int launch2d(const PoissonParams& pp, const Geom2D& g, int P, int ngp, int batch, bool vec, hipStream_t s) {
    // Q1 with nodal (or absent) forcing: closed-form element kernel, cost independent of the number of Gauss points;
    // forcing given at the Gauss points goes through the per-rule marching kernels
    const bool force_rule = config(CFG_Q1_RULE_KERNEL) != nullptr;              // A/B switch (dn_config_set)
    if (P == 1 && pp.fgp == nullptr && !force_rule) return launch_poisson2d_q1_cf(pp, g, batch, vec, s);
    switch (P * 10 + ngp) {
        case 12: return launch_poisson2d_q1_g2(pp, g, batch, vec, s);
        case 13: return launch_poisson2d_q1_g3(pp, g, batch, vec, s);
        case 14: return launch_poisson2d_q1_g4(pp, g, batch, vec, s);
        case 23: return launch2d_e<2, 3>(pp, g, batch, vec, s);
        case 24: return launch2d_e<2, 4>(pp, g, batch, vec, s);
        case 33: return launch2d_e<3, 3>(pp, g, batch, vec, s);
        case 34: return launch2d_e<3, 4>(pp, g, batch, vec, s);
        default: return DN_E_UNSUPPORTED;
    }
}

}  // namespace dn

// May the launch of (mesh, args) run the 3-D node-owner kernel with two elements per thread?  (exact 2-point rule, even nx, 8-byte aligned
// node pairs, nodal / absent forcing, constant-value conditions held as images of one kind)
static bool q1n2_ok(const dn_mesh* m, const dn_poisson_args* a) {
    if (m->nsd != 3 || m->degree != 1 || m->ngp != 2 || (m->nx & 1) || a->f_gp) return false;
    if (m->gpw[0] != 1.0f || m->gpw[1] != 1.0f) return false;
    if (dn::config(dn::CFG_Q1_3D_T16) != nullptr || dn::config(dn::CFG_Q1_3D_E1) != nullptr) return false;
    auto al = [](const void* p, uintptr_t bytes) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) % bytes) == 0; };
    if (!al(a->u, 8) || !al(a->nu, 8) || !al(a->f, 8) || !al(a->out, 8)) return false;
    int kinds = 0;
    for (int k = 0; k < 2; ++k) {
        const dn_dirichlet& d = a->bc[k];
        if (!d.mask) continue;
        if (d.field) return false;
        if (d.mask_kind == DN_MASK_U8) { if (!al(d.mask, 2)) return false; kinds |= 1; }
        else if (d.mask_kind == DN_MASK_F32) { if (!al(d.mask, 8)) return false; kinds |= 2; }
        else return false;
    }
    return kinds != 3;
}

namespace dn {
// Final reduction of a launch's per-workgroup partial sums as its own (one-workgroup) kernel: fixed order -- thread t adds partials
// t, t + 256, ... in turn, then the fixed-order block sum -- so the scalars are bitwise repeatable.  Same output semantics as the in-kernel
// reduction (finish_sums in poisson_common.h).
__global__ void __launch_bounds__(256) poisson_finish_sums_kernel(const double* __restrict__ part_energy, const double* __restrict__ part_sumsq, int n,
                                                                  double* energy, double* sumsq, float* energy_f32, double energy_scale, int acc) {
    __shared__ double red[2 * 4];
    double e = 0.0, s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) { e += part_energy[i]; s += part_sumsq[i]; }
    block_sum2(e, s, red, (int)threadIdx.x, 256);
    if (threadIdx.x == 0) {
        if (acc) { e += *energy; s += *sumsq; }
        if (energy) *energy = e;
        if (sumsq) *sumsq = s;
        if (energy_f32) *energy_f32 = (float)(e * energy_scale);
    }
}
}  // namespace dn

using namespace dn;

extern "C" int64_t dn_poisson_workspace_bytes(const dn_mesh* mesh) {
    if (validate_mesh(mesh) != 0) return DN_E_BADARG;
    if (mesh->nsd == 3 && mesh->degree > 1) {             // Q2 / Q3 in 3-D: partial sums of its two kernels + the element vectors
        long long n1, n2, ef;
        gen3d_layout(mesh, n1, n2, ef);
        return DN_WS_HEADER + (int64_t)sizeof(double) * (n1 + n2) + (int64_t)sizeof(float) * ef;
    }
    const long long n = std::max(num_workgroups(mesh, true), num_workgroups(mesh, false));   // either launch plan fits
    return DN_WS_HEADER + (int64_t)(2 * sizeof(double)) * n;
}

static long long launched_workgroups(const dn_mesh* m, const dn_poisson_args* a);

// Vector loads / stores of NW nodes are legal when every row segment of every array of the call starts NW-element aligned.  ONE definition:
// dn_poisson_apply picks the launch geometry with it, dn_poisson_finish_sums / launched_workgroups must arrive at the same geometry (the
// stride between the two partial-sum arrays is the launch's workgroup count).
static bool poisson_vec_ok(const dn_mesh* m, const dn_poisson_args* a, int NW) {
    auto aligned = [](const void* p, int bytes) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) % bytes) == 0; };
    bool ok = (NW == 2 || NW == 4) && (m->nx % NW == 0) && aligned(a->u, 4 * NW) && aligned(a->nu, 4 * NW) && aligned(a->f, 4 * NW) &&
              aligned(a->out, 4 * NW);
    for (int k = 0; k < 2; ++k)
        if (a->bc[k].mask_kind == DN_MASK_F32 || a->bc[k].mask_kind == DN_MASK_U8)
            ok = ok && aligned(a->bc[k].mask, a->bc[k].mask_kind == DN_MASK_U8 ? NW : 4 * NW) && aligned(a->bc[k].field, 4 * NW);
    return ok;
}

extern "C" int dn_poisson_apply(const dn_mesh* m, const dn_poisson_args* a, void* stream) {
    int rc = validate_mesh(m);
    if (rc) return rc;
    if (!a || !a->u) return DN_E_BADARG;
    if (a->f && a->f_gp) return DN_E_BADARG;
    if (!a->out && !a->energy && !a->sumsq && !a->energy_f32) return DN_E_BADARG;
    const int P = m->degree;
    const bool gen3d = m->nsd == 3 && P != 1;            // Q2 / Q3 in 3-D: poisson3d_gen.hip (needs its workspace even without sums)
    if (m->ngp < (P == 1 ? 2 : 3)) return DN_E_UNSUPPORTED;
    bool packed_bc = false;                   // bit-packed / geometry-derived conditions: 2-D Q1, nodal or absent forcing, constant values
    for (int k = 0; k < 2; ++k) {
        const dn_dirichlet& d = a->bc[k];
        if (d.mask_kind < DN_MASK_F32 || d.mask_kind > DN_MASK_BOX) return DN_E_BADARG;
        const bool present = d.mask_kind == DN_MASK_BOX ? d.box_faces != 0 : d.mask != nullptr;
        if (present && (d.mask_kind == DN_MASK_BITS || d.mask_kind == DN_MASK_BOX)) {
            packed_bc = true;
            if (d.field) return DN_E_BADARG;
            if (d.mask_kind == DN_MASK_BITS && d.row_words < (m->nx + 31) / 32) return DN_E_BADARG;
        }
    }
    if (packed_bc && m->nsd == 3) {
        // 3-D (round 4): box faces -- no bit arrays -- are taken by the two-element node-owner kernel, alone or beside mask images
        for (int k = 0; k < 2; ++k)
            if (a->bc[k].mask_kind == DN_MASK_BITS) return DN_E_UNSUPPORTED;
        if (gen3d || !q1n2_ok(m, a)) return DN_E_UNSUPPORTED;
    } else if (packed_bc) {
        if (m->nsd != 2 || P != 1 || a->f_gp || config(CFG_Q1_RULE_KERNEL) != nullptr) return DN_E_UNSUPPORTED;
        for (int k = 0; k < 2; ++k) {         // the compact form does not mix with per-node mask images
            const dn_dirichlet& d = a->bc[k];
            if (d.mask && (d.mask_kind == DN_MASK_F32 || d.mask_kind == DN_MASK_U8)) return DN_E_UNSUPPORTED;
        }
    }
    const bool want_red = a->energy || a->sumsq || a->energy_f32;
    auto vec_ok = [&](int NW) { return poisson_vec_ok(m, a, NW); };
    const bool allow_e4 = vec_ok(4);
    const long long nwg = gen3d ? 1 : num_workgroups(m, allow_e4);
    if (nwg >= (1ll << 31)) return DN_E_UNSUPPORTED;                  // the 3-D launch is a 1-D grid
    if (!gen3d && want_red && (!a->workspace || a->workspace_bytes < DN_WS_HEADER + (int64_t)(2 * sizeof(double)) * nwg)) return DN_E_WORKSPACE;
    if ((int64_t)m->nx * m->ny * (m->nsd == 3 ? m->nz : 1) >= (1ll << 30)) return DN_E_UNSUPPORTED;   // 32-bit in-sample offsets

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
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) pp.T.w2[i][j] = pp.T.w[i] * pp.T.wx[j];
    for (int d = 0; d < 3; ++d) {
        pp.T.hs[d] = 0.5f * m->scale[d];
        pp.T.ahs[d] = a->alpha * pp.T.hs[d];
    }
    pp.T.q1c[0] = a->alpha * pp.T.hs[0] * pp.T.hs[0]; pp.T.q1c[1] = a->alpha * pp.T.hs[1] * pp.T.hs[1];
    pp.T.q1c[2] = pp.T.hs[0] * pp.T.hs[0];            pp.T.q1c[3] = pp.T.hs[1] * pp.T.hs[1];
    {   // moments of the 1-D rule against b = phi_1 (Q1 marching kernels), accumulated in double
        double mm[4] = {0.0, 0.0, 0.0, 0.0};
        for (int g = 0; g < m->ngp; ++g) {
            const double bb = m->basis[g][1];
            mm[0] += m->gpw[g]; mm[1] += m->gpw[g] * bb; mm[2] += m->gpw[g] * bb * bb; mm[3] += m->gpw[g] * bb * bb * bb;
        }
        pp.T.q1my[0] = (float)(mm[0] - 2.0 * mm[1] + mm[2]); pp.T.q1my[1] = (float)(mm[1] - mm[2]); pp.T.q1my[2] = (float)mm[2];
        for (int r = 0; r < 3; ++r) pp.T.q1mx[r] = (float)((double)(r == 0 ? mm[0] - 2.0 * mm[1] + mm[2] : (r == 1 ? mm[1] - mm[2] : mm[2])) * (double)a->wscale);
        for (int r = 0; r < 4; ++r) {
            pp.T.m[r] = (float)mm[r];
            pp.T.mxs[r] = (float)(mm[r] * (double)a->wscale);
            if (r < 3)
                for (int g = 0; g < 4; ++g) pp.T.kx[r][g] = (float)(mm[r] * (double)m->gpw[g] * (double)a->wscale);
        }
    }
    for (int g = 0; g < 4; ++g) pp.T.wb[g] = m->gpw[g] * m->basis[g][1];
    for (int d = 0; d < 3; ++d) {
        pp.T.hs2[d] = pp.T.hs[d] * pp.T.hs[d];
        pp.T.kap[d] = a->alpha * a->wscale * pp.T.hs2[d];
    }
    pp.T.m01 = pp.T.m[0] - pp.T.m[1]; pp.T.m12 = pp.T.m[1] - pp.T.m[2];
    pp.T.esc = a->wscale; pp.T.nbw = -a->beta * a->wscale;
    pp.T.alpha = a->alpha; pp.T.beta = a->beta; pp.T.c = a->c;
    pp.u = a->u; pp.nu = a->nu; pp.f = a->f; pp.fgp = a->f_gp;
    pp.nu_batched = a->nu_batched; pp.f_batched = a->f_batched;
    pp.f_is_load = (a->f_is_load && a->f) ? 1 : 0;
    if (pp.f_is_load && !(m->nsd == 3 && !gen3d && q1n2_ok(m, a))) return DN_E_UNSUPPORTED;      // load vectors: the 3-D two-element kernel only (so far)
    for (int k = 0; k < 2; ++k) {
        const dn_dirichlet& d = a->bc[k];
        const bool present = d.mask_kind == DN_MASK_BOX ? d.box_faces != 0 : d.mask != nullptr;
        pp.bc[k].mask = d.mask_kind == DN_MASK_BOX ? nullptr : d.mask; pp.bc[k].field = d.field; pp.bc[k].value = d.value;
        pp.bc[k].mask_is_u8 = d.mask_kind == DN_MASK_U8; pp.bc[k].mask_batched = d.mask_batched;
        pp.bc[k].field_batched = d.field_batched;
        pp.bc[k].kind = present ? d.mask_kind : -1;
        pp.bc[k].box_faces = d.box_faces; pp.bc[k].row_bytes = 4 * d.row_words;
    }
    pp.out_scale = a->out_scale; pp.out = a->out;
    pp.counter = reinterpret_cast<unsigned*>(a->workspace);
    pp.part_energy = a->workspace ? reinterpret_cast<double*>(reinterpret_cast<char*>(a->workspace) + DN_WS_HEADER) : nullptr;
    pp.part_sumsq = pp.part_energy ? pp.part_energy + nwg : nullptr;
    pp.energy = a->energy; pp.sumsq = a->sumsq;
    pp.energy_f32 = a->energy_f32; pp.energy_scale = a->energy_scale;
    pp.nx = m->nx; pp.ny = m->ny; pp.nz = m->nsd == 3 ? m->nz : 1;
    pp.nelx = (m->nx - 1) / P; pp.nely = (m->ny - 1) / P; pp.nelz = m->nsd == 3 ? (m->nz - 1) / P : 1;
    pp.want_sums = want_red ? 1 : 0;
    pp.spin_limit = config(CFG_HANDOVER_SPIN_LIMIT) ? std::atoi(config(CFG_HANDOVER_SPIN_LIMIT)) : 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // fold_prev: this launch's first workgroup closes an earlier evaluation (its partial sums, left with defer_sums, lie in ITS workspace)
    pp.fold_n = 0;
    if (a->fold_prev) {
        const dn_poisson_args* q = a->fold_prev;
        if (gen3d || !q->defer_sums || !q->workspace || !(q->energy || q->sumsq || q->energy_f32)) return DN_E_BADARG;
        if (q->accumulate_sums && !(q->energy && q->sumsq)) return DN_E_BADARG;
        if (q->workspace == a->workspace && want_red) return DN_E_BADARG;        // this launch would overwrite the partials while they are read
        const long long qn = launched_workgroups(m, q), qstride = num_workgroups(m, poisson_vec_ok(m, q, 4));
        if (q->workspace_bytes < DN_WS_HEADER + (int64_t)(2 * sizeof(double)) * qstride || qn > qstride) return DN_E_WORKSPACE;
        const bool cf2d = m->nsd == 2 && P == 1 && a->f_gp == nullptr && config(CFG_Q1_RULE_KERNEL) == nullptr;
        if (!cf2d && !(m->nsd == 3 && q1n2_ok(m, a))) return DN_E_UNSUPPORTED;
        pp.fold_pe = reinterpret_cast<const double*>(reinterpret_cast<const char*>(q->workspace) + DN_WS_HEADER);
        pp.fold_ps = pp.fold_pe + qstride;
        pp.fold_n = (int)qn;
        pp.fold_energy = q->energy; pp.fold_sumsq = q->sumsq; pp.fold_energy_f32 = q->energy_f32;
        pp.fold_scale = q->energy_scale;
        pp.fold_acc = q->accumulate_sums ? 1 : 0;
    }

    // split evaluation (dn_poisson_args.strip_select): a launch over the first and the last strip of the marched axis, or over the rest
    if (a->strip_select < 0 || a->strip_select > 2) return DN_E_BADARG;
    if (a->accumulate_sums && want_red && !(a->energy && a->sumsq)) return DN_E_BADARG;      // the running sums live in the two double slots
    pp.strip_sel = a->strip_select;
    pp.acc_sums = (a->accumulate_sums && want_red && !a->defer_sums) ? 1 : 0;      // (deferred: dn_poisson_finish_sums accumulates)
    pp.defer_sums = (a->defer_sums && want_red) ? 1 : 0;
    auto launched = [&](int total) { return a->strip_select == 1 ? std::min(total, 2) : (a->strip_select == 2 ? std::max(total - 2, 0) : total); };
    if (gen3d) {
        if (a->strip_select != 0 || a->accumulate_sums || a->defer_sums) return DN_E_UNSUPPORTED;      // no split evaluation of this form
        rc = launch_poisson3d_gen(pp, m, a->workspace, a->workspace_bytes, s);
        if (rc) return rc;
        DN_LAUNCH_CHECK();
        return 0;
    }
    if (m->nsd == 2) {
        // chained strips: only the closed-form Q1 kernel has them, and a split evaluation selects whole strips
        const bool chain_ok = P == 1 && a->f_gp == nullptr && config(CFG_Q1_RULE_KERNEL) == nullptr && a->strip_select == 0;
        const bool cf_kernel = P == 1 && a->f_gp == nullptr && config(CFG_Q1_RULE_KERNEL) == nullptr;      // (launch2d)
        Geom2D g = plan2d_env(m, P, allow_e4, chain_ok, cf_kernel);
        if (!chain_ok) g.W = 1;                          // (a "PLAN2D" override may ask for chained strips where they do not exist)
        pp.rows_per_strip = g.R;
        pp.nstrips = g.strips;
        g.strips = launched(g.strips);
        if (g.strips == 0) return 0;                         // fewer than three strips: the other launch did everything
        // (the partial-sum arrays were laid out for nwg workgroups: never launch more than that)
        if (want_red && (long long)g.chunks * ((g.strips + g.W - 1) / g.W) * m->batch > nwg) return DN_E_WORKSPACE;
        const int NW = g.E * P;
        const bool vec = vec_ok(NW);
        rc = launch2d(pp, g, P, m->ngp, m->batch, vec, s);
    } else {
        Geom3D g = plan3d_env(m, q1n2_ok(m, a));
        pp.rows_per_strip = g.R;
        pp.nstrips = g.strips;
        g.strips = launched(g.strips);
        if (g.strips == 0) return 0;
        if (want_red && (long long)g.chunks * g.tiles * g.strips * m->batch > nwg) return DN_E_WORKSPACE;
        const int NW = g.E;
        const bool vec = vec_ok(NW);
        switch (m->ngp) {
            case 2: rc = launch_poisson3d_q1_g2(pp, g, m->batch, vec, s); break;
            case 3: rc = launch_poisson3d_q1_g3(pp, g, m->batch, vec, s); break;
            case 4: rc = launch_poisson3d_q1_g4(pp, g, m->batch, vec, s); break;
            default: rc = DN_E_UNSUPPORTED;
        }
    }
    if (rc) return rc;
    DN_LAUNCH_CHECK();
    return 0;
}

// Workgroups the launch of (mesh, args) consists of = number of per-workgroup partial sums it leaves in the workspace.
static long long launched_workgroups(const dn_mesh* m, const dn_poisson_args* a) {
    const bool e4 = poisson_vec_ok(m, a, 4);
    auto sel = [&](int total) { return a->strip_select == 1 ? std::min(total, 2) : (a->strip_select == 2 ? std::max(total - 2, 0) : total); };
    if (m->nsd == 2) {
        const bool chain_ok = m->degree == 1 && a->f_gp == nullptr && config(CFG_Q1_RULE_KERNEL) == nullptr && a->strip_select == 0;
        const bool cf_kernel = m->degree == 1 && a->f_gp == nullptr && config(CFG_Q1_RULE_KERNEL) == nullptr;
        Geom2D g = plan2d_env(m, m->degree, e4, chain_ok, cf_kernel);
        if (!chain_ok) g.W = 1;
        return (long long)g.chunks * ((sel(g.strips) + g.W - 1) / g.W) * m->batch;
    }
    Geom3D g = plan3d_env(m, q1n2_ok(m, a));
    return (long long)g.chunks * g.tiles * sel(g.strips) * m->batch;
}

extern "C" int dn_poisson_finish_sums(const dn_mesh* m, const dn_poisson_args* a, void* stream) {
    int rc = validate_mesh(m);
    if (rc) return rc;
    if (!a || !a->workspace || !(a->energy || a->sumsq || a->energy_f32)) return DN_E_BADARG;
    if (!a->defer_sums) return DN_E_BADARG;                               // only a launch with defer_sums leaves partials behind the header
    if (m->nsd == 3 && m->degree > 1) return DN_E_UNSUPPORTED;            // poisson3d_gen.hip: own workspace layout, no deferred sums
    if (a->accumulate_sums && !(a->energy && a->sumsq)) return DN_E_BADARG;
    const long long n = launched_workgroups(m, a);
    if (n <= 0) return 0;
    const long long nall = num_workgroups(m, true) > num_workgroups(m, false) ? num_workgroups(m, true) : num_workgroups(m, false);
    if (a->workspace_bytes < DN_WS_HEADER + (int64_t)(2 * sizeof(double)) * nall) return DN_E_WORKSPACE;
    const double* pe = reinterpret_cast<const double*>(reinterpret_cast<const char*>(a->workspace) + DN_WS_HEADER);
    // the partial arrays are laid out for the launch's own workgroup count (dn_poisson_apply: part_sumsq = part_energy + nwg)
    const long long stride = num_workgroups(m, poisson_vec_ok(m, a, 4));
    hipLaunchKernelGGL(poisson_finish_sums_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), pe, pe + stride, (int)n, a->energy,
                       a->sumsq, a->energy_f32, a->energy_scale, (int)(a->accumulate_sums != 0));
    DN_LAUNCH_CHECK();
    return 0;
}
