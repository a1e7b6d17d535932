// dn_poisson_apply for 3-D meshes of Q2 / Q3 elements (any rule of 3 or 4 points per axis): the reference evaluates these with the same
// gauss_pt_evaluation* convolutions as everything else (DiffNetFEM.py:384-465 with fem_basis_deg >= 2); no BASELINE config uses them, so
// this is the plain form of the fused operator, not a tuned one:
//   1. one thread per element: the element's (P+1)^3 nodal values of u (Dirichlet conditions applied), nu and f are parked in LDS, the
//      quadrature runs sum-factorised one x-Gauss point at a time (x-stage, y-stage, z-stage, constitutive law, the three transposes), the
//      element's nodal contributions go to an element-vector buffer in the workspace, its energy into a per-workgroup partial sum;
//   2. one thread per node: the <= 8 element vectors a node takes part in are added in a FIXED order (no atomics: bitwise reproducible),
//      Dirichlet rows are zeroed, the result is scaled and stored, sum(out^2) goes into per-workgroup partial sums;
//   3. one workgroup adds the partial sums in index order.
// Same operator definition as the other kernels (poisson_elem.h: elem2d): W = w_k w_j w_i wscale,
//   out_a = sum W (alpha nu grad N_a . grad u - beta N_a f),  energy = sum W (c nu |grad u|^2 - u f).
#include <hip/hip_runtime.h>

#include "poisson_common.h"

namespace dn {

struct Gen3DParams {
    float* elem;                    // [B][nbf][nel] element vectors (workspace)
    double* part_e;                 // [n1] energy partials of the element kernel
    double* part_s;                 // [n2] sum-of-squares partials of the node kernel
    int n1, n2;
    long long nel, nelem_total;     // elements per sample, elements of the launch
};

__device__ __forceinline__ bool gen_mask_set(const PoissonParams& p, const SampleBases& sb, int k, unsigned off) {
    if (sb.mask[k] == nullptr) return false;
    return p.bc[k].mask_is_u8 ? reinterpret_cast<const uint8_t*>(sb.mask[k])[off] != 0 : reinterpret_cast<const float*>(sb.mask[k])[off] > 0.5f;
}

template <int P, int NGP, bool FGP>
__global__ void __launch_bounds__(64) poisson3d_gen_elem_kernel(const PoissonParams p, const Gen3DParams q) {
    constexpr int NB = P + 1, NBF = NB * NB * NB;
    extern __shared__ float lds[];                 // [3][NBF][64]: u, nu, f of the thread's element
    const int lane = threadIdx.x;
    const long long gid = (long long)blockIdx.x * 64 + lane;
    const bool active = gid < q.nelem_total;
    const long long eid = active ? gid : q.nelem_total - 1;          // inactive lanes repeat the last element, scaled by 0
    const int b = (int)(eid / q.nel);
    const long long el = eid - (long long)b * q.nel;
    const int ex = (int)(el % p.nelx), ey = (int)((el / p.nelx) % p.nely), ez = (int)(el / ((long long)p.nelx * p.nely));
    const int64_t nps = (int64_t)p.nx * p.ny * p.nz;
    const SampleBases sb = sample_bases(p, b, nps);
    float* const U = lds + lane;
    float* const N = lds + NBF * 64 + lane;
    float* const F = lds + 2 * NBF * 64 + lane;
    // The element's nodal values into LDS.  Every optional array is handled in a loop of its own, the wave-uniform test OUTSIDE the loop: the NBF loads of a
    // loop are in flight together (round 4; with the tests inside one loop over the nodes the compiler waited for every load at every branch join -- ~27 x 3
    // memory latencies per element at 2 waves per SIMD: 91 us of the 129^3 Q2 call).  Same values, same order of the Dirichlet conditions.
    const unsigned off0 = ((unsigned)(ez * P) * (unsigned)p.ny + (unsigned)(ey * P)) * (unsigned)p.nx + (unsigned)(ex * P);
    const unsigned sy = (unsigned)p.nx, sz = (unsigned)p.nx * (unsigned)p.ny;
    auto node_off = [&](int a) { return off0 + (unsigned)(a / (NB * NB)) * sz + (unsigned)((a / NB) % NB) * sy + (unsigned)(a % NB); };
#pragma unroll
    for (int a = 0; a < NBF; ++a) U[a * 64] = sb.u[node_off(a)];
    if (sb.nu) {
#pragma unroll
        for (int a = 0; a < NBF; ++a) N[a * 64] = sb.nu[node_off(a)];
    } else {
#pragma unroll
        for (int a = 0; a < NBF; ++a) N[a * 64] = 1.f;
    }
    if (!FGP && sb.f) {
#pragma unroll
        for (int a = 0; a < NBF; ++a) F[a * 64] = sb.f[node_off(a)];
    } else {
#pragma unroll
        for (int a = 0; a < NBF; ++a) F[a * 64] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        if (sb.mask[k] == nullptr) continue;
        unsigned bits = 0u;          // NBF <= 64 nodes: two words
        unsigned bits_hi = 0u;
        if (p.bc[k].mask_is_u8) {
            const uint8_t* mk = reinterpret_cast<const uint8_t*>(sb.mask[k]);
#pragma unroll
            for (int a = 0; a < NBF; ++a) {
                const unsigned bit = mk[node_off(a)] != 0 ? 1u : 0u;
                if (a < 32) bits |= bit << (a & 31); else bits_hi |= bit << (a & 31);
            }
        } else {
            const float* mk = reinterpret_cast<const float*>(sb.mask[k]);
#pragma unroll
            for (int a = 0; a < NBF; ++a) {
                const unsigned bit = mk[node_off(a)] > 0.5f ? 1u : 0u;
                if (a < 32) bits |= bit << (a & 31); else bits_hi |= bit << (a & 31);
            }
        }
        if (sb.field[k]) {
#pragma unroll
            for (int a = 0; a < NBF; ++a) {
                const float v = sb.field[k][node_off(a)];
                const bool set = ((a < 32 ? bits : bits_hi) >> (a & 31)) & 1u;
                U[a * 64] = set ? v : U[a * 64];
            }
        } else {
            const float v = p.bc[k].value;
#pragma unroll
            for (int a = 0; a < NBF; ++a) {
                const bool set = ((a < 32 ? bits : bits_hi) >> (a & 31)) & 1u;
                U[a * 64] = set ? v : U[a * 64];
            }
        }
    }
    const float* fg = nullptr;
    if constexpr (FGP) fg = p.fgp + ((long long)(p.f_batched ? b : 0) * (NGP * NGP * NGP)) * q.nel + el;      // f_gp[b][g][element]

    float g[NB][NB][NB];
#pragma unroll
    for (int kb = 0; kb < NB; ++kb)
#pragma unroll
        for (int jb = 0; jb < NB; ++jb)
#pragma unroll
            for (int ib = 0; ib < NB; ++ib) g[kb][jb][ib] = 0.f;
    float a1 = 0.f, a2 = 0.f;
    const ElemTab& T = p.T;
#pragma unroll
    for (int ig = 0; ig < NGP; ++ig) {
        // x-stage for this x-point: value / x-derivative of u, values of nu and f, per node (kb, jb); cotangents per (kb, jb)
        float xv[NB][NB], xd[NB][NB], xn[NB][NB], xf[NB][NB], sv[NB][NB], sd[NB][NB];
#pragma unroll
        for (int kb = 0; kb < NB; ++kb)
#pragma unroll
            for (int jb = 0; jb < NB; ++jb) {
                float a = 0.f, d = 0.f, n = 0.f, f = 0.f;
#pragma unroll
                for (int ib = 0; ib < NB; ++ib) {
                    const int at = ((kb * NB + jb) * NB + ib) * 64;
                    const float uu = U[at];
                    a = fmaf(T.b[ig][ib], uu, a);
                    d = fmaf(T.dx[ig][ib], uu, d);
                    n = fmaf(T.b[ig][ib], N[at], n);
                    if constexpr (!FGP) f = fmaf(T.b[ig][ib], F[at], f);
                }
                xv[kb][jb] = a; xd[kb][jb] = d; xn[kb][jb] = n; xf[kb][jb] = f; sv[kb][jb] = 0.f; sd[kb][jb] = 0.f;
            }
#pragma unroll
        for (int jg = 0; jg < NGP; ++jg) {
            // y-stage: per node plane kb
            float yv[NB], yx[NB], yy[NB], yn[NB], yf[NB], rv[NB], rx[NB], ry[NB];
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) {
                float v = 0.f, x = 0.f, y = 0.f, n = 0.f, f = 0.f;
#pragma unroll
                for (int jb = 0; jb < NB; ++jb) {
                    v = fmaf(T.b[jg][jb], xv[kb][jb], v);
                    x = fmaf(T.b[jg][jb], xd[kb][jb], x);
                    y = fmaf(T.dy[jg][jb], xv[kb][jb], y);
                    n = fmaf(T.b[jg][jb], xn[kb][jb], n);
                    if constexpr (!FGP) f = fmaf(T.b[jg][jb], xf[kb][jb], f);
                }
                yv[kb] = v; yx[kb] = x; yy[kb] = y; yn[kb] = n; yf[kb] = f; rv[kb] = 0.f; rx[kb] = 0.f; ry[kb] = 0.f;
            }
#pragma unroll
            for (int kg = 0; kg < NGP; ++kg) {
                float val = 0.f, ux = 0.f, uy = 0.f, uz = 0.f, nuv = 0.f, fv = 0.f;
#pragma unroll
                for (int kb = 0; kb < NB; ++kb) {
                    val = fmaf(T.b[kg][kb], yv[kb], val);
                    ux = fmaf(T.b[kg][kb], yx[kb], ux);
                    uy = fmaf(T.b[kg][kb], yy[kb], uy);
                    uz = fmaf(T.dz[kg][kb], yv[kb], uz);
                    nuv = fmaf(T.b[kg][kb], yn[kb], nuv);
                    if constexpr (!FGP) fv = fmaf(T.b[kg][kb], yf[kb], fv);
                }
                if constexpr (FGP) fv = fg[(long long)((kg * NGP + jg) * NGP + ig) * q.nel];
                const float W = T.w[kg] * T.w2[jg][ig];
                const float Wn = W * nuv, Wf = W * fv;
                a1 = fmaf(Wn, fmaf(ux, ux, fmaf(uy, uy, uz * uz)), a1);
                a2 = fmaf(Wf, val, a2);
                const float qx = T.alpha * Wn * ux, qy = T.alpha * Wn * uy, qz = T.alpha * Wn * uz, qv = -T.beta * Wf;
#pragma unroll
                for (int kb = 0; kb < NB; ++kb) {
                    rv[kb] = fmaf(T.b[kg][kb], qv, rv[kb]);
                    rv[kb] = fmaf(T.dz[kg][kb], qz, rv[kb]);
                    rx[kb] = fmaf(T.b[kg][kb], qx, rx[kb]);
                    ry[kb] = fmaf(T.b[kg][kb], qy, ry[kb]);
                }
            }
#pragma unroll
            for (int kb = 0; kb < NB; ++kb)
#pragma unroll
                for (int jb = 0; jb < NB; ++jb) {
                    sv[kb][jb] = fmaf(T.b[jg][jb], rv[kb], sv[kb][jb]);
                    sv[kb][jb] = fmaf(T.dy[jg][jb], ry[kb], sv[kb][jb]);
                    sd[kb][jb] = fmaf(T.b[jg][jb], rx[kb], sd[kb][jb]);
                }
        }
#pragma unroll
        for (int kb = 0; kb < NB; ++kb)
#pragma unroll
            for (int jb = 0; jb < NB; ++jb)
#pragma unroll
                for (int ib = 0; ib < NB; ++ib) {
                    g[kb][jb][ib] = fmaf(T.b[ig][ib], sv[kb][jb], g[kb][jb][ib]);
                    g[kb][jb][ib] = fmaf(T.dx[ig][ib], sd[kb][jb], g[kb][jb][ib]);
                }
    }
    if (active) {
        float* const dst = q.elem + ((long long)b * NBF) * q.nel + el;
#pragma unroll
        for (int kb = 0; kb < NB; ++kb)
#pragma unroll
            for (int jb = 0; jb < NB; ++jb)
#pragma unroll
                for (int ib = 0; ib < NB; ++ib) dst[(long long)((kb * NB + jb) * NB + ib) * q.nel] = g[kb][jb][ib];
    }
    // energy of the workgroup's 64 elements: fixed-order butterfly inside the wave
    double es = active ? (double)p.T.c * (double)a1 - (double)a2 : 0.0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) es += __shfl_xor(es, o, 64);
    if (lane == 0) q.part_e[blockIdx.x] = es;
}

// node kernel: gather the element vectors (fixed order: z, y, x; lower element first), zero Dirichlet rows, scale, store, sum of squares
template <int P>
__global__ void __launch_bounds__(256) poisson3d_gen_node_kernel(const PoissonParams p, const Gen3DParams q, const int batch) {
    constexpr int NB = P + 1, NBF = NB * NB * NB;
    __shared__ double red[4];
    const long long nps = (long long)p.nx * p.ny * p.nz;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    double sq = 0.0;
    if (gid < nps * batch) {
        const int b = (int)(gid / nps);
        const long long n = gid - (long long)b * nps;
        const int x = (int)(n % p.nx), y = (int)((n / p.nx) % p.ny), z = (int)(n / ((long long)p.nx * p.ny));
        const SampleBases sb = sample_bases(p, b, nps);
        // per axis: the node is local node (c - e P) of element e = min(c / P, nel - 1) and, on an element boundary, local node P of element e - 1
        int e0[3], l0[3], cnt[3];
        const int cs[3] = {x, y, z}, nels[3] = {p.nelx, p.nely, p.nelz};
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            int e = cs[d] / P;
            if (e >= nels[d]) e = nels[d] - 1;
            const int l = cs[d] - e * P;
            e0[d] = e; l0[d] = l;
            cnt[d] = (l == 0 && e > 0) ? 2 : 1;
        }
        const float* const src = q.elem + ((long long)b * NBF) * q.nel;
        // the <= 8 element vectors, branch-free: all eight candidates are requested together (clamped, always valid addresses), then added in the fixed
        // order -- lower element first, x fastest -- with the absent ones skipped (round 4: the loops over 1 or 2 elements per axis issued one load at a time)
        float v[8];
        bool ok[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int kx = c & 1, ky = (c >> 1) & 1, kz = c >> 2;               // per axis: 0 = lower element (local node P), 1 = the node's own element
            const bool two_x = cnt[0] == 2, two_y = cnt[1] == 2, two_z = cnt[2] == 2;
            ok[c] = (kx == 1 || two_x) && (ky == 1 || two_y) && (kz == 1 || two_z);
            const int ex = (kx == 0 && two_x) ? e0[0] - 1 : e0[0], lx = (kx == 0 && two_x) ? P : l0[0];
            const int ey = (ky == 0 && two_y) ? e0[1] - 1 : e0[1], ly = (ky == 0 && two_y) ? P : l0[1];
            const int ez = (kz == 0 && two_z) ? e0[2] - 1 : e0[2], lz = (kz == 0 && two_z) ? P : l0[2];
            const long long el = ((long long)ez * p.nely + ey) * p.nelx + ex;
            v[c] = src[(long long)((lz * NB + ly) * NB + lx) * q.nel + el];
        }
        float t = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) t = ok[c] ? t + v[c] : t;
        const unsigned off = (unsigned)n;
        const bool fixed = gen_mask_set(p, sb, 0, off) || gen_mask_set(p, sb, 1, off);
        t = fixed ? 0.f : t;
        sq = (double)t * (double)t;
        if (sb.out) sb.out[off] = t * p.out_scale;
    }
    // fixed-order sum of the workgroup: butterfly per wave, waves in index order
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0) q.part_s[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

__global__ void __launch_bounds__(256) poisson3d_gen_finish_kernel(const PoissonParams p, const Gen3DParams q) {
    __shared__ double re[256], rs[256];
    // the partials are requested eight at a time before any is added (a load-add loop is one memory round trip per partial: 14.3 us of the 129^3 Q2
    // call); same per-thread order of additions
    double e = 0.0, s = 0.0;
    auto strided_sum = [&](const double* part, int n, double& acc) {
        for (int i0 = threadIdx.x; i0 < n; i0 += 256 * 8) {
            double v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = i0 + j * 256;
                v[j] = part[i < n ? i : 0];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += (i0 + j * 256 < n) ? v[j] : 0.0;
        }
    };
    strided_sum(q.part_e, q.n1, e);
    strided_sum(q.part_s, q.n2, s);
    re[threadIdx.x] = e; rs[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { re[threadIdx.x] += re[threadIdx.x + o]; rs[threadIdx.x] += rs[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (p.energy) *p.energy = re[0];
        if (p.sumsq) *p.sumsq = rs[0];
        if (p.energy_f32) *p.energy_f32 = (float)(re[0] * p.energy_scale);
    }
}

template <int P, int NGP>
static int gen3d_launch(const PoissonParams& pp, const Gen3DParams& q, int batch, hipStream_t s) {
    constexpr int NBF = (P + 1) * (P + 1) * (P + 1);
    const size_t lds = (size_t)3 * NBF * 64 * sizeof(float);
    if (pp.fgp) hipLaunchKernelGGL((poisson3d_gen_elem_kernel<P, NGP, true>), dim3((unsigned)q.n1), dim3(64), lds, s, pp, q);
    else hipLaunchKernelGGL((poisson3d_gen_elem_kernel<P, NGP, false>), dim3((unsigned)q.n1), dim3(64), lds, s, pp, q);
    hipLaunchKernelGGL((poisson3d_gen_node_kernel<P>), dim3((unsigned)q.n2), dim3(256), 0, s, pp, q, batch);
    if (pp.want_sums) hipLaunchKernelGGL(poisson3d_gen_finish_kernel, dim3(1), dim3(256), 0, s, pp, q);
    return 0;
}

// workspace of the 3-D Q2 / Q3 operator behind the common header: energy partials, sum-of-squares partials, element vectors
void gen3d_layout(const dn_mesh* m, long long& n1, long long& n2, long long& elem_floats) {
    const int P = m->degree;
    const long long nel = (long long)((m->nx - 1) / P) * ((m->ny - 1) / P) * ((m->nz - 1) / P);
    const long long nbf = (long long)(P + 1) * (P + 1) * (P + 1);
    n1 = (nel * m->batch + 63) / 64;
    n2 = ((long long)m->nx * m->ny * m->nz * m->batch + 255) / 256;
    elem_floats = nel * nbf * m->batch;
}

int launch_poisson3d_gen(const PoissonParams& pp, const dn_mesh* m, void* workspace, int64_t workspace_bytes, hipStream_t s) {
    long long n1, n2, ef;
    gen3d_layout(m, n1, n2, ef);
    if (n1 >= (1ll << 31) || n2 >= (1ll << 31)) return DN_E_UNSUPPORTED;
    const int64_t need = DN_WS_HEADER + (int64_t)sizeof(double) * (n1 + n2) + (int64_t)sizeof(float) * ef;
    if (!workspace || workspace_bytes < need) return DN_E_WORKSPACE;
    Gen3DParams q;
    q.part_e = reinterpret_cast<double*>(reinterpret_cast<char*>(workspace) + DN_WS_HEADER);
    q.part_s = q.part_e + n1;
    q.elem = reinterpret_cast<float*>(q.part_s + n2);
    q.n1 = (int)n1; q.n2 = (int)n2;
    q.nel = (long long)pp.nelx * pp.nely * pp.nelz;
    q.nelem_total = q.nel * m->batch;
    const int P = m->degree;
    if (P == 2 && m->ngp == 3) return gen3d_launch<2, 3>(pp, q, m->batch, s);
    if (P == 2 && m->ngp == 4) return gen3d_launch<2, 4>(pp, q, m->batch, s);
    if (P == 3 && m->ngp == 3) return gen3d_launch<3, 3>(pp, q, m->batch, s);
    if (P == 3 && m->ngp == 4) return gen3d_launch<3, 4>(pp, q, m->batch, s);
    return DN_E_UNSUPPORTED;
}

}  // namespace dn
