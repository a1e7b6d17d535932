// gauss_pt_eval (DiffNet/DiffNetFEM.py:7-18) for an arbitrary user table list, its adjoint, and the
// element->node assembly helper (e8_2d_poisson_mms.py:85-90, e8_3d_poisson_mms.py:78-87) with its adjoint.
//
// The reference issues one single-channel strided conv per Gauss point and concatenates; here one
// launch produces all G channels: a thread owns one element, keeps its nbf^nsd nodal values in
// registers and streams the G x nbf^nsd table out of LDS (wave-uniform address => broadcast reads).
// HBM: read 4 B/node (+ overlap served by L1/L2), write 4*G B/element -- write dominated.
// The adjoint is in gather form: a thread owns one node and sums, in a fixed order, the
// contributions of the <= 2^nsd elements that contain it, so every output is written exactly once.
#include "dn_common.h"

namespace dn {

struct GpeGeom {
    int batch, nsd, nbf, stride, G;
    int n[3];      // nodes  x,y,z
    int nel[3];    // elements x,y,z
};

template <int NSD, int NB>
__global__ void __launch_bounds__(256) gpe_fwd_kernel(const float* __restrict__ in, const float* __restrict__ tables,
                                                      float* __restrict__ out, const GpeGeom g) {
    constexpr int NBT = NSD == 1 ? NB : (NSD == 2 ? NB * NB : NB * NB * NB);
    extern __shared__ float tab[];
    for (int i = threadIdx.x; i < g.G * NBT; i += blockDim.x) tab[i] = tables[i];
    __syncthreads();
    const int64_t nel_s = (int64_t)g.nel[0] * g.nel[1] * g.nel[2];
    const int64_t nps = (int64_t)g.n[0] * g.n[1] * g.n[2];
    const int64_t total = nel_s * g.batch;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(idx / nel_s);
        const int64_t e = idx - (int64_t)b * nel_s;
        const int ex = (int)(e % g.nel[0]);
        const int ey = (int)((e / g.nel[0]) % g.nel[1]);
        const int ez = (int)(e / ((int64_t)g.nel[0] * g.nel[1]));
        const float* src = in + (int64_t)b * nps + ((int64_t)ez * g.stride * g.n[1] + (int64_t)ey * g.stride) * g.n[0] + (int64_t)ex * g.stride;
        float v[NBT];
#pragma unroll
        for (int a = 0; a < NBT; ++a) {
            const int ib = a % NB, jb = (a / NB) % NB, kb = a / (NB * NB);
            v[a] = src[((int64_t)kb * g.n[1] + jb) * g.n[0] + ib];
        }
        float* dst = out + (int64_t)b * g.G * nel_s + e;
        for (int gi = 0; gi < g.G; ++gi) {
            float s = 0.f;
#pragma unroll
            for (int a = 0; a < NBT; ++a) s = fmaf(tab[gi * NBT + a], v[a], s);
            dst[(int64_t)gi * nel_s] = s;
        }
    }
}

template <int NSD, int NB>
__global__ void __launch_bounds__(256) gpe_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ tables,
                                                      float* __restrict__ gin, const GpeGeom g) {
    constexpr int NBT = NSD == 1 ? NB : (NSD == 2 ? NB * NB : NB * NB * NB);
    extern __shared__ float tab[];
    for (int i = threadIdx.x; i < g.G * NBT; i += blockDim.x) tab[i] = tables[i];
    __syncthreads();
    const int64_t nel_s = (int64_t)g.nel[0] * g.nel[1] * g.nel[2];
    const int64_t nps = (int64_t)g.n[0] * g.n[1] * g.n[2];
    const int64_t total = nps * g.batch;
    const int S = g.stride;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(idx / nps);
        const int64_t nd = idx - (int64_t)b * nps;
        const int x = (int)(nd % g.n[0]);
        const int y = (int)((nd / g.n[0]) % g.n[1]);
        const int z = (int)(nd / ((int64_t)g.n[0] * g.n[1]));
        // candidate (window, local index) pairs per axis: every window e with 0 <= x - e*S <= NB-1 (<= 4 of them)
        int ce[3][4], cl[3][4], cn[3];
        const int xyz[3] = {x, y, z};
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            cn[d] = 0;
            if (d >= NSD) { ce[d][0] = 0; cl[d][0] = 0; cn[d] = 1; continue; }
            // every element e with 0 <= xyz - e*S <= NB-1
            const int ehi = min(xyz[d] / S, g.nel[d] - 1);
            for (int e = ehi; e >= 0 && xyz[d] - e * S <= NB - 1 && cn[d] < 4; --e) { ce[d][cn[d]] = e; cl[d][cn[d]] = xyz[d] - e * S; ++cn[d]; }
        }
        float s = 0.f;
        const float* gsrc = gout + (int64_t)b * g.G * nel_s;
        for (int kz = cn[2] - 1; kz >= 0; --kz)
            for (int ky = cn[1] - 1; ky >= 0; --ky)
                for (int kx = cn[0] - 1; kx >= 0; --kx) {
                    const int64_t e = ((int64_t)ce[2][kz] * g.nel[1] + ce[1][ky]) * g.nel[0] + ce[0][kx];
                    const int aa = (cl[2][kz] * NB + cl[1][ky]) * NB + cl[0][kx];   // cl is 0 on unused axes
                    for (int gi = 0; gi < g.G; ++gi) s = fmaf(tab[gi * NBT + aa], gsrc[(int64_t)gi * nel_s + e], s);
                }
        gin[idx] = s;
    }
}

// out[b,node] (+)= sum_{(e,a) -> node} r_split[b,a,e]; contributions are added in ascending local id a,
// i.e. the order of the reference's sliced "+=" lines.
template <int NSD, int NB>
__global__ void __launch_bounds__(256) assemble_kernel(const float* __restrict__ rs, float* __restrict__ out, const GpeGeom g,
                                                       const int accumulate) {
    constexpr int NBT = NSD == 2 ? NB * NB : NB * NB * NB;
    const int64_t nel_s = (int64_t)g.nel[0] * g.nel[1] * g.nel[2];
    const int64_t nps = (int64_t)g.n[0] * g.n[1] * g.n[2];
    const int64_t total = nps * g.batch;
    const int S = g.stride;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(idx / nps);
        const int64_t nd = idx - (int64_t)b * nps;
        const int xyz[3] = {(int)(nd % g.n[0]), (int)((nd / g.n[0]) % g.n[1]), (int)(nd / ((int64_t)g.n[0] * g.n[1]))};
        int ce[3][4], cl[3][4], cn[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            cn[d] = 0;
            if (d >= NSD) { ce[d][0] = 0; cl[d][0] = 0; cn[d] = 1; continue; }
            const int ehi = min(xyz[d] / S, g.nel[d] - 1);
            for (int e = ehi; e >= 0 && xyz[d] - e * S <= NB - 1 && cn[d] < 4; --e) { ce[d][cn[d]] = e; cl[d][cn[d]] = xyz[d] - e * S; ++cn[d]; }
        }
        // candidates are stored with the local index ascending (upper element first => local id 0 first)
        float s = accumulate ? out[idx] : 0.f;
        const float* src = rs + (int64_t)b * NBT * nel_s;
        for (int kz = 0; kz < cn[2]; ++kz)
            for (int ky = 0; ky < cn[1]; ++ky)
                for (int kx = 0; kx < cn[0]; ++kx) {
                    const int64_t e = ((int64_t)ce[2][kz] * g.nel[1] + ce[1][ky]) * g.nel[0] + ce[0][kx];
                    const int a = NSD == 2 ? cl[1][ky] * NB + cl[0][kx] : (cl[2][kz] * NB + cl[1][ky]) * NB + cl[0][kx];
                    s += src[(int64_t)a * nel_s + e];
                }
        out[idx] = s;
    }
}

// adjoint of assemble: grad_split[b,a,e] = grad_out[b,node(e,a)]
template <int NSD, int NB>
__global__ void __launch_bounds__(256) assemble_bwd_kernel(const float* __restrict__ gout, float* __restrict__ gs, const GpeGeom g) {
    constexpr int NBT = NSD == 2 ? NB * NB : NB * NB * NB;
    const int64_t nel_s = (int64_t)g.nel[0] * g.nel[1] * g.nel[2];
    const int64_t nps = (int64_t)g.n[0] * g.n[1] * g.n[2];
    const int64_t total = nel_s * g.batch;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(idx / nel_s);
        const int64_t e = idx - (int64_t)b * nel_s;
        const int ex = (int)(e % g.nel[0]);
        const int ey = (int)((e / g.nel[0]) % g.nel[1]);
        const int ez = (int)(e / ((int64_t)g.nel[0] * g.nel[1]));
        const float* src = gout + (int64_t)b * nps + ((int64_t)ez * g.stride * g.n[1] + (int64_t)ey * g.stride) * g.n[0] + (int64_t)ex * g.stride;
        float* dst = gs + (int64_t)b * NBT * nel_s + e;
#pragma unroll
        for (int a = 0; a < NBT; ++a) {
            const int ib = a % NB, jb = (a / NB) % NB, kb = a / (NB * NB);
            dst[(int64_t)a * nel_s] = src[((int64_t)kb * g.n[1] + jb) * g.n[0] + ib];
        }
    }
}

static int make_geom(GpeGeom& g, int batch, int nsd, const int32_t n[3], int nbf, int stride, int G) {
    if (!n || batch < 1 || nsd < 1 || nsd > 3 || nbf < 2 || nbf > 4 || stride < 1 || G < 1) return DN_E_BADARG;
    g.batch = batch; g.nsd = nsd; g.nbf = nbf; g.stride = stride; g.G = G;
    for (int d = 0; d < 3; ++d) {
        g.n[d] = d < nsd ? n[d] : 1;
        if (g.n[d] < 1) return DN_E_BADARG;
        g.nel[d] = d < nsd ? (g.n[d] - nbf) / stride + 1 : 1;   // conv output length
        if (d < nsd && g.n[d] < nbf) return DN_E_BADARG;
    }
    return 0;
}

static int grid_for(int64_t total) {
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;   // grid-stride beyond 16 workgroups per CU
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

#define DN_DISPATCH_NSD_NB(KERNEL, nsd, nbf, ...)                                                   \
    switch ((nsd) * 10 + (nbf)) {                                                                   \
        case 12: KERNEL(1, 2, __VA_ARGS__); break;                                                  \
        case 13: KERNEL(1, 3, __VA_ARGS__); break;                                                  \
        case 14: KERNEL(1, 4, __VA_ARGS__); break;                                                  \
        case 22: KERNEL(2, 2, __VA_ARGS__); break;                                                  \
        case 23: KERNEL(2, 3, __VA_ARGS__); break;                                                  \
        case 24: KERNEL(2, 4, __VA_ARGS__); break;                                                  \
        case 32: KERNEL(3, 2, __VA_ARGS__); break;                                                  \
        case 33: KERNEL(3, 3, __VA_ARGS__); break;                                                  \
        case 34: KERNEL(3, 4, __VA_ARGS__); break;                                                  \
        default: return DN_E_UNSUPPORTED;                                                           \
    }

}  // namespace dn

using namespace dn;

extern "C" int dn_gauss_pt_eval_fwd(const float* in, const float* tables, float* out, int32_t batch, int32_t nsd,
                                    const int32_t n[3], int32_t nbf, int32_t stride, int32_t G, void* stream) {
    GpeGeom g;
    int rc = make_geom(g, batch, nsd, n, nbf, stride, G);
    if (rc) return rc;
    if (!in || !tables || !out) return DN_E_BADARG;
    int nbt = 1;
    for (int d = 0; d < nsd; ++d) nbt *= nbf;
    const size_t lds = sizeof(float) * (size_t)G * nbt;
    if (lds > 64 * 1024) return DN_E_UNSUPPORTED;
    const int64_t total = (int64_t)g.nel[0] * g.nel[1] * g.nel[2] * batch;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define K_FWD(NSD, NB, ...) hipLaunchKernelGGL((gpe_fwd_kernel<NSD, NB>), dim3(grid_for(total)), dim3(256), lds, s, in, tables, out, g)
    DN_DISPATCH_NSD_NB(K_FWD, nsd, nbf, 0)
#undef K_FWD
    DN_LAUNCH_CHECK();
    return 0;
}

extern "C" int dn_gauss_pt_eval_bwd(const float* grad_out, const float* tables, float* grad_in, int32_t batch, int32_t nsd,
                                    const int32_t n[3], int32_t nbf, int32_t stride, int32_t G, void* stream) {
    GpeGeom g;
    int rc = make_geom(g, batch, nsd, n, nbf, stride, G);
    if (rc) return rc;
    if (!grad_out || !tables || !grad_in) return DN_E_BADARG;
    int nbt = 1;
    for (int d = 0; d < nsd; ++d) nbt *= nbf;
    const size_t lds = sizeof(float) * (size_t)G * nbt;
    if (lds > 64 * 1024) return DN_E_UNSUPPORTED;
    const int64_t total = (int64_t)g.n[0] * g.n[1] * g.n[2] * batch;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define K_BWD(NSD, NB, ...) hipLaunchKernelGGL((gpe_bwd_kernel<NSD, NB>), dim3(grid_for(total)), dim3(256), lds, s, grad_out, tables, grad_in, g)
    DN_DISPATCH_NSD_NB(K_BWD, nsd, nbf, 0)
#undef K_BWD
    DN_LAUNCH_CHECK();
    return 0;
}

extern "C" int dn_assemble(const float* r_split, float* out, int32_t batch, int32_t nsd, const int32_t n[3], int32_t nbf,
                           int32_t stride, int32_t accumulate, void* stream) {
    GpeGeom g;
    int rc = make_geom(g, batch, nsd, n, nbf, stride, 1);
    if (rc) return rc;
    if (!r_split || !out || nsd < 2 || stride != nbf - 1) return DN_E_BADARG;
    const int64_t total = (int64_t)g.n[0] * g.n[1] * g.n[2] * batch;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define K_ASM(NSD, NB, ...) hipLaunchKernelGGL((assemble_kernel<NSD, NB>), dim3(grid_for(total)), dim3(256), 0, s, r_split, out, g, accumulate)
    switch (nsd * 10 + nbf) {
        case 22: K_ASM(2, 2); break;
        case 23: K_ASM(2, 3); break;
        case 24: K_ASM(2, 4); break;
        case 32: K_ASM(3, 2); break;
        case 33: K_ASM(3, 3); break;
        case 34: K_ASM(3, 4); break;
        default: return DN_E_UNSUPPORTED;
    }
#undef K_ASM
    DN_LAUNCH_CHECK();
    return 0;
}

extern "C" int dn_assemble_bwd(const float* grad_out, float* grad_split, int32_t batch, int32_t nsd, const int32_t n[3],
                               int32_t nbf, int32_t stride, void* stream) {
    GpeGeom g;
    int rc = make_geom(g, batch, nsd, n, nbf, stride, 1);
    if (rc) return rc;
    if (!grad_out || !grad_split || nsd < 2 || stride != nbf - 1) return DN_E_BADARG;
    const int64_t total = (int64_t)g.nel[0] * g.nel[1] * g.nel[2] * batch;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define K_ASMB(NSD, NB, ...) hipLaunchKernelGGL((assemble_bwd_kernel<NSD, NB>), dim3(grid_for(total)), dim3(256), 0, s, grad_out, grad_split, g)
    switch (nsd * 10 + nbf) {
        case 22: K_ASMB(2, 2); break;
        case 23: K_ASMB(2, 3); break;
        case 24: K_ASMB(2, 4); break;
        case 32: K_ASMB(3, 2); break;
        case 33: K_ASMB(3, 3); break;
        case 34: K_ASMB(3, 4); break;
        default: return DN_E_UNSUPPORTED;
    }
#undef K_ASMB
    DN_LAUNCH_CHECK();
    return 0;
}
