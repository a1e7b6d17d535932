// 4 x 4, stride-2, padding-1 convolution family of the 2-D networks on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact
// fp32 products, fp32 accumulation -- the reference's precision), NCHW in and out, no layout transposes, no im2col buffer.
//
// Reference layers: DiffNet/networks/unets.py:13-45 (`UNetDown`: Conv2d(cin, cout, 4, 2, 1, bias=False); `UNetUp`:
// ConvTranspose2d(cin, cout, 4, 2, 1, bias=False)) and DiffNet/networks/autoencoders.py:24-45 (the same two layers in the AE).
//
// With `fine` = the tensor on the 2H x 2W grid, `coarse` = the tensor on the H x W grid and ONE weight layout
// w[m][c][ky][kx] (m = coarse channel, c = fine channel) -- which is Conv2d's (cout, cin, 4, 4) as it stands AND
// ConvTranspose2d's (cin, cout, 4, 4) as it stands -- every pass of both layers is one of three contractions:
//
//   down :  coarse[b,m,i,j]   = sum_{c,ky,kx} w[m,c,ky,kx] * fine[b,c,2i+ky-1,2j+kx-1]          Conv2d forward, ConvT input gradient
//   up   :  fine[b,c,y,x]     = sum_{m,ky,kx} w[m,c,ky,kx] * coarse[b,m,(y+1-ky)/2,(x+1-kx)/2]  ConvT forward, Conv2d input gradient
//                                                                   (terms with odd y+1-ky / x+1-kx or outside the grid vanish)
//   wrw  :  gw[m,c,ky,kx]     = sum_{b,i,j}   coarse[b,m,i,j] * fine[b,c,2i+ky-1,2j+kx-1]        weight gradient of both
//
// All three are implicit GEMMs tiled 64 x 64 per 256-thread workgroup (2 x 2 waves, 32 x 32 per wave = 2 x 2 MFMA tiles),
// operands staged through LDS with conflict-free strides, the next K-step's global loads in flight during the MFMAs
// (registers -> the other LDS buffer after the math, one barrier per step).
//   down: K = (c, ky, kx); A = w rows (contiguous in memory), B = the 4 x 4 input patches of 64 coarse positions.
//   up  : one GEMM per output parity (y & 1, x & 1) with K = (m, 2 x 2 taps); the four parities share the staged coarse
//         neighbourhood; a lane holds both x parities of a position and stores them as one float2.
//   wrw : K = positions; split over workgroups, partial sums reduced in index order by a second launch (deterministic).
#include "dn_common.h"

namespace dn {

// round 4: raw-row-tile forms (conv2d_k4s2_v2.hip); false = preconditions not met (or "CONV2D_V1" set): run the kernels above
bool conv2d_down_v2(const float* fine, const float* w, float* coarse, int64_t B, int64_t C, int64_t M, int64_t H, int64_t W, hipStream_t s);
bool conv2d_up_v2(const float* coarse, const float* w, float* fine, int64_t B, int64_t C, int64_t M, int64_t H, int64_t W, hipStream_t s);
bool conv2d_wrw_v2_ok(const float* fine, const float* coarse, int64_t C, int64_t H, int64_t W);
bool conv2d_wrw_v2_shape_ok(int64_t C, int64_t H, int64_t W);
int conv2d_wrw_v2_ncw(int64_t C);
bool conv2d_wrw_v2(const float* fine, const float* coarse, float* part, int64_t B, int64_t C, int64_t M, int64_t H, int64_t W, int nz, int tiles_per_wg,
                   hipStream_t s);


typedef float c2_f32x4 __attribute__((ext_vector_type(4)));

constexpr int C2_TN = 64;          // positions per tile
constexpr int C2_SA = 66;          // LDS row stride of an A tile [row][k]:   (66 r + k) mod 32 distinct for 16 rows x 2 k
constexpr int C2_SB = 80;          // LDS row stride of a B tile [k][col]:    (80 k + c) mod 32 distinct for 2 k x 16 cols

__device__ __forceinline__ c2_f32x4 mfma4(float a, float b, c2_f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// v where ok, +0 elsewhere, as a bit mask: the loaded value is always consumed, so the compiler cannot sink the (clamped, always
// legal) load into a divergent branch -- which it does for `ok ? v : 0`, adding a branch and a vmcnt(0) wait per load
__device__ __forceinline__ unsigned opaque_mask(bool ok) {
    unsigned m = ok ? 0xffffffffu : 0u;
    asm volatile("" : "+v"(m));           // the compiler must not see that m is 0 / ~0 (it would rebuild the select and sink the load)
    return m;
}
__device__ __forceinline__ float keep_if(float v, bool ok) { return __uint_as_float(__float_as_uint(v) & opaque_mask(ok)); }
__device__ __forceinline__ float keep_mask(float v, unsigned m) { return __uint_as_float(__float_as_uint(v) & m); }

// =====================================================================================================================
// down:  coarse[b, m, p] = sum_k w[m][k] * patch[k][p],  k = c * 16 + ky * 4 + kx,  p = i * W + j
// grid = (B * ceil(HW / 64), ceil(M / TM)), block = 256.  TM = 64 or 32 coarse channels per workgroup.
// =====================================================================================================================
template <int TM>
__global__ void __launch_bounds__(256) conv2d_k4s2_down_kernel(const float* __restrict__ fine, const float* __restrict__ w,
                                                               float* __restrict__ coarse, int B, int C, int M, int H, int W) {
    constexpr int KC = 4;                               // fine channels per K-step (K-step = 64)
    constexpr int RT = TM / 32;                         // 16-row MFMA tiles per wave along m
    __shared__ __attribute__((aligned(16))) float As[2][TM][C2_SA];
    __shared__ __attribute__((aligned(16))) float Bs[2][KC * 16][C2_SB];
    const int tid = threadIdx.x;
    const int HW = H * W, tiles_per_sample = (HW + C2_TN - 1) / C2_TN;
    const int b = blockIdx.x / tiles_per_sample, p0 = (blockIdx.x % tiles_per_sample) * C2_TN;
    const int m0 = blockIdx.y * TM;
    const int W2 = 2 * W, H2 = 2 * H;
    // staging roles: B patches -- thread (n, q) loads the patch of position p0 + n, channel c0 + q
    const int n = tid & 63, q = tid >> 6;
    const int p = p0 + n;
    const bool pok = p < HW;
    const int pi = pok ? p / W : 0, pj = pok ? p % W : 0;
    const float* fb = fine + (size_t)b * C * H2 * W2;
    // MFMA roles
    const int lane = tid & 63, wv = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int wm = wv >> 1, wn = wv & 1;
    const int row0 = wm * (TM / 2), col0 = wn * 32;
    c2_f32x4 acc[RT][2];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int s = 0; s < 2; ++s) acc[r][s] = (c2_f32x4){0.f, 0.f, 0.f, 0.f};

    float breg[16];
    float4 areg[TM / 16];
    // patch loads are unconditional on clamped (in-bounds) addresses; out-of-range taps are zeroed by selects afterwards: no
    // exec-masked branch per load, SGPR base + 32-bit offset addressing
    unsigned xoff[4], yoff[4], tapmask[16];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int x = 2 * pj + t - 1, y = 2 * pi + t - 1;
        xoff[t] = (unsigned)min(max(x, 0), W2 - 1);
        yoff[t] = (unsigned)min(max(y, 0), H2 - 1) * (unsigned)W2;
    }
#pragma unroll
    for (int ky = 0; ky < 4; ++ky)
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) {
            const int x = 2 * pj + kx - 1, y = 2 * pi + ky - 1;
            tapmask[ky * 4 + kx] = opaque_mask(pok && x >= 0 && x < W2 && y >= 0 && y < H2);
        }
    auto issue = [&](int c0) {
        const int c = c0 + q;
        const unsigned cmask = opaque_mask(c < C);
        const unsigned coff = (unsigned)min(c, C - 1) * (unsigned)(H2 * W2);
#pragma unroll
        for (int ky = 0; ky < 4; ++ky)
#pragma unroll
            for (int kx = 0; kx < 4; ++kx)
                breg[ky * 4 + kx] = keep_mask(ld_at<float>(fb, coff + yoff[ky] + xoff[kx]), tapmask[ky * 4 + kx] & cmask);
        // A tile: w[m0 + mm][c0 .. c0 + 3][16] = 64 contiguous floats per row (zero beyond M / C)
#pragma unroll
        for (int r = 0; r < TM / 16; ++r) {
            const int idx4 = tid + 256 * r, mm = idx4 >> 4, kk = (idx4 & 15) * 4;
            const int cc = c0 + (kk >> 4);
            areg[r] = (m0 + mm < M && cc < C) ? *reinterpret_cast<const float4*>(w + ((size_t)(m0 + mm) * C + cc) * 16 + (kk & 15))
                                              : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int t = 0; t < 16; ++t) Bs[buf][q * 16 + t][n] = breg[t];
#pragma unroll
        for (int r = 0; r < TM / 16; ++r) {
            const int idx4 = tid + 256 * r, mm = idx4 >> 4, kk = (idx4 & 15) * 4;
            *reinterpret_cast<float2*>(&As[buf][mm][kk]) = make_float2(areg[r].x, areg[r].y);
            *reinterpret_cast<float2*>(&As[buf][mm][kk + 2]) = make_float2(areg[r].z, areg[r].w);
        }
    };

    const int nsteps = (C + KC - 1) / KC;
    issue(0);
    commit(0);
    __syncthreads();
    for (int st = 0; st < nsteps; ++st) {
        const int buf = st & 1;
        const bool more = st + 1 < nsteps;
        if (more) issue((st + 1) * KC);
        // operand fragments are read from LDS one group of 4 k-steps ahead of the MFMAs that consume them (register double buffer;
        // 16 MFMAs = 512 cycles cover the LDS latency).  sched_barrier keeps the compiler from re-serialising read -> wait -> MFMA.
        constexpr int GK = 4, NG = KC * 4 / GK;
        float a[2][GK][RT], bb[2][GK][2];
        auto frags = [&](int g, float (&fa)[GK][RT], float (&fbv)[GK][2]) {
#pragma unroll
            for (int kk = 0; kk < GK; ++kk) {
#pragma unroll
                for (int r = 0; r < RT; ++r) fa[kk][r] = As[buf][row0 + 16 * r + li][4 * (g * GK + kk) + lk];
#pragma unroll
                for (int s = 0; s < 2; ++s) fbv[kk][s] = Bs[buf][4 * (g * GK + kk) + lk][col0 + 16 * s + li];
            }
        };
        frags(0, a[0], bb[0]);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int cur = g & 1, nxt = cur ^ 1;
            if (g + 1 < NG) frags(g + 1, a[nxt], bb[nxt]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < GK; ++kk)
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int s = 0; s < 2; ++s) acc[r][s] = mfma4(a[cur][kk][r], bb[cur][kk][s], acc[r][s]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) commit(buf ^ 1);
        __syncthreads();
    }
    // D layout: column = lane & 15, row = 4 (lane >> 4) + register
    float* ob = coarse + (size_t)b * M * HW;
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int pp = p0 + col0 + 16 * s + li;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int m = m0 + row0 + 16 * r + 4 * lk + qq;
                if (m < M && pp < HW) ob[(size_t)m * HW + pp] = acc[r][s][qq];
            }
        }
}

// =====================================================================================================================
// up:  fine[b, c, 2I + py, 2J + px] = sum_m sum_{a,bb in {0,1}} w[m][c][ky(py,a)][kx(px,bb)] * coarse[b, m, I + di(py,a), J + dj(px,bb)]
//      py = 0: (ky, di) = (1, 0), (3, -1);   py = 1: (ky, di) = (0, +1), (2, 0);   likewise in x.
// One MFMA k-step = one coarse channel x its 2 x 2 taps.  grid = (B * H * ceil(W / 64)... positions are tiled as 64 consecutive
// flattened coarse positions P = I * W + J of one sample), block = 256, TC = 64 fine channels per workgroup.
// =====================================================================================================================
template <int TC>                                     // 64 or 32 fine channels per workgroup tile
__global__ void __launch_bounds__(256) conv2d_k4s2_up_kernel(const float* __restrict__ coarse, const float* __restrict__ w,
                                                             float* __restrict__ fine, int B, int C, int M, int H, int W) {
    constexpr int KC = 4;                               // coarse channels per K-step
    constexpr int RT = TC / 32;                         // 16-row MFMA tiles per wave along c
    // weights of the step: Ws[mm][tap][c]  (c fastest: conflict-free A reads), tap = ky * 4 + kx
    __shared__ __attribute__((aligned(16))) float Ws[2][KC][16][TC + 8];     // stride 72: taps two apart land 16 banks apart
    // coarse neighbourhood: Ps[mm][d = (di + 1) * 3 + (dj + 1)][n]  (the 3 x 3 neighbours of the 64 positions)
    __shared__ __attribute__((aligned(16))) float Ps[2][KC][9][C2_SB];
    const int tid = threadIdx.x;
    const int HW = H * W, tiles_per_sample = (HW + C2_TN - 1) / C2_TN;
    const int b = blockIdx.x / tiles_per_sample, p0 = (blockIdx.x % tiles_per_sample) * C2_TN;
    const int c0 = blockIdx.y * TC;
    const int n = tid & 63, q = tid >> 6;
    const int P = p0 + n;
    const bool pok = P < HW;
    const int I = pok ? P / W : 0, J = pok ? P % W : 0;
    const float* cb = coarse + (size_t)b * M * HW;
    const int lane = tid & 63, wv = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int wm = wv >> 1, wn = wv & 1;
    const int row0 = wm * (TC / 2), col0 = wn * 32;
    const int ta = lk >> 1, tb = lk & 1;                // this lane group's tap (a, bb) inside the 2 x 2 tap block
    c2_f32x4 acc[4][RT][2];                             // [parity py * 2 + px][row tile][col tile]
#pragma unroll
    for (int par = 0; par < 4; ++par)
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int s = 0; s < 2; ++s) acc[par][r][s] = (c2_f32x4){0.f, 0.f, 0.f, 0.f};

    constexpr int NWL = KC * TC * 16 / 4 / 256;         // float4 weight loads per thread and step
    constexpr int CSH = TC == 64 ? 6 : 5;               // log2(TC)
    float preg[9];
    float4 wreg[NWL];
    auto issue = [&](int mstep) {
        const int m = mstep + q;
        const bool mok = pok && m < M;
        const float* cm = cb + (size_t)(m < M ? m : 0) * HW;
#pragma unroll
        for (int di = -1; di <= 1; ++di) {
            const int ii = I + di;
            const bool iok = mok && ii >= 0 && ii < H;
            const int io = min(max(ii, 0), H - 1) * W;
#pragma unroll
            for (int dj = -1; dj <= 1; ++dj) {
                const int jj = J + dj;
                const float v = cm[io + min(max(jj, 0), W - 1)];
                preg[(di + 1) * 3 + dj + 1] = keep_if(v, iok && jj >= 0 && jj < W);
            }
        }
        // weights: w[mstep + mm][c0 + cc][16]: KC * TC * 16 floats as float4
#pragma unroll
        for (int r = 0; r < NWL; ++r) {
            const int idx4 = tid + 256 * r, mm = idx4 >> (CSH + 2), cc = (idx4 >> 2) & (TC - 1), t4 = (idx4 & 3) * 4;
            wreg[r] = (mstep + mm < M && c0 + cc < C) ? *reinterpret_cast<const float4*>(w + ((size_t)(mstep + mm) * C + c0 + cc) * 16 + t4)
                                                     : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int d = 0; d < 9; ++d) Ps[buf][q][d][n] = preg[d];
#pragma unroll
        for (int r = 0; r < NWL; ++r) {
            const int idx4 = tid + 256 * r, mm = idx4 >> (CSH + 2), cc = (idx4 >> 2) & (TC - 1), t4 = (idx4 & 3) * 4;
            Ws[buf][mm][t4 + 0][cc] = wreg[r].x;
            Ws[buf][mm][t4 + 1][cc] = wreg[r].y;
            Ws[buf][mm][t4 + 2][cc] = wreg[r].z;
            Ws[buf][mm][t4 + 3][cc] = wreg[r].w;
        }
    };

    const int nsteps = (M + KC - 1) / KC;
    issue(0);
    commit(0);
    __syncthreads();
    for (int st = 0; st < nsteps; ++st) {
        const int buf = st & 1;
        const bool more = st + 1 < nsteps;
        if (more) issue((st + 1) * KC);
        // KC * 4 sub-steps (coarse channel mm, parity py, px), each one k-step of 4 taps; fragments of sub-step t + 1 are read
        // from LDS while the MFMAs of sub-step t run
        auto frag = [&](int t, float (&a)[RT], float (&bv)[2]) {
            const int mm = t >> 2, py = (t >> 1) & 1, px = t & 1;
            const int ky = py == 0 ? (ta == 0 ? 1 : 3) : (ta == 0 ? 0 : 2);
            const int di = py == 0 ? (ta == 0 ? 0 : -1) : (ta == 0 ? 1 : 0);
            const int kx = px == 0 ? (tb == 0 ? 1 : 3) : (tb == 0 ? 0 : 2);
            const int dj = px == 0 ? (tb == 0 ? 0 : -1) : (tb == 0 ? 1 : 0);
#pragma unroll
            for (int r = 0; r < RT; ++r) a[r] = Ws[buf][mm][ky * 4 + kx][row0 + 16 * r + li];
#pragma unroll
            for (int s = 0; s < 2; ++s) bv[s] = Ps[buf][mm][(di + 1) * 3 + dj + 1][col0 + 16 * s + li];
        };
        float fa[2][4][RT], fb2[2][4][2];
#pragma unroll
        for (int t = 0; t < 4; ++t) frag(t, fa[0][t], fb2[0][t]);
#pragma unroll
        for (int mm = 0; mm < KC; ++mm) {
            const int cur = mm & 1, nxt = cur ^ 1;
            if (mm + 1 < KC) {
#pragma unroll
                for (int t = 0; t < 4; ++t) frag((mm + 1) * 4 + t, fa[nxt][t], fb2[nxt][t]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int par = 0; par < 4; ++par)
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int s = 0; s < 2; ++s) acc[par][r][s] = mfma4(fa[cur][par][r], fb2[cur][par][s], acc[par][r][s]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) commit(buf ^ 1);
        __syncthreads();
    }
    // epilogue: the lane's position (column) is the same in all four parity tiles: two float2 stores per channel (rows 2I, 2I + 1)
    const int W2 = 2 * W;
    float* fo = fine + (size_t)b * C * (4 * (size_t)HW);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int PP = p0 + col0 + 16 * s + li;
        if (PP >= HW) continue;
        const int II = PP / W, JJ = PP % W;
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int c = c0 + row0 + 16 * r + 4 * lk + qq;
                if (c >= C) continue;
                float* dst = fo + ((size_t)c * (2 * H) + 2 * II) * W2 + 2 * JJ;
                *reinterpret_cast<float2*>(dst) = make_float2(acc[0][r][s][qq], acc[1][r][s][qq]);
                *reinterpret_cast<float2*>(dst + W2) = make_float2(acc[2][r][s][qq], acc[3][r][s][qq]);
            }
    }
}

// =====================================================================================================================
// wrw:  gw[m][c][tap] = sum_{b, p} coarse[b][m][p] * patch[b][c][tap][p]
// GEMM rows = m (64 per workgroup), columns = (c, tap) (4 fine channels x 16 taps), K = positions in tiles of 64, split over
// gridDim.z workgroups; partials [z][m][c][tap] summed in index order by conv2d_k4s2_wsum_kernel.
// =====================================================================================================================
__global__ void __launch_bounds__(256) conv2d_k4s2_wrw_kernel(const float* __restrict__ fine, const float* __restrict__ coarse,
                                                              float* __restrict__ part, int B, int C, int M, int H, int W,
                                                              int tiles_per_wg) {
    // A[m][pos] read as A[row li][k lk]: Vs[m][pos] stride 66; B[pos][col] read as B[k lk][col li]: Ss[pos][col] stride 80
    __shared__ __attribute__((aligned(16))) float Vs[2][64][C2_SA];
    __shared__ __attribute__((aligned(16))) float Ss[2][C2_TN][C2_SB];
    const int tid = threadIdx.x;
    const int HW = H * W, H2 = 2 * H, W2 = 2 * W;
    const long npos = (long)B * HW;
    const int m0 = blockIdx.y * 64, c0 = blockIdx.x * 4;
    const int n = tid & 63, q = tid >> 6;
    const int lane = tid & 63, wv = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int wm = wv >> 1, wn = wv & 1;
    const int row0 = wm * 32, col0 = wn * 32;
    c2_f32x4 acc[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int s = 0; s < 2; ++s) acc[r][s] = (c2_f32x4){0.f, 0.f, 0.f, 0.f};
    float sreg[16], vreg[16];
    auto issue = [&](int t) {
        const long pg = ((long)blockIdx.z * tiles_per_wg + t) * C2_TN + n;
        const bool ok = pg < npos;
        const int b = ok ? (int)(pg / HW) : 0, p = ok ? (int)(pg % HW) : 0;
        const int i = p / W, j = p % W;
        const int c = c0 + q;
        const bool cok = ok && c < C;
        const float* fc = fine + ((size_t)b * C + (c < C ? c : 0)) * H2 * W2;
#pragma unroll
        for (int ky = 0; ky < 4; ++ky) {
            const int y = 2 * i + ky - 1;
            const bool yok = cok && y >= 0 && y < H2;
            const unsigned yo = (unsigned)min(max(y, 0), H2 - 1) * (unsigned)W2;
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) {
                const int x = 2 * j + kx - 1;
                const float v = fc[yo + (unsigned)min(max(x, 0), W2 - 1)];
                sreg[ky * 4 + kx] = keep_if(v, yok && x >= 0 && x < W2);
            }
        }
        const float* cb = coarse + (size_t)b * M * HW + p;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + q + 4 * r;
            const float v = cb[(size_t)min(m, M - 1) * HW];
            vreg[r] = keep_if(v, ok && m < M);
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4)
            *reinterpret_cast<float4*>(&Ss[buf][n][q * 16 + t4 * 4]) = make_float4(sreg[t4 * 4], sreg[t4 * 4 + 1], sreg[t4 * 4 + 2], sreg[t4 * 4 + 3]);
#pragma unroll
        for (int r = 0; r < 16; ++r) Vs[buf][q + 4 * r][n] = vreg[r];
    };
    issue(0);
    commit(0);
    __syncthreads();
    for (int t = 0; t < tiles_per_wg; ++t) {
        const int buf = t & 1;
        const bool more = t + 1 < tiles_per_wg;
        if (more) issue(t + 1);
        constexpr int GK = 4, NG = C2_TN / 4 / GK;
        float a[2][GK][2], bb[2][GK][2];
        auto frags = [&](int g, float (&fa)[GK][2], float (&fbv)[GK][2]) {
#pragma unroll
            for (int kk = 0; kk < GK; ++kk) {
#pragma unroll
                for (int r = 0; r < 2; ++r) fa[kk][r] = Vs[buf][row0 + 16 * r + li][4 * (g * GK + kk) + lk];
#pragma unroll
                for (int s = 0; s < 2; ++s) fbv[kk][s] = Ss[buf][4 * (g * GK + kk) + lk][col0 + 16 * s + li];
            }
        };
        frags(0, a[0], bb[0]);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int cur = g & 1, nxt = cur ^ 1;
            if (g + 1 < NG) frags(g + 1, a[nxt], bb[nxt]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < GK; ++kk)
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int s = 0; s < 2; ++s) acc[r][s] = mfma4(a[cur][kk][r], bb[cur][kk][s], acc[r][s]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) commit(buf ^ 1);
        __syncthreads();
    }
    const size_t nout = (size_t)M * C * 16;
    float* po = part + (size_t)blockIdx.z * nout;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int col = col0 + 16 * s + li, c = c0 + (col >> 4), tap = col & 15;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int m = m0 + row0 + 16 * r + 4 * lk + qq;
                if (m < M && c < C) po[((size_t)m * C + c) * 16 + tap] = acc[r][s][qq];
            }
        }
}

__global__ void __launch_bounds__(256) conv2d_k4s2_wsum_kernel(const float* __restrict__ part, float* __restrict__ gw, int nz, long n) {
    const long k = (long)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    double s = 0.0;
    for (int z = 0; z < nz; ++z) s += (double)part[(size_t)z * n + k];
    gw[k] = (float)s;
}

// the same sum with one WAVE per output (lane l adds slices l, l + 64, ... in order, then the fixed-order wave sum): for launches with many
// K-slices and few outputs -- the U-Net's first layer has 1024 outputs: one thread per output left the whole reduction to four workgroups
__global__ void __launch_bounds__(256) conv2d_k4s2_wsum_wave_kernel(const float* __restrict__ part, float* __restrict__ gw, int nz, long n) {
    const long k = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= n) return;
    const int lane = threadIdx.x & 63;
    double s = 0.0;
    for (int z = lane; z < nz; z += 64) s += (double)part[(size_t)z * n + k];
    s = wave_sum(s);
    if (lane == 0) gw[k] = (float)s;
}

static void c2_wrw_plan(int64_t B, int64_t C, int64_t M, int64_t H, int64_t W, int& nz, int& tiles_per_wg) {
    const int64_t tiles = (B * H * W + C2_TN - 1) / C2_TN;
    const int64_t ncw = conv2d_wrw_v2_shape_ok(C, H, W) ? conv2d_wrw_v2_ncw(C) : 4;       // fine channels per workgroup of the kernel that will run
    const int64_t groups = ((C + ncw - 1) / ncw) * ((M + 63) / 64);
    const char* tw = config(CFG_CONV_WRW_WGS);             // (tuning switch: workgroups a weight-gradient launch aims at)
    const int64_t total = tw ? std::max(256, std::atoi(tw)) : 1024;      // round 4: 2048 -> 1024 with the v2 kernels (3-4 resident per CU): fewer, longer K-slices and
                                                                         // half the partial arrays to sum -- U-Net step 7.28 -> 7.12 ms (2048 / 1280 / 1024 / 768: 7.28 / 7.26 / 7.12 / 7.19)
    int64_t want = (total + groups - 1) / groups;
    if (want < 1) want = 1;
    if (want > tiles) want = tiles;
    if (want > 1024) want = 1024;                         // (round 4: 256 -> 1024 for layers with ONE (channel group, row group) pair -- the U-Net's first, 2 -> 32,
                                                          // was a launch of 256 workgroups of 64 dependent chunks each; their partials are summed a wave per output)
    tiles_per_wg = (int)((tiles + want - 1) / want);
    nz = (int)((tiles + tiles_per_wg - 1) / tiles_per_wg);
}

static int c2_check(int64_t B, int64_t C, int64_t M, int64_t H, int64_t W) {
    if (B < 1 || C < 1 || M < 1 || H < 1 || W < 1) return DN_E_BADARG;
    if (C * 4 * H * W >= (1ll << 30) || M * H * W >= (1ll << 30) || M > 65535 * 32 || C > 65535 * 4) return DN_E_UNSUPPORTED;   // 32-bit in-sample offsets
    return 0;
}

}  // namespace dn

using namespace dn;

extern "C" int dn_conv2d_k4s2_down(const float* fine, const float* w, float* coarse, int64_t B, int64_t C, int64_t M, int64_t H, int64_t W,
                                   void* stream) {
    if (int rc = c2_check(B, C, M, H, W)) return rc;
    if (!fine || !w || !coarse) return DN_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (conv2d_down_v2(fine, w, coarse, B, C, M, H, W, s)) {
        DN_LAUNCH_CHECK();
        return 0;
    }
    const int64_t tiles = B * ((H * W + C2_TN - 1) / C2_TN);
    if (tiles >= (1ll << 31)) return DN_E_UNSUPPORTED;
    if (M <= 32) {
        hipLaunchKernelGGL((conv2d_k4s2_down_kernel<32>), dim3((unsigned)tiles, (unsigned)((M + 31) / 32)), dim3(256), 0, s, fine, w, coarse, (int)B,
                           (int)C, (int)M, (int)H, (int)W);
    } else {
        hipLaunchKernelGGL((conv2d_k4s2_down_kernel<64>), dim3((unsigned)tiles, (unsigned)((M + 63) / 64)), dim3(256), 0, s, fine, w, coarse, (int)B,
                           (int)C, (int)M, (int)H, (int)W);
    }
    DN_LAUNCH_CHECK();
    return 0;
}

extern "C" int dn_conv2d_k4s2_up(const float* coarse, const float* w, float* fine, int64_t B, int64_t C, int64_t M, int64_t H, int64_t W,
                                 void* stream) {
    if (int rc = c2_check(B, C, M, H, W)) return rc;
    if (!fine || !w || !coarse) return DN_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (conv2d_up_v2(coarse, w, fine, B, C, M, H, W, s)) {
        DN_LAUNCH_CHECK();
        return 0;
    }
    const int64_t tiles = B * ((H * W + C2_TN - 1) / C2_TN);
    if (tiles >= (1ll << 31)) return DN_E_UNSUPPORTED;
    if (C <= 32) {
        hipLaunchKernelGGL((conv2d_k4s2_up_kernel<32>), dim3((unsigned)tiles, (unsigned)((C + 31) / 32)), dim3(256), 0, s, coarse, w, fine, (int)B,
                           (int)C, (int)M, (int)H, (int)W);
    } else {
        hipLaunchKernelGGL((conv2d_k4s2_up_kernel<64>), dim3((unsigned)tiles, (unsigned)((C + 63) / 64)), dim3(256), 0, s, coarse, w, fine, (int)B,
                           (int)C, (int)M, (int)H, (int)W);
    }
    DN_LAUNCH_CHECK();
    return 0;
}

extern "C" int64_t dn_conv2d_k4s2_wrw_workspace_bytes(int64_t B, int64_t C, int64_t M, int64_t H, int64_t W) {
    if (int rc = c2_check(B, C, M, H, W)) return rc;
    int nz, tpw;
    c2_wrw_plan(B, C, M, H, W, nz, tpw);
    return nz == 1 ? 0 : (int64_t)sizeof(float) * M * C * 16 * nz;
}

extern "C" int dn_conv2d_k4s2_wrw(const float* fine, const float* coarse, float* grad_weight, int64_t B, int64_t C, int64_t M, int64_t H,
                                  int64_t W, void* workspace, int64_t workspace_bytes, void* stream) {
    if (int rc = c2_check(B, C, M, H, W)) return rc;
    if (!fine || !coarse || !grad_weight) return DN_E_BADARG;
    int nz, tpw;
    c2_wrw_plan(B, C, M, H, W, nz, tpw);
    if (nz > 1 && (!workspace || workspace_bytes < dn_conv2d_k4s2_wrw_workspace_bytes(B, C, M, H, W))) return DN_E_WORKSPACE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float* part = nz == 1 ? grad_weight : static_cast<float*>(workspace);
    // (the K-split plan counts tiles of 64 flattened positions; the v2 kernel's tiles are 64 positions too: TR x TW blocks of one sample)
    if (!(conv2d_wrw_v2_ok(fine, coarse, C, H, W) && conv2d_wrw_v2(fine, coarse, part, B, C, M, H, W, nz, tpw, s)))
        hipLaunchKernelGGL(conv2d_k4s2_wrw_kernel, dim3((unsigned)((C + 3) / 4), (unsigned)((M + 63) / 64), (unsigned)nz), dim3(256), 0, s, fine, coarse,
                           part, (int)B, (int)C, (int)M, (int)H, (int)W, tpw);
    if (nz > 1) {
        const long nout = (long)M * C * 16;
        if (nz >= 128 && nout <= 16384)
            hipLaunchKernelGGL(conv2d_k4s2_wsum_wave_kernel, dim3((unsigned)((nout + 3) / 4)), dim3(256), 0, s, part, grad_weight, nz, nout);
        else
            hipLaunchKernelGGL(conv2d_k4s2_wsum_kernel, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, s, part, grad_weight, nz, nout);
    }
    DN_LAUNCH_CHECK();
    return 0;
}
