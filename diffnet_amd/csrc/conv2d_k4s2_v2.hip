// Second generation of the 4 x 4 / stride-2 / padding-1 convolution family (round 4): the same three contractions as conv2d_k4s2.hip
// (down / up / wrw, fp32 MFMA, NCHW, one weight layout w[m][c][ky][kx]) with the operand that comes from the IMAGE staged as RAW ROW TILES
// in LDS instead of as im2col patches:
//
//   * round-2 kernels: every thread gathers the 16 taps of "its" position with 16 scalar loads per K-step (stride-2 addresses, 64 B segments),
//     writes them to LDS as a [k][position] matrix (16 ds_write_b32) -- 4 x the bytes of the image tile that holds them, and the loads of one
//     K-step (64 k = 1 us of MFMA work for a wave) are only ONE step ahead: the MFMA pipe was 51 % busy (profiles/r2_pmc_conv2d_down.txt);
//   * here: a workgroup tile is TR x TW output positions (128 of them) of one sample; per K-chunk of 4 fine channels the input rows
//     2 i0 - 1 .. 2 i0 + 2 TR of those channels are loaded ONCE, as 16-byte vectors, into LDS rows (5 vector loads + 5 ds_write_b128 per
//     thread and chunk for 128 positions instead of 16 + 16 for 64); an MFMA B-fragment is read straight out of the raw tile: lane (position j,
//     tap kx) of k-step (channel c, row ky) reads raw[c][2 i + ky][2 j + kx] -- consecutive lanes 2 floats apart, the four kx of a position
//     overlap its neighbour's (LDS broadcasts equal addresses);
//   * 64 (or 32) coarse channels x 128 positions per 256-thread workgroup, 32 x 64 per wave (2 x 4 MFMA tiles: 6 fragment reads per 8 MFMAs
//     instead of 4 per 4), a K-chunk = 128 MFMAs per wave = 2 us: the next chunk's loads (issued before the chunk's MFMAs) have landed when
//     they are written to the other LDS buffer; 2 workgroups per CU (69 KB of LDS each).
//
// Preconditions of the v2 forms (checked by the dn_conv2d_k4s2_* entry points, which fall back to the round-2 kernels otherwise):
// W % 16 == 0, H % TR == 0 for the tile shape of W (TW = min(W, 128), TR = 128 / TW), 16-byte aligned tensors.
// Reference layers: DiffNet/networks/unets.py:13-45, DiffNet/networks/autoencoders.py:24-45.
#include "dn_common.h"

namespace dn {

typedef float v2_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v2_f32x4 v2_mfma(float a, float b, v2_f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

constexpr int V2_NT = 128;         // output positions per workgroup tile
constexpr int V2_KC = 4;           // fine channels per K-chunk (K-chunk = 64)
constexpr int V2_SA = 68;          // LDS row stride of the weight tile [m][c * 16 + tap] (floats): 16-byte aligned rows, rows 4 banks apart

// =====================================================================================================================
// down v2:  coarse[b, m, i, j] = sum_{c, ky, kx} w[m, c, ky, kx] * fine[b, c, 2 i + ky - 1, 2 j + kx - 1]
// grid = (tiles_x * tiles_y * B, ceil(M / TM)), block = 256.  TW = tile width in positions (16 .. 128), TR = 128 / TW rows.
// Raw tile of one channel: RY = 2 TR + 2 rows (y = 2 i0 - 1 ...), RX = 2 TW + 8 columns (x = 2 j0 - 4 ...: 16-byte aligned vectors, the
// left halo column x = 2 j0 - 1 is element 3, the right one x = 2 j0 + 2 TW element 2 TW + 4).
// =====================================================================================================================
template <int TM, int TW, int NT>
__global__ void __launch_bounds__(256, 2) conv2d_down_v2_kernel(const float* __restrict__ fine, const float* __restrict__ w, float* __restrict__ coarse,
                                                                 int B, int C, int M, int H, int W) {
    constexpr int TR = NT / TW, RY = 2 * TR + 2, RX = 2 * TW + 8, RX4 = RX / 4;
    constexpr int CH = RY * RX;                          // floats per channel of the raw tile
    constexpr int NV = V2_KC * RY * RX4;                 // 16-byte vectors per chunk of the raw tile
    constexpr int NLB = (NV + 255) / 256;                // ... per thread
    constexpr int NLA = TM * 16 / 256;                   // weight vectors per thread and chunk (TM rows x 64 floats)
    constexpr int RT = TM / 32;                          // 16-row MFMA tiles per wave along m (2 waves along m)
    constexpr int ST = NT / 32;                          // 16-column tiles per wave along the positions (2 waves x NT / 2 positions)
    __shared__ __attribute__((aligned(16))) float As[2][TM * V2_SA];
    __shared__ __attribute__((aligned(16))) float Bs[2][V2_KC * CH];
    const int tid = threadIdx.x;
    const int tiles_x = W / TW, tiles_y = H / TR;
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int i0 = ty * TR, j0 = tx * TW, m0 = blockIdx.y * TM;
    const int W2 = 2 * W, H2 = 2 * H;
    const float* fb = fine + (size_t)b * C * H2 * W2;

    // ---- staging roles -------------------------------------------------------------------------------------------------
    // raw tile: vector v = tid + 256 r -> (channel cq, row ry, vector column rx4); out-of-image vectors are zero (the tile's column
    // origin 2 j0 - 4 and its width are multiples of 4 and W2 is one: a vector is inside or outside as a whole)
    unsigned boff[NLB];                                  // element offset inside the sample of channel c0 = 0 (clamped into the image)
    unsigned bmask[NLB];
    int bdst[NLB];
#pragma unroll
    for (int r = 0; r < NLB; ++r) {
        const int v = tid + 256 * r;
        const bool live = v < NV;
        const int vv = live ? v : 0;
        const int cq = vv / (RY * RX4), rem = vv % (RY * RX4), ry = rem / RX4, rx4 = rem % RX4;
        const int y = 2 * i0 - 1 + ry, x = 2 * j0 - 4 + 4 * rx4;
        const bool ok = live && y >= 0 && y < H2 && x >= 0 && x < W2;
        boff[r] = (unsigned)cq * (unsigned)(H2 * W2) + (unsigned)min(max(y, 0), H2 - 1) * (unsigned)W2 + (unsigned)min(max(x, 0), W2 - 4);
        bmask[r] = ok ? 0xffffffffu : 0u;
        asm volatile("" : "+v"(bmask[r]));                // opaque: keeps the compiler from sinking the load into a branch (see conv2d_k4s2.hip)
        bdst[r] = live ? (cq * CH + ry * RX + 4 * rx4) : -1;
    }
    // ---- MFMA roles -----------------------------------------------------------------------------------------------------
    const int lane = tid & 63, wv = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int wm = wv >> 1, wn = wv & 1;
    const int row0 = wm * (TM / 2);
    int boffs[ST];                                       // raw-tile offset of (position, tap kx = lk) at ky = 0, channel 0
#pragma unroll
    for (int s = 0; s < ST; ++s) {
        const int n = wn * (NT / 2) + 16 * s + li, ti = n / TW, tj = n % TW;
        boffs[s] = (2 * ti) * RX + 2 * tj + lk + 3;
    }
    const int aoff = (row0 + li) * V2_SA + lk;
    v2_f32x4 acc[RT][ST];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int s = 0; s < ST; ++s) acc[r][s] = (v2_f32x4){0.f, 0.f, 0.f, 0.f};

    float4 breg[NLB], areg[NLA];
    auto issue = [&](int c0) {
        const unsigned cbase = (unsigned)c0 * (unsigned)(H2 * W2);
        const int crem = C - c0;                         // channels of this chunk that exist
#pragma unroll
        for (int r = 0; r < NLB; ++r) {
            const int cq = (tid + 256 * r) / (RY * RX4);
            const bool cok = cq < crem;
            const float4 v = *reinterpret_cast<const float4*>(fb + (cok ? cbase + boff[r] : boff[r] % (unsigned)(H2 * W2)));
            const unsigned mk = cok ? bmask[r] : 0u;
            breg[r] = make_float4(__uint_as_float(__float_as_uint(v.x) & mk), __uint_as_float(__float_as_uint(v.y) & mk),
                                  __uint_as_float(__float_as_uint(v.z) & mk), __uint_as_float(__float_as_uint(v.w) & mk));
        }
        // weights: w[m0 + mm][c0 .. c0 + 3][16] = 64 contiguous floats per row (zero beyond M / C)
#pragma unroll
        for (int r = 0; r < NLA; ++r) {
            const int idx4 = tid + 256 * r, mm = idx4 >> 4, kk = (idx4 & 15) * 4;
            const int cc = c0 + (kk >> 4);
            areg[r] = (m0 + mm < M && cc < C) ? *reinterpret_cast<const float4*>(w + ((size_t)(m0 + mm) * C + cc) * 16 + (kk & 15))
                                              : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int r = 0; r < NLB; ++r)
            if (bdst[r] >= 0) *reinterpret_cast<float4*>(&Bs[buf][bdst[r]]) = breg[r];
#pragma unroll
        for (int r = 0; r < NLA; ++r) {
            const int idx4 = tid + 256 * r, mm = idx4 >> 4, kk = (idx4 & 15) * 4;
            *reinterpret_cast<float4*>(&As[buf][mm * V2_SA + kk]) = areg[r];
        }
    };

    const int nchunks = (C + V2_KC - 1) / V2_KC;
    issue(0);
    commit(0);
    __syncthreads();
    for (int st = 0; st < nchunks; ++st) {
        const int buf = st & 1;
        const bool more = st + 1 < nchunks;
        if (more) issue((st + 1) * V2_KC);
        const float* Ab = &As[buf][aoff];
        const float* Bb = &Bs[buf][0];
        // 16 k-steps (channel cq, row ky), fragments of step k + 1 read while the MFMAs of step k run
        float fa[2][RT], fbv[2][ST];
        auto frags = [&](int ks, float (&a)[RT], float (&bv)[ST]) {
            const int cq = ks >> 2, ky = ks & 3;
#pragma unroll
            for (int r = 0; r < RT; ++r) a[r] = Ab[16 * r * V2_SA + cq * 16 + ky * 4];
#pragma unroll
            for (int s = 0; s < ST; ++s) bv[s] = Bb[boffs[s] + cq * CH + ky * RX];
        };
        frags(0, fa[0], fbv[0]);
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            if (ks + 1 < 16) frags(ks + 1, fa[nxt], fbv[nxt]);
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int s = 0; s < ST; ++s) acc[r][s] = v2_mfma(fa[cur][r], fbv[cur][s], acc[r][s]);
        }
        if (more) commit(buf ^ 1);
        __syncthreads();
    }
    // D layout: column = lane & 15 (position), row = 4 (lane >> 4) + register (channel)
    float* ob = coarse + (size_t)b * M * H * W;
#pragma unroll
    for (int s = 0; s < ST; ++s) {
        const int n = wn * (NT / 2) + 16 * s + li, ti = n / TW, tj = n % TW;
        const unsigned po = (unsigned)(i0 + ti) * (unsigned)W + (unsigned)(j0 + tj);
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int m = m0 + row0 + 16 * r + 4 * lk + qq;
                if (m < M) ob[(size_t)m * (H * W) + po] = acc[r][s][qq];
            }
    }
}

template <int TM, int NT>
static bool down_v2_launch(const float* fine, const float* w, float* coarse, int B, int C, int M, int H, int W, hipStream_t s) {
    const int TW = W >= NT ? NT : W;                     // 16 .. NT (W % 16 == 0 checked by the caller; other widths: fall back)
    if (TW != 16 && TW != 32 && TW != 64 && TW != 128) return false;
    const int TR = NT / TW;
    if (W % TW != 0 || H % TR != 0) return false;
    const long long tiles = (long long)(W / TW) * (H / TR) * B;
    if (tiles >= (1ll << 31)) return false;
    const dim3 grid((unsigned)tiles, (unsigned)((M + TM - 1) / TM)), block(256);
    switch (TW) {
        case 16: hipLaunchKernelGGL((conv2d_down_v2_kernel<TM, 16, NT>), grid, block, 0, s, fine, w, coarse, B, C, M, H, W); break;
        case 32: hipLaunchKernelGGL((conv2d_down_v2_kernel<TM, 32, NT>), grid, block, 0, s, fine, w, coarse, B, C, M, H, W); break;
        case 64: hipLaunchKernelGGL((conv2d_down_v2_kernel<TM, 64, NT>), grid, block, 0, s, fine, w, coarse, B, C, M, H, W); break;
        default:
            if constexpr (NT == 128) hipLaunchKernelGGL((conv2d_down_v2_kernel<TM, 128, NT>), grid, block, 0, s, fine, w, coarse, B, C, M, H, W);
            else return false;
    }
    return true;
}

// Tile choice: 128 positions x 64 channels per workgroup where that still gives the 256 CUs ~2 workgroups each; layers with few positions
// (16^2 .. 32^2 coarse grids at batch 16) get 64-position tiles and, below that, 32-channel tiles: a launch of 128 workgroups leaves half the
// chip idle (256 -> 256 channels at 16^2: 180 us with 128 x 64 tiles against 143 us for the round-2 kernel's 64 x 64)
static inline long long v2_wgs(int64_t B, int64_t H, int64_t W, int64_t rows, int nt, int tm) { return B * ((H * W + nt - 1) / nt) * ((rows + tm - 1) / tm); }

// true when the v2 kernel was launched; false: the caller runs the round-2 kernel
bool conv2d_down_v2(const float* fine, const float* w, float* coarse, int64_t B, int64_t C, int64_t M, int64_t H, int64_t W, hipStream_t s) {
    if (config(CFG_CONV2D_V1) != nullptr) return false;
    if (W % 16 != 0 || (reinterpret_cast<uintptr_t>(fine) & 15) || (reinterpret_cast<uintptr_t>(w) & 15)) return false;
    const int b = (int)B, c = (int)C, m = (int)M, h = (int)H, ww = (int)W;
    if (M <= 32) return v2_wgs(B, H, W, M, 128, 32) >= 384 ? down_v2_launch<32, 128>(fine, w, coarse, b, c, m, h, ww, s)
                                                            : down_v2_launch<32, 64>(fine, w, coarse, b, c, m, h, ww, s);
    if (v2_wgs(B, H, W, M, 128, 64) >= 384) return down_v2_launch<64, 128>(fine, w, coarse, b, c, m, h, ww, s);
    if (v2_wgs(B, H, W, M, 64, 64) >= 384) return down_v2_launch<64, 64>(fine, w, coarse, b, c, m, h, ww, s);
    return down_v2_launch<32, 64>(fine, w, coarse, b, c, m, h, ww, s);
}

// =====================================================================================================================
// up v2:  fine[b, c, 2 I + py, 2 J + px] = sum_m sum_{ta, tb in {0, 1}} w[m][c][ky][kx] * coarse[b, m, I + py - ta, J + px - tb],
//         ky = (1 - py) + 2 ta, kx = (1 - px) + 2 tb  (ConvTranspose2d forward, Conv2d input gradient).
// One MFMA k-step = one coarse channel x its 2 x 2 taps; the four output parities share the staged coarse tile.
// Raw tile of one coarse channel: RY = TR + 2 rows (I0 - 1 ...), RX = TW + 8 columns (J0 - 4 ...: aligned 16-byte vectors).
// grid = (tiles * B, ceil(C / TC)), block = 256; wave tile 32 (TC = 64) or 16 (TC = 32) fine channels x 64 positions x 4 parities.
// =====================================================================================================================
template <int TC, int TW, int NT>
__global__ void __launch_bounds__(256, 2) conv2d_up_v2_kernel(const float* __restrict__ coarse, const float* __restrict__ w, float* __restrict__ fine,
                                                               int B, int C, int M, int H, int W) {
    constexpr int TR = NT / TW, RY = TR + 2, RX = TW + 8, RX4 = RX / 4;
    constexpr int CH = RY * RX;
    constexpr int NV = V2_KC * RY * RX4, NLB = (NV + 255) / 256;
    constexpr int NLA = V2_KC * TC * 16 / 4 / 256;       // weight vectors per thread and chunk
    constexpr int CSH = TC == 64 ? 6 : 5;
    constexpr int SW = TC + 8;                           // row stride of Ws[mm][tap][c]: taps two apart land 16 banks apart
    constexpr int RT = TC / 32, ST = NT / 32;
    __shared__ __attribute__((aligned(16))) float Ws[2][V2_KC * 16 * SW];
    __shared__ __attribute__((aligned(16))) float Ps[2][V2_KC * CH];
    const int tid = threadIdx.x;
    const int tiles_x = W / TW, tiles_y = H / TR;
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int I0 = ty * TR, J0 = tx * TW, c0 = blockIdx.y * TC;
    const int HW = H * W;
    const float* cb = coarse + (size_t)b * M * HW;

    unsigned boff[NLB], bmask[NLB];
    int bdst[NLB], bcq[NLB];
#pragma unroll
    for (int r = 0; r < NLB; ++r) {
        const int v = tid + 256 * r;
        const bool live = v < NV;
        const int vv = live ? v : 0;
        const int cq = vv / (RY * RX4), rem = vv % (RY * RX4), ry = rem / RX4, rx4 = rem % RX4;
        const int y = I0 - 1 + ry, x = J0 - 4 + 4 * rx4;
        const bool ok = live && y >= 0 && y < H && x >= 0 && x < W;
        boff[r] = (unsigned)min(max(y, 0), H - 1) * (unsigned)W + (unsigned)min(max(x, 0), W - 4);
        bmask[r] = ok ? 0xffffffffu : 0u;
        asm volatile("" : "+v"(bmask[r]));
        bdst[r] = live ? (cq * CH + ry * RX + 4 * rx4) : -1;
        bcq[r] = cq;
    }
    const int lane = tid & 63, wv = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int wm = wv >> 1, wn = wv & 1;
    const int row0 = wm * (TC / 2);
    const int ta = lk >> 1, tb = lk & 1;
    int boffs[ST];
#pragma unroll
    for (int s = 0; s < ST; ++s) {
        const int n = wn * (NT / 2) + 16 * s + li, ti = n / TW, tj = n % TW;
        boffs[s] = (ti + 1 - ta) * RX + tj + 4 - tb;     // + py * RX + px per parity
    }
    const int aoff = (8 * ta + 2 * tb) * SW + row0 + li; // + (4 (1 - py) + (1 - px)) * SW per parity, + mm * 16 * SW per channel
    v2_f32x4 acc[4][RT][ST];
#pragma unroll
    for (int par = 0; par < 4; ++par)
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int s = 0; s < ST; ++s) acc[par][r][s] = (v2_f32x4){0.f, 0.f, 0.f, 0.f};

    float4 breg[NLB], wreg[NLA];
    auto issue = [&](int mstep) {
#pragma unroll
        for (int r = 0; r < NLB; ++r) {
            const int m = mstep + bcq[r];
            const bool mok = m < M;
            const float4 v = *reinterpret_cast<const float4*>(cb + (size_t)(mok ? m : 0) * HW + boff[r]);
            const unsigned mk = mok ? bmask[r] : 0u;
            breg[r] = make_float4(__uint_as_float(__float_as_uint(v.x) & mk), __uint_as_float(__float_as_uint(v.y) & mk),
                                  __uint_as_float(__float_as_uint(v.z) & mk), __uint_as_float(__float_as_uint(v.w) & mk));
        }
#pragma unroll
        for (int r = 0; r < NLA; ++r) {
            const int idx4 = tid + 256 * r, mm = idx4 >> (CSH + 2), cc = (idx4 >> 2) & (TC - 1), t4 = (idx4 & 3) * 4;
            wreg[r] = (mstep + mm < M && c0 + cc < C) ? *reinterpret_cast<const float4*>(w + ((size_t)(mstep + mm) * C + c0 + cc) * 16 + t4)
                                                     : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int r = 0; r < NLB; ++r)
            if (bdst[r] >= 0) *reinterpret_cast<float4*>(&Ps[buf][bdst[r]]) = breg[r];
#pragma unroll
        for (int r = 0; r < NLA; ++r) {
            const int idx4 = tid + 256 * r, mm = idx4 >> (CSH + 2), cc = (idx4 >> 2) & (TC - 1), t4 = (idx4 & 3) * 4;
            float* d = &Ws[buf][(mm * 16 + t4) * SW + cc];
            d[0] = wreg[r].x; d[SW] = wreg[r].y; d[2 * SW] = wreg[r].z; d[3 * SW] = wreg[r].w;
        }
    };

    const int nchunks = (M + V2_KC - 1) / V2_KC;
    issue(0);
    commit(0);
    __syncthreads();
    for (int st = 0; st < nchunks; ++st) {
        const int buf = st & 1;
        const bool more = st + 1 < nchunks;
        if (more) issue((st + 1) * V2_KC);
        const float* Ab = &Ws[buf][aoff];
        const float* Bb = &Ps[buf][0];
        // 16 sub-steps (coarse channel mm, parity py, px), fragments of sub-step t + 1 read while the MFMAs of sub-step t run
        float fa[2][RT], fbv[2][ST];
        auto frags = [&](int ks, float (&a)[RT], float (&bv)[ST]) {
            const int mm = ks >> 2, py = (ks >> 1) & 1, px = ks & 1;
#pragma unroll
            for (int r = 0; r < RT; ++r) a[r] = Ab[(mm * 16 + 4 * (1 - py) + (1 - px)) * SW + 16 * r];
#pragma unroll
            for (int s = 0; s < ST; ++s) bv[s] = Bb[boffs[s] + mm * CH + py * RX + px];
        };
        frags(0, fa[0], fbv[0]);
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1, par = ks & 3;
            if (ks + 1 < 16) frags(ks + 1, fa[nxt], fbv[nxt]);
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int s = 0; s < ST; ++s) acc[par][r][s] = v2_mfma(fa[cur][r], fbv[cur][s], acc[par][r][s]);
        }
        if (more) commit(buf ^ 1);
        __syncthreads();
    }
    // epilogue: the lane's position (column) is the same in all four parity tiles: two float2 stores per channel (rows 2 I, 2 I + 1)
    const int W2 = 2 * W;
    float* fo = fine + (size_t)b * C * (4 * (size_t)HW);
#pragma unroll
    for (int s = 0; s < ST; ++s) {
        const int n = wn * (NT / 2) + 16 * s + li, ti = n / TW, tj = n % TW;
        const int II = I0 + ti, JJ = J0 + tj;
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

template <int TC, int NT>
static bool up_v2_launch(const float* coarse, const float* w, float* fine, int B, int C, int M, int H, int W, hipStream_t s) {
    const int TW = W >= NT ? NT : W;
    if (TW != 16 && TW != 32 && TW != 64 && TW != 128) return false;
    const int TR = NT / TW;
    if (W % TW != 0 || H % TR != 0) return false;
    const long long tiles = (long long)(W / TW) * (H / TR) * B;
    if (tiles >= (1ll << 31)) return false;
    const dim3 grid((unsigned)tiles, (unsigned)((C + TC - 1) / TC)), block(256);
    switch (TW) {
        case 16: hipLaunchKernelGGL((conv2d_up_v2_kernel<TC, 16, NT>), grid, block, 0, s, coarse, w, fine, B, C, M, H, W); break;
        case 32: hipLaunchKernelGGL((conv2d_up_v2_kernel<TC, 32, NT>), grid, block, 0, s, coarse, w, fine, B, C, M, H, W); break;
        case 64: hipLaunchKernelGGL((conv2d_up_v2_kernel<TC, 64, NT>), grid, block, 0, s, coarse, w, fine, B, C, M, H, W); break;
        default:
            if constexpr (NT == 128) hipLaunchKernelGGL((conv2d_up_v2_kernel<TC, 128, NT>), grid, block, 0, s, coarse, w, fine, B, C, M, H, W);
            else return false;
    }
    return true;
}

bool conv2d_up_v2(const float* coarse, const float* w, float* fine, int64_t B, int64_t C, int64_t M, int64_t H, int64_t W, hipStream_t s) {
    if (config(CFG_CONV2D_V1) != nullptr) return false;
    if (W % 16 != 0 || (reinterpret_cast<uintptr_t>(coarse) & 15) || (reinterpret_cast<uintptr_t>(w) & 15) || (reinterpret_cast<uintptr_t>(fine) & 7)) return false;
    const int b = (int)B, c = (int)C, m = (int)M, h = (int)H, ww = (int)W;
    if (C <= 32) return v2_wgs(B, H, W, C, 128, 32) >= 384 ? up_v2_launch<32, 128>(coarse, w, fine, b, c, m, h, ww, s)
                                                            : up_v2_launch<32, 64>(coarse, w, fine, b, c, m, h, ww, s);
    if (v2_wgs(B, H, W, C, 128, 64) >= 384) return up_v2_launch<64, 128>(coarse, w, fine, b, c, m, h, ww, s);
    if (v2_wgs(B, H, W, C, 64, 64) >= 384) return up_v2_launch<64, 64>(coarse, w, fine, b, c, m, h, ww, s);
    return up_v2_launch<32, 64>(coarse, w, fine, b, c, m, h, ww, s);
}

constexpr int V2W_NT = 64;         // positions per K-split plan tile (conv2d_k4s2.hip: c2_wrw_plan)
// =====================================================================================================================
// wrw v2:  gw[m][c][tap] = sum_{b, i, j} coarse[b][m][i][j] * fine[b][c][2 i + ky - 1][2 j + kx - 1]
// GEMM rows = m (64 per workgroup), columns = (c, tap) (NCW fine channels x 16 taps), K = positions: chunks of NT positions (TR x TW of one
// sample), the fine operand as raw row tiles (see the top of the file), the coarse operand as [m][position] rows (contiguous 16-byte vectors).
// Two shapes: NCW = 4 channels x NT = 64 positions (52 KB of LDS), and NCW = 8 x NT = 32 (37 KB): twice the columns per staged coarse tile,
// i.e. the coarse tensor -- re-read once per channel group -- is read half as often (layers with >= 8 fine channels).
// K is split over gridDim.z workgroups; partials [z][m][c][tap] are summed in index order by conv2d_k4s2_wsum_kernel (deterministic).
// =====================================================================================================================
template <int TW, int NT, int NCW>
__global__ void __launch_bounds__(256, 3) conv2d_wrw_v2_kernel(const float* __restrict__ fine, const float* __restrict__ coarse, float* __restrict__ part,
                                                                int B, int C, int M, int H, int W, int tiles_per_wg) {
    constexpr int TR = NT / TW, RY = 2 * TR + 2, RX = 2 * TW + 8, RX4 = RX / 4;
    constexpr int CH = RY * RX;
    constexpr int NV = NCW * RY * RX4, NLB = (NV + 255) / 256;
    constexpr int NLA = 64 * NT / 4 / 256;               // coarse vectors per thread and chunk (64 rows x NT positions)
    constexpr int SA = NT + 4;                           // row stride of Vs[m][position]
    constexpr int ST = NCW / 2;                          // 16-column MFMA tiles (= channels) per wave along the columns
    __shared__ __attribute__((aligned(16))) float Vs[2][64 * SA];
    __shared__ __attribute__((aligned(16))) float Rs[2][NCW * CH];
    const int tid = threadIdx.x;
    const int HW = H * W, H2 = 2 * H, W2 = 2 * W;
    const int tiles_x = W / TW, tiles_y = H / TR, tiles_s = tiles_x * tiles_y;
    const long ntiles = (long)B * tiles_s;
    const int m0 = blockIdx.y * 64, c0 = blockIdx.x * NCW;
    const int lane = tid & 63, wv = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int wm = wv >> 1, wn = wv & 1;
    const int row0 = wm * 32;
    const int ky = li >> 2, kx = li & 3;
    // raw-tile staging roles (tile-independent parts)
    int bry[NLB], brx[NLB], bdst[NLB], bcq[NLB];
#pragma unroll
    for (int r = 0; r < NLB; ++r) {
        const int v = tid + 256 * r;
        const bool live = v < NV;
        const int vv = live ? v : 0;
        const int cq = vv / (RY * RX4), rem = vv % (RY * RX4);
        bry[r] = rem / RX4; brx[r] = 4 * (rem % RX4); bcq[r] = cq;
        bdst[r] = live ? (cq * CH + bry[r] * RX + brx[r]) : -1;
    }
    const int boff = ky * RX + kx + 2 * lk + 3;          // + channel * CH + (2 ti) RX + 2 tj0 per k-step
    const int aoff = (row0 + li) * SA + lk;
    v2_f32x4 acc[2][ST];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int s = 0; s < ST; ++s) acc[r][s] = (v2_f32x4){0.f, 0.f, 0.f, 0.f};

    float4 breg[NLB], areg[NLA];
    auto issue = [&](int t) {
        const long tg = (long)blockIdx.z * tiles_per_wg + t;
        const bool tok = tg < ntiles;
        const long tt = tok ? tg : 0;
        const int b = (int)(tt / tiles_s), ts = (int)(tt % tiles_s), i0 = (ts / tiles_x) * TR, j0 = (ts % tiles_x) * TW;
        const float* fb = fine + (size_t)b * C * H2 * W2;
#pragma unroll
        for (int r = 0; r < NLB; ++r) {
            const int c = c0 + bcq[r];
            const int y = 2 * i0 - 1 + bry[r], x = 2 * j0 - 4 + brx[r];
            const bool ok = tok && c < C && y >= 0 && y < H2 && x >= 0 && x < W2 && bdst[r] >= 0;
            unsigned mk = ok ? 0xffffffffu : 0u;
            asm volatile("" : "+v"(mk));
            const float4 v = *reinterpret_cast<const float4*>(fb + (size_t)min(c, C - 1) * (H2 * W2) + (unsigned)min(max(y, 0), H2 - 1) * (unsigned)W2 +
                                                              (unsigned)min(max(x, 0), W2 - 4));
            breg[r] = make_float4(__uint_as_float(__float_as_uint(v.x) & mk), __uint_as_float(__float_as_uint(v.y) & mk),
                                  __uint_as_float(__float_as_uint(v.z) & mk), __uint_as_float(__float_as_uint(v.w) & mk));
        }
        // coarse[b][m0 + mm][tile]: TR rows of TW contiguous floats per channel -> Vs[mm][ti * TW + tj]
        const float* cbp = coarse + (size_t)b * M * HW + (size_t)i0 * W + j0;
#pragma unroll
        for (int r = 0; r < NLA; ++r) {
            const int idx4 = tid + 256 * r, mm = idx4 / (NT / 4), p4 = (idx4 % (NT / 4)) * 4, ti = p4 / TW, tj = p4 % TW;
            const bool ok = tok && m0 + mm < M;
            areg[r] = ok ? *reinterpret_cast<const float4*>(cbp + (size_t)(m0 + mm) * HW + ti * W + tj) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int r = 0; r < NLB; ++r)
            if (bdst[r] >= 0) *reinterpret_cast<float4*>(&Rs[buf][bdst[r]]) = breg[r];
#pragma unroll
        for (int r = 0; r < NLA; ++r) {
            const int idx4 = tid + 256 * r, mm = idx4 / (NT / 4), p4 = (idx4 % (NT / 4)) * 4;
            *reinterpret_cast<float4*>(&Vs[buf][mm * SA + p4]) = areg[r];
        }
    };
    issue(0);
    commit(0);
    __syncthreads();
    for (int t = 0; t < tiles_per_wg; ++t) {
        const int buf = t & 1;
        const bool more = t + 1 < tiles_per_wg;
        if (more) issue(t + 1);
        const float* Ab = &Vs[buf][aoff];
        const float* Bb = &Rs[buf][boff + (ST * wn) * CH];
        float fa[2][2], fbv[2][ST];
        auto frags = [&](int ks, float (&a)[2], float (&bv)[ST]) {      // k-step ks: positions 4 ks .. 4 ks + 3 of the tile
            const int ti = (4 * ks) / TW, tj0 = (4 * ks) % TW;
#pragma unroll
            for (int r = 0; r < 2; ++r) a[r] = Ab[16 * r * SA + 4 * ks];
#pragma unroll
            for (int s = 0; s < ST; ++s) bv[s] = Bb[s * CH + (2 * ti) * RX + 2 * tj0];
        };
        frags(0, fa[0], fbv[0]);
#pragma unroll
        for (int ks = 0; ks < NT / 4; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            if (ks + 1 < NT / 4) frags(ks + 1, fa[nxt], fbv[nxt]);
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int s = 0; s < ST; ++s) acc[r][s] = v2_mfma(fa[cur][r], fbv[cur][s], acc[r][s]);
        }
        if (more) commit(buf ^ 1);
        __syncthreads();
    }
    const size_t nout = (size_t)M * C * 16;
    float* po = part + (size_t)blockIdx.z * nout;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int s = 0; s < ST; ++s) {
            const int c = c0 + ST * wn + s;              // D column = lane & 15 = tap
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int m = m0 + row0 + 16 * r + 4 * lk + qq;
                if (m < M && c < C) po[((size_t)m * C + c) * 16 + li] = acc[r][s][qq];
            }
        }
}

// Channels per workgroup of the v2 weight-gradient kernel for a layer with C fine channels (8 from 8 channels on: see the kernel)
int conv2d_wrw_v2_ncw(int64_t C) { return C >= 8 ? 8 : 4; }

// launches the v2 weight-gradient kernel over `nz` K-slices of `tiles_per_wg` 64-position plan tiles each; false: preconditions not met
bool conv2d_wrw_v2(const float* fine, const float* coarse, float* part, int64_t B, int64_t C, int64_t M, int64_t H, int64_t W, int nz, int tiles_per_wg,
                   hipStream_t s) {
    const int ncw = conv2d_wrw_v2_ncw(C);
    const dim3 grid((unsigned)((C + ncw - 1) / ncw), (unsigned)((M + 63) / 64), (unsigned)nz), block(256);
    const int b = (int)B, c = (int)C, m = (int)M, h = (int)H, w = (int)W;
    if (ncw == 8) {                                      // chunks of 32 positions: two per plan tile
        const int TW = W >= 32 ? 32 : (int)W, tpw = 2 * tiles_per_wg;
        if (TW == 32) hipLaunchKernelGGL((conv2d_wrw_v2_kernel<32, 32, 8>), grid, block, 0, s, fine, coarse, part, b, c, m, h, w, tpw);
        else if (TW == 16) hipLaunchKernelGGL((conv2d_wrw_v2_kernel<16, 32, 8>), grid, block, 0, s, fine, coarse, part, b, c, m, h, w, tpw);
        else return false;
        return true;
    }
    const int TW = W >= 64 ? 64 : (int)W;
    switch (TW) {
        case 16: hipLaunchKernelGGL((conv2d_wrw_v2_kernel<16, 64, 4>), grid, block, 0, s, fine, coarse, part, b, c, m, h, w, tiles_per_wg); break;
        case 32: hipLaunchKernelGGL((conv2d_wrw_v2_kernel<32, 64, 4>), grid, block, 0, s, fine, coarse, part, b, c, m, h, w, tiles_per_wg); break;
        case 64: hipLaunchKernelGGL((conv2d_wrw_v2_kernel<64, 64, 4>), grid, block, 0, s, fine, coarse, part, b, c, m, h, w, tiles_per_wg); break;
        default: return false;
    }
    return true;
}

// are the preconditions of the v2 weight-gradient kernel met?  Shape-only part: the K-split plan (workspace size) depends on it
bool conv2d_wrw_v2_shape_ok(int64_t C, int64_t H, int64_t W) {
    if (config(CFG_CONV2D_V1) != nullptr) return false;
    if (W % 16 != 0) return false;
    if (conv2d_wrw_v2_ncw(C) == 8) {
        const int64_t TW = W >= 32 ? 32 : W;
        return (TW == 16 || TW == 32) && W % TW == 0 && H % (32 / TW) == 0 && (H * W) % 64 == 0;
    }
    const int64_t TW = W >= 64 ? 64 : W;
    if (TW != 16 && TW != 32 && TW != 64) return false;
    return W % TW == 0 && H % (V2W_NT / TW) == 0;
}
bool conv2d_wrw_v2_ok(const float* fine, const float* coarse, int64_t C, int64_t H, int64_t W) {
    return conv2d_wrw_v2_shape_ok(C, H, W) && !(reinterpret_cast<uintptr_t>(fine) & 15) && !(reinterpret_cast<uintptr_t>(coarse) & 15);
}

}  // namespace dn
