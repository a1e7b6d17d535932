// ABI bookkeeping entry points of libdiffnet_hip.so.
#include <cstdlib>
#include <cstring>

#include "dn_common.h"

#define DN_STR2(x) #x
#define DN_STR(x) DN_STR2(x)

extern "C" int dn_abi_version(void) { return DN_ABI_VERSION; }

extern "C" const char* dn_build_info(void) {
    return "libdiffnet_hip abi " DN_STR(DN_ABI_VERSION) " target gfx950 (CDNA4, wave64) hip " DN_STR(HIP_VERSION_MAJOR) "." DN_STR(
        HIP_VERSION_MINOR) " built " __DATE__;
}

// ---- tuning switches: one table, filled from DN_<KEY> when the library is loaded, changed only through dn_config_set ----
namespace {
const char* const kKeys[dn::CFG_COUNT] = {"PLAN2D", "PLAN3D", "PLAN_FSDT", "Q1_RULE_KERNEL", "GPE_GATHER", "GPE_TILED", "Q1_3D_T16", "Q1_3D_E1SUM", "FSDT_GENERIC", "Q1_3D_E1", "HANDOVER_SPIN_LIMIT", "CONV2D_V1", "CONV_WRW_WGS", "Q1_3D_N2", "FSDT_FORM"};
char g_cfg[dn::CFG_COUNT][64];

int key_index(const char* key) {
    if (!key) return -1;
    for (int k = 0; k < dn::CFG_COUNT; ++k)
        if (std::strcmp(key, kKeys[k]) == 0) return k;
    return -1;
}

struct ConfigInit {
    ConfigInit() {
        for (int k = 0; k < dn::CFG_COUNT; ++k) {
            char name[80] = "DN_";
            std::strncat(name, kKeys[k], sizeof(name) - 4);
            const char* e = std::getenv(name);                   // the ONLY getenv of the library: once, at load
            g_cfg[k][0] = 0;
            if (e && std::strlen(e) < sizeof(g_cfg[k])) std::strcpy(g_cfg[k], e);
        }
    }
} g_cfg_init;
}  // namespace

namespace dn {
const char* config(ConfigKey k) { return g_cfg[k][0] ? g_cfg[k] : nullptr; }
}  // namespace dn

extern "C" int dn_config_set(const char* key, const char* value) {
    const int k = key_index(key);
    if (k < 0) return DN_E_BADARG;
    if (value && std::strlen(value) >= sizeof(g_cfg[k])) return DN_E_BADARG;
    g_cfg[k][0] = 0;
    if (value) std::strcpy(g_cfg[k], value);
    return 0;
}

extern "C" const char* dn_config_get(const char* key) {
    const int k = key_index(key);
    return k < 0 ? nullptr : g_cfg[k];
}

// ---- sticky error word of a launch workspace (poisson_common.h: DN_WS_ERRWORD) ------------------------------------------------
extern "C" int dn_workspace_status(void* workspace, void* stream) {
    if (!workspace) return DN_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    unsigned* word = reinterpret_cast<unsigned*>(workspace) + 8;          // DN_WS_ERRWORD
    unsigned v = 0u;
    hipError_t e = hipMemcpyAsync(&v, word, sizeof(v), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return (int)e;
    if (v == 0u) return 0;
    e = hipMemsetAsync(word, 0, sizeof(v), s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return (int)e;
    return DN_E_HANDOVER;
}

// ---- Dirichlet mask images <-> DN_MASK_BITS (include/diffnet_hip.h) ----------------------------------------------------------
namespace dn {
__global__ void __launch_bounds__(256) pack_mask_bits_kernel(const void* __restrict__ mask, int is_u8, long long rows, int nx, int row_words,
                                                             uint32_t* __restrict__ bits) {
    const long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // one 32-bit word per thread
    if (w >= rows * row_words) return;
    const long long row = w / row_words;
    const int x0 = (int)(w % row_words) * 32;
    uint32_t v = 0u;
    for (int i = 0; i < 32 && x0 + i < nx; ++i) {
        const long long idx = row * nx + x0 + i;
        const bool set = is_u8 ? reinterpret_cast<const uint8_t*>(mask)[idx] != 0 : reinterpret_cast<const float*>(mask)[idx] > 0.5f;
        v |= set ? (1u << i) : 0u;
    }
    bits[w] = v;
}
__global__ void __launch_bounds__(256) unpack_mask_bits_kernel(const uint32_t* __restrict__ bits, long long rows, int nx, int row_words,
                                                               uint8_t* __restrict__ mask) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * nx) return;
    const long long row = i / nx;
    const int x = (int)(i % nx);
    mask[i] = (uint8_t)((bits[row * row_words + (x >> 5)] >> (x & 31)) & 1u);
}
}  // namespace dn

extern "C" int dn_pack_mask_bits(const void* mask, int32_t mask_kind, int64_t rows, int32_t nx, int32_t row_words, uint32_t* bits, void* stream) {
    if (!mask || !bits || rows < 0 || nx < 1 || row_words < (nx + 31) / 32 || (mask_kind != DN_MASK_F32 && mask_kind != DN_MASK_U8)) return DN_E_BADARG;
    const long long n = (long long)rows * row_words;
    if (n == 0) return 0;
    if ((n + 255) / 256 >= (1ll << 31)) return DN_E_UNSUPPORTED;
    hipLaunchKernelGGL(dn::pack_mask_bits_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), mask,
                       (int)(mask_kind == DN_MASK_U8), (long long)rows, (int)nx, (int)row_words, bits);
    DN_LAUNCH_CHECK();
    return 0;
}

extern "C" int dn_unpack_mask_bits(const uint32_t* bits, int64_t rows, int32_t nx, int32_t row_words, uint8_t* mask_u8, void* stream) {
    if (!mask_u8 || !bits || rows < 0 || nx < 1 || row_words < (nx + 31) / 32) return DN_E_BADARG;
    const long long n = (long long)rows * nx;
    if (n == 0) return 0;
    if ((n + 255) / 256 >= (1ll << 31)) return DN_E_UNSUPPORTED;
    hipLaunchKernelGGL(dn::unpack_mask_bits_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), bits,
                       (long long)rows, (int)nx, (int)row_words, mask_u8);
    DN_LAUNCH_CHECK();
    return 0;
}
