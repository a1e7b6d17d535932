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
const char* const kKeys[dn::CFG_COUNT] = {"PLAN2D", "PLAN3D", "PLAN_FSDT", "Q1_RULE_KERNEL", "GPE_GATHER", "Q1_3D_V1"};
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
