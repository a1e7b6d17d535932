// ABI bookkeeping entry points of libdiffnet_hip.so.
#include "dn_common.h"

#define DN_STR2(x) #x
#define DN_STR(x) DN_STR2(x)

extern "C" int dn_abi_version(void) { return DN_ABI_VERSION; }

extern "C" const char* dn_build_info(void) {
    return "libdiffnet_hip abi " DN_STR(DN_ABI_VERSION) " target gfx950 (CDNA4, wave64) hip " DN_STR(HIP_VERSION_MAJOR) "." DN_STR(
        HIP_VERSION_MINOR) " built " __DATE__;
}
