// abi_common.hip -- error reporting and version entry points of the C ABI (include/naf_hip.h).
#include <cstdio>
#include <cstring>

#include "naf_host.h"

namespace naf {

static thread_local char g_last_error[512] = "";

int fail(int code, const char *msg) {
    std::snprintf(g_last_error, sizeof(g_last_error), "%s", msg);
    return code;
}

int check_launch(const char *kernel) {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return NAF_OK;
    std::snprintf(g_last_error, sizeof(g_last_error), "%s: launch failed: %s", kernel, hipGetErrorString(e));
    return NAF_ERR_LAUNCH;
}

}  // namespace naf

extern "C" const char *naf_last_error(void) { return naf::g_last_error; }
extern "C" int naf_abi_version(void) { return 1; }
