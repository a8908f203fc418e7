// naf_host.h -- host-side helpers shared by the translation units of libnaf_hip.so.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "../../include/naf_hip.h"

namespace naf {

// Records `msg` as the calling thread's last error and returns `code` (see naf_last_error()).
int fail(int code, const char *msg);
// hipGetLastError() -> NAF_OK / NAF_ERR_LAUNCH (message names the kernel).
int check_launch(const char *kernel);

}  // namespace naf
