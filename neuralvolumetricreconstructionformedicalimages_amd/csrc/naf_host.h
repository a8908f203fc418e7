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

// Optional per-kernel timing (naf_profile_enable / naf_profile_collect): a pair of HIP events on the launch stream
// around every kernel, so bench.py can time one kernel inside the real pipeline.  Costs nothing when disabled.
class ProfScope {
   public:
    ProfScope(const char *kernel, hipStream_t stream);
    ~ProfScope();
    ProfScope(const ProfScope &) = delete;
    ProfScope &operator=(const ProfScope &) = delete;

   private:
    int slot_;
    hipStream_t stream_;
};

}  // namespace naf
