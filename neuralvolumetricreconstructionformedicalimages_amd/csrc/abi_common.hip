// abi_common.hip -- error reporting and version entry points of the C ABI (include/naf_hip.h).
#include <cstdio>
#include <cstring>
#include <mutex>

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

// ---- per-kernel event timing ------------------------------------------------------------------------------
namespace {
constexpr int kMaxSlots = 8192;
struct Slot { const char *name; hipEvent_t a, b; };
bool g_prof_on = false;
int g_prof_used = 0;
int g_prof_created = 0;
Slot g_slots[kMaxSlots];
std::mutex g_prof_mu;
}  // namespace

ProfScope::ProfScope(const char *kernel, hipStream_t stream) : slot_(-1), stream_(stream) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (g_prof_used >= kMaxSlots) return;
    if (g_prof_used >= g_prof_created) {
        if (hipEventCreate(&g_slots[g_prof_used].a) != hipSuccess || hipEventCreate(&g_slots[g_prof_used].b) != hipSuccess) return;
        g_prof_created = g_prof_used + 1;
    }
    slot_ = g_prof_used++;
    g_slots[slot_].name = kernel;
    (void)hipEventRecord(g_slots[slot_].a, stream_);
}

ProfScope::~ProfScope() {
    if (slot_ >= 0) (void)hipEventRecord(g_slots[slot_].b, stream_);
}

}  // namespace naf

extern "C" int naf_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(naf::g_prof_mu);
    naf::g_prof_on = on != 0;
    naf::g_prof_used = 0;
    return NAF_OK;
}

// Synchronises the recorded events and writes one line per kernel: "<name> <launches> <total_ms>\n".
extern "C" int naf_profile_collect(char *buf, size_t buflen) {
    if (!buf || buflen == 0) return naf::fail(NAF_ERR_INVALID_ARGUMENT, "profile_collect: null buffer");
    std::lock_guard<std::mutex> lk(naf::g_prof_mu);
    struct Agg { const char *name; int count; double ms; };
    Agg agg[64];
    int n_agg = 0;
    for (int i = 0; i < naf::g_prof_used; ++i) {
        float ms = 0.0f;
        if (hipEventSynchronize(naf::g_slots[i].b) != hipSuccess) continue;
        if (hipEventElapsedTime(&ms, naf::g_slots[i].a, naf::g_slots[i].b) != hipSuccess) continue;
        int k = 0;
        for (; k < n_agg; ++k) if (std::strcmp(agg[k].name, naf::g_slots[i].name) == 0) break;
        if (k == n_agg) { if (n_agg == 64) continue; agg[n_agg++] = {naf::g_slots[i].name, 0, 0.0}; }
        agg[k].count += 1;
        agg[k].ms += ms;
    }
    size_t off = 0;
    buf[0] = 0;
    for (int k = 0; k < n_agg; ++k) {
        const int w = std::snprintf(buf + off, buflen - off, "%s %d %.6f\n", agg[k].name, agg[k].count, agg[k].ms);
        if (w < 0 || (size_t)w >= buflen - off) break;
        off += (size_t)w;
    }
    naf::g_prof_used = 0;
    return NAF_OK;
}

extern "C" const char *naf_last_error(void) { return naf::g_last_error; }
extern "C" int naf_abi_version(void) { return 5; }      // 4: naf_hash_encode_workspace_bytes / _backward_ws, NAF_CFG_SCATTER_PAIR12; 5: naf_render_train_adam_draw
