// Error plumbing and version for the C ABI (include/quantool_amd.h).
#include <stdarg.h>
#include <stdlib.h>

#include "common.h"

static thread_local char g_err[512] = "";

void qt_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* qt_last_error(void) { return g_err; }

int qt_chain_prio() {
    const char* e = getenv("QT_CHAIN_PRIO");
    const int v = e ? atoi(e) : 3;   // measured in the bench: 0: 92.8-93.4 ms/step, 3: 92.5
    return v < 0 ? 0 : (v > 3 ? 3 : v);
}
extern "C" int qt_version(void) { return 400; }  // round * 100: bumped whenever a signature in include/quantool_amd.h changes

// ---- optional per-kernel timing with HIP events (bench.py's roofline leg) ---------------------
// Events are recorded on the launch stream immediately around the kernel, so the elapsed time is
// that kernel's device duration, not the enclosing API call.
#include <atomic>
#include <vector>

namespace {
struct ProfSlot {
    std::vector<hipEvent_t> ev;  // pairs
    size_t used = 0;
};
std::atomic<bool> g_prof_on{false};
std::mutex g_prof_mutex;  // the entry points may be called from several host threads (one per stream)
ProfSlot g_prof[QT_PROF_NUM_KERNELS];
}  // namespace

void qt_prof_mark(int kernel_id, hipStream_t stream) {
    if (!g_prof_on.load(std::memory_order_relaxed) || kernel_id < 0 || kernel_id >= QT_PROF_NUM_KERNELS) return;
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    ProfSlot& s = g_prof[kernel_id];
    if (s.used == s.ev.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return;
        s.ev.push_back(e);
    }
    (void)hipEventRecord(s.ev[s.used++], stream);
}

extern "C" int qt_profile_enable(int on) {
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    g_prof_on = on != 0;
    for (int k = 0; k < QT_PROF_NUM_KERNELS; ++k) g_prof[k].used = 0;
    return QT_OK;
}

extern "C" int qt_profile_read(int kernel_id, double* total_ms, int64_t* launches) {
    QT_CHECK_ARG(kernel_id >= 0 && kernel_id < QT_PROF_NUM_KERNELS && total_ms && launches,
                 "qt_profile_read: bad arguments");
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    ProfSlot& s = g_prof[kernel_id];
    double tot = 0.0;
    int64_t n = 0;
    for (size_t i = 0; i + 1 < s.used; i += 2) {
        QT_HIP(hipEventSynchronize(s.ev[i + 1]));
        float ms = 0.f;
        QT_HIP(hipEventElapsedTime(&ms, s.ev[i], s.ev[i + 1]));
        tot += ms;
        ++n;
    }
    s.used = 0;
    *total_ms = tot;
    *launches = n;
    return QT_OK;
}
