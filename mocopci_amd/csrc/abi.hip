// abi.hip -- version / error strings / optional per-kernel event timing for libmocopci_hip.so.
#include <mutex>
#include <vector>

#include "common.h"

MCP_EXPORT int mcp_abi_version(void) { return MCP_ABI_VERSION; }

MCP_EXPORT const char *mcp_error_string(int code) {
    if (code == MCP_OK) return "ok";
    if (code == MCP_ERR_BAD_ARG) return "mocopci: bad argument (null pointer or non-positive dimension)";
    if (code == MCP_ERR_UNSUPPORTED) return "mocopci: size not supported by the compiled kernels";
    return hipGetErrorString((hipError_t)code);
}

// ---- instrumentation: hipEvent pairs around launches of ONE selected kernel id ----
namespace {
std::mutex g_mu;
int g_kernel = 0;
std::vector<hipEvent_t> g_events;  // begin/end pairs
std::vector<hipEvent_t> g_pool;
hipEvent_t take_event() {
    if (!g_pool.empty()) {
        hipEvent_t e = g_pool.back();
        g_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace

void mcp_prof_begin(int kernel_id, hipStream_t s) {
    if (g_kernel != kernel_id) return;  // common case: one relaxed int compare
    std::lock_guard<std::mutex> lk(g_mu);
    hipEvent_t e = take_event();
    (void)hipEventRecord(e, s);
    g_events.push_back(e);
}
void mcp_prof_end(int kernel_id, hipStream_t s) {
    if (g_kernel != kernel_id) return;
    std::lock_guard<std::mutex> lk(g_mu);
    hipEvent_t e = take_event();
    (void)hipEventRecord(e, s);
    g_events.push_back(e);
}

MCP_EXPORT int mcp_prof_enable(int kernel_id) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (hipEvent_t e : g_events) g_pool.push_back(e);
    g_events.clear();
    g_kernel = kernel_id;
    return MCP_OK;
}

MCP_EXPORT int mcp_prof_collect(int *launches, float *total_ms) {
    std::lock_guard<std::mutex> lk(g_mu);
    int n = 0;
    float tot = 0.f;
    for (size_t i = 0; i + 1 < g_events.size(); i += 2) {
        (void)hipEventSynchronize(g_events[i + 1]);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, g_events[i], g_events[i + 1]) == hipSuccess) {
            tot += ms;
            ++n;
        }
    }
    for (hipEvent_t e : g_events) g_pool.push_back(e);
    g_events.clear();
    if (launches) *launches = n;
    if (total_ms) *total_ms = tot;
    return MCP_OK;
}
