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

// ---- instrumentation: hipEvent pairs around launches of the selected kernel ids (bit mask) ----
namespace {
std::mutex g_mu;
unsigned g_mask = 0;  // bit k set = time kernel id k
struct Span {
    int id;
    hipEvent_t begin, end;
};
std::vector<Span> g_spans;
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
    if (!(g_mask & (1u << kernel_id))) return;  // common case: one load and test
    std::lock_guard<std::mutex> lk(g_mu);
    Span sp{kernel_id, take_event(), nullptr};
    (void)hipEventRecord(sp.begin, s);
    g_spans.push_back(sp);
}
void mcp_prof_end(int kernel_id, hipStream_t s) {
    if (!(g_mask & (1u << kernel_id))) return;
    std::lock_guard<std::mutex> lk(g_mu);
    for (size_t i = g_spans.size(); i-- > 0;) {  // innermost open span of this id (interp3 nests a knn launch)
        if (g_spans[i].id == kernel_id && g_spans[i].end == nullptr) {
            g_spans[i].end = take_event();
            (void)hipEventRecord(g_spans[i].end, s);
            return;
        }
    }
}

MCP_EXPORT int mcp_prof_enable(int kernel_mask) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (Span &sp : g_spans) {
        g_pool.push_back(sp.begin);
        if (sp.end) g_pool.push_back(sp.end);
    }
    g_spans.clear();
    g_mask = (unsigned)kernel_mask;
    return MCP_OK;
}

MCP_EXPORT int mcp_prof_collect(int kernel_id, int *launches, float *total_ms) {
    std::lock_guard<std::mutex> lk(g_mu);
    int n = 0;
    float tot = 0.f;
    for (Span &sp : g_spans) {
        if (sp.id != kernel_id || !sp.end) continue;
        (void)hipEventSynchronize(sp.end);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, sp.begin, sp.end) == hipSuccess) {
            tot += ms;
            ++n;
        }
    }
    if (launches) *launches = n;
    if (total_ms) *total_ms = tot;
    return MCP_OK;
}
