// Version / error-string entry points of libasrk.so, and the optional per-kernel event profiler
// bench.py uses to time the dominant kernels live (HIP events on the launch stream).
#include "asrk_common.h"
#include "asrk_prof.h"

#include <mutex>
#include <vector>

extern "C" int asrk_version(void) { return ASRK_VERSION; }

extern "C" const char* asrk_strerror(int code) {
    switch (code) {
        case ASRK_OK: return "ok";
        case ASRK_ERR_NULL: return "required pointer is NULL";
        case ASRK_ERR_SHAPE: return "unsupported or inconsistent shape";
        case ASRK_ERR_ALIGN: return "pointer is not 16-byte aligned";
        case ASRK_ERR_HIP: return "HIP runtime call or kernel launch failed";
        case ASRK_ERR_WORKSPACE: return "workspace too small";
        case ASRK_ERR_UNSUPPORTED: return "dtype or mode not implemented";
        case ASRK_ERR_LENGTH: return "mel input features must be 3000 frames long";
        default: return "unknown asrk error";
    }
}

// ---------------------------------------------------------------------------------------------
namespace {
struct Rec { hipEvent_t a, b; int cat; double work; };
std::mutex g_mu;
bool g_on = false;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;

hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace

bool asrk_prof_enabled() { return g_on; }

int asrk_prof_begin(int cat, double work, hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_mu);
    Rec r{get_event(), get_event(), cat, work};
    if (!r.a || !r.b) return -1;
    (void)hipEventRecord(r.a, st);
    g_recs.push_back(r);
    return (int)g_recs.size() - 1;
}

void asrk_prof_end(int id, hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (id >= 0 && id < (int)g_recs.size()) (void)hipEventRecord(g_recs[id].b, st);
}

extern "C" int asrk_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_on = on != 0;
    return ASRK_OK;
}

// Sums the recorded launches of one category (the caller has synchronised the stream) and, when
// `reset` is set, recycles their events.  ms = total kernel time, work = total algorithmic work
// (flops or bytes, as the category defines), count = launches.
extern "C" int asrk_profile_read(int cat, double* ms, double* work, int* count, int reset) {
    std::lock_guard<std::mutex> lk(g_mu);
    double tms = 0.0, tw = 0.0;
    int n = 0;
    std::vector<Rec> keep;
    for (const Rec& r : g_recs) {
        if (r.cat == cat) {
            float e = 0.f;
            if (hipEventElapsedTime(&e, r.a, r.b) == hipSuccess) { tms += e; tw += r.work; ++n; }
            if (reset) { g_pool.push_back(r.a); g_pool.push_back(r.b); continue; }
        }
        keep.push_back(r);
    }
    if (reset) g_recs.swap(keep);
    if (ms) *ms = tms;
    if (work) *work = tw;
    if (count) *count = n;
    return ASRK_OK;
}
