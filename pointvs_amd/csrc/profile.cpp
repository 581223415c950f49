#include <mutex>
#include <string.h>
#include <vector>
#include "profile.h"
#include "../../include/pvs_egnn.h"

namespace {
struct Rec { int id; hipEvent_t a, b; };
std::mutex g_mu;
unsigned g_mask = 0;      // bit 0: every category; bit (id + 1): category id
std::vector<Rec*> g_recs;
const char* kNames[PVS_PROF_COUNT] = {"edge_fwd", "edge_bwd", "col_gather", "graph_prepare", "edge_fwd_partial"};
thread_local int t_fwd_tag = PVS_PROF_EDGE_FWD;
}  // namespace

void pvs_prof_set_fwd_tag(int id) { t_fwd_tag = id; }
int pvs_prof_fwd_tag() { return t_fwd_tag; }

PvsProfScope::PvsProfScope(hipStream_t stream, int id) : s(stream), rec(nullptr) {
    if (!((g_mask & 1u) || ((g_mask >> (id + 1)) & 1u))) return;
    Rec* r = new Rec{id, nullptr, nullptr};
    if (hipEventCreate(&r->a) != hipSuccess || hipEventCreate(&r->b) != hipSuccess) { delete r; return; }
    (void)hipEventRecord(r->a, s);
    rec = r;
}

PvsProfScope::~PvsProfScope() {
    if (!rec) return;
    Rec* r = (Rec*)rec;
    (void)hipEventRecord(r->b, s);
    std::lock_guard<std::mutex> lk(g_mu);
    g_recs.push_back(r);
}

// on: 0 = off, 1 = every category, otherwise a mask with bit (id + 1) set for each category to record (an event pair
// costs the stream ~6 us of bubble per launch: bench.py times only the dominant kernel inside its timed region)
extern "C" int pvs_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_mask = (unsigned)on;
    return 0;
}

extern "C" int pvs_profile_reset(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (Rec* r : g_recs) { (void)hipEventDestroy(r->a); (void)hipEventDestroy(r->b); delete r; }
    g_recs.clear();
    return 0;
}

extern "C" int pvs_profile_read(const char* kernel, double* total_ms, int64_t* launches) {
    std::lock_guard<std::mutex> lk(g_mu);
    int id = -1;
    for (int i = 0; i < PVS_PROF_COUNT; ++i)
        if (strcmp(kernel, kNames[i]) == 0) id = i;
    if (id < 0) return -1;
    double tot = 0.0;
    int64_t n = 0;
    for (Rec* r : g_recs) {
        if (r->id != id) continue;
        if (hipEventSynchronize(r->b) != hipSuccess) return -2;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r->a, r->b) != hipSuccess) return -2;
        tot += ms;
        ++n;
    }
    *total_ms = tot;
    *launches = n;
    return 0;
}

extern "C" int pvs_profile_read_each(const char* kernel, double* out_ms, int64_t cap, int64_t* launches) {
    std::lock_guard<std::mutex> lk(g_mu);
    int id = -1;
    for (int i = 0; i < PVS_PROF_COUNT; ++i)
        if (strcmp(kernel, kNames[i]) == 0) id = i;
    if (id < 0) return -1;
    int64_t n = 0;
    for (Rec* r : g_recs) {
        if (r->id != id) continue;
        if (hipEventSynchronize(r->b) != hipSuccess) return -2;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r->a, r->b) != hipSuccess) return -2;
        if (n < cap && out_ms) out_ms[n] = ms;
        ++n;
    }
    *launches = n;
    return 0;
}
