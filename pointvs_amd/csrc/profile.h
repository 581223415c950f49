// Optional per-kernel timing with HIP events on the launch stream (bench.py's roofline leg).
#pragma once
#include <hip/hip_runtime.h>

enum { PVS_PROF_EDGE_FWD = 0, PVS_PROF_EDGE_BWD = 1, PVS_PROF_COL_GATHER = 2, PVS_PROF_PREPARE = 3,
       PVS_PROF_EDGE_FWD_PARTIAL = 4,   // the screening path's ligand-edges-only first layer
       PVS_PROF_COUNT = 5 };

// tag the edge forward launches of the calling thread carry (layer_api: full layer vs partial layer)
void pvs_prof_set_fwd_tag(int id);
int pvs_prof_fwd_tag();

struct PvsProfScope {
    hipStream_t s;
    void* rec;
    PvsProfScope(hipStream_t stream, int id);
    ~PvsProfScope();
};
