// dropout_adj(force_undirected=True) of torch_geometric 2.0.4, the edge dropout SartorrasEGNN.get_embeddings applies
// when dropout > 0 and the model is training (/root/reference/point_vs/models/geometric/egnn_satorras.py:320-323):
//     mask = rand(E) >= p;  mask[row > col] = False;  keep the masked edges, then append their reverses
// i.e. of every undirected pair only the (row <= col) copy is drawn, and both directions live or die together; the
// output lists the survivors first and their reverses behind them, with the edge attributes repeated.
// Third-party semantics (torch_geometric is absent from /root/reference): restated from its published source; the
// random stream is this library's own - Philox4x32-10 keyed on (seed, step), counter = edge id - so a run is
// reproducible from (seed, step) on any launch geometry, but NOT bit-matched to torch's generator: no parity vectors
// (SURVEY.md §8a Q7). Two passes around one exclusive scan: mark + scan -> positions (pos[E] = survivors), fill.
#include "common.h"
#include <hipcub/hipcub.hpp>

namespace {

__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
}

// uniform in [0, 1) with 24 bits, from Philox4x32-10(key = seed, counter = (edge id, step))
__device__ __forceinline__ float philox_uniform(uint64_t seed, uint64_t step, uint64_t e) {
    uint32_t c[4] = {(uint32_t)e, (uint32_t)(e >> 32), (uint32_t)step, (uint32_t)(step >> 32)};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return (float)(c[0] >> 8) * (1.0f / 16777216.0f);
}

__global__ void k_dropout_mark(const int64_t* __restrict__ ei, int E, float p, uint64_t seed, uint64_t step,
                               int32_t* __restrict__ flag) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e > E) return;
    if (e == E) { flag[e] = 0; return; }            // (the scan's last output is the number of survivors)
    const int64_t r = ei[e], c = ei[(size_t)E + e];
    flag[e] = (r <= c && philox_uniform(seed, step, (uint64_t)e) >= p) ? 1 : 0;
}

__global__ void k_dropout_fill(const int64_t* __restrict__ ei, const int64_t* __restrict__ ea, int A, int E,
                               const int32_t* __restrict__ pos, int K, int64_t* __restrict__ out_index,
                               int64_t* __restrict__ out_attr) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int q = pos[e];
    if (pos[e + 1] == q) return;                    // dropped
    const int64_t r = ei[e], c = ei[(size_t)E + e];
    // out_index [2][2K]: survivors, then their reverses
    out_index[q] = r;                 out_index[(size_t)2 * K + q] = c;
    out_index[(size_t)K + q] = c;     out_index[(size_t)2 * K + K + q] = r;
    for (int a = 0; a < A; ++a) {
        const int64_t v = ea[(size_t)e * A + a];
        out_attr[(size_t)q * A + a] = v;
        out_attr[(size_t)(K + q) * A + a] = v;
    }
}

}  // namespace

extern "C" size_t pvs_dropout_adj_workspace_bytes(int32_t n_edges) {
    size_t scan = 0;
    hipcub::DeviceScan::ExclusiveSum(nullptr, scan, (const int32_t*)nullptr, (int32_t*)nullptr, n_edges + 1, 0);
    return pvs_align_up((size_t)(n_edges + 1) * sizeof(int32_t), 256) + pvs_align_up(scan, 256) + 256;
}

extern "C" int pvs_dropout_adj_mark(const int64_t* edge_index, int32_t E, float p, uint64_t seed, uint64_t step,
                                    int32_t* pos, void* workspace, size_t workspace_bytes, pvs_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    PVS_REQUIRE(edge_index && pos && E >= 0, "pvs_dropout_adj_mark: bad arguments");
    PVS_REQUIRE(p >= 0.f && p < 1.f, "pvs_dropout_adj_mark: dropout probability %g not in [0, 1)", (double)p);
    PVS_REQUIRE(workspace_bytes >= pvs_dropout_adj_workspace_bytes(E), "pvs_dropout_adj_mark: workspace too small");
    PvsArena a(workspace, workspace_bytes);
    int32_t* flag = a.take<int32_t>((size_t)E + 1);
    size_t scan = 0;
    hipcub::DeviceScan::ExclusiveSum(nullptr, scan, (const int32_t*)nullptr, (int32_t*)nullptr, E + 1, 0);
    void* tmp = a.take<char>(scan);
    k_dropout_mark<<<(E + 1 + 255) / 256, 256, 0, stream>>>(edge_index, E, p, seed, step, flag);
    PVS_CHECK_LAUNCH();
    PVS_CHECK_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, scan, flag, pos, E + 1, stream));
    return 0;
}

extern "C" int pvs_dropout_adj_fill(const int64_t* edge_index, const int64_t* edge_attr, int32_t n_edge_attr,
                                    int32_t E, const int32_t* pos, int32_t n_kept, int64_t* out_index,
                                    int64_t* out_attr, pvs_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    PVS_REQUIRE(edge_index && pos && out_index && E >= 0 && n_kept >= 0, "pvs_dropout_adj_fill: bad arguments");
    PVS_REQUIRE(n_edge_attr == 0 || (edge_attr && out_attr), "pvs_dropout_adj_fill: edge_attr / out_attr NULL");
    if (E == 0 || n_kept == 0) return 0;
    k_dropout_fill<<<(E + 255) / 256, 256, 0, stream>>>(edge_index, edge_attr, n_edge_attr, E, pos, n_kept, out_index,
                                                        out_attr);
    PVS_CHECK_LAUNCH();
    return 0;
}
