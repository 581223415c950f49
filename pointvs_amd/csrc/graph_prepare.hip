// pvs_graph_prepare: int64 COO + int64 one-hot edge_attr  ->  CSR (by row) + CSC (by col).
// Once per batch; every layer's forward and backward reuse the result.
#include "common.h"
#include "profile.h"
#include <hipcub/hipcub.hpp>

namespace {

__global__ void k_extract(const int64_t* __restrict__ ei, const int64_t* __restrict__ ea, int A,
                          int N, int E, int32_t* row32, int32_t* col32, int32_t* ids,
                          uint8_t* etype_in, int32_t* status) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    int64_t r = ei[e], c = ei[(size_t)E + e];
    int bad = 0;
    if (r < 0 || r >= N || c < 0 || c >= N) { bad |= 1; r = 0; c = 0; }
    row32[e] = (int32_t)r;
    col32[e] = (int32_t)c;
    ids[e] = e;
    if (A > 0) {
        int hot = -1, ones = 0, other = 0;
        for (int a = 0; a < A; ++a) {
            int64_t v = ea[(size_t)e * A + a];
            if (v == 1) { ones++; hot = a; }
            else if (v != 0) other = 1;
        }
        if (ones != 1 || other) { bad |= 2; if (hot < 0) hot = 0; }
        etype_in[e] = (uint8_t)hot;
    }
    if (bad) atomicOr(status, bad);
}

__global__ void k_gather_sorted(const int32_t* __restrict__ perm, const int32_t* __restrict__ col32,
                                const uint8_t* __restrict__ etype_in, int E, int32_t* col,
                                uint8_t* etype, int32_t* iota) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= E) return;
    int e = perm[p];
    col[p] = col32[e];
    if (etype) etype[p] = etype_in[e];
    iota[p] = p;
}

// ptr[r] = first position p with keys[p] >= r  (keys ascending), r in [0, N]
__global__ void k_lower_bounds(const int32_t* __restrict__ keys, int E, int N, int32_t* ptr,
                               float* inv_deg) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > N) return;
    auto lb = [&](int key) {
        int lo = 0, hi = E;
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if (keys[mid] < key) lo = mid + 1; else hi = mid;
        }
        return lo;
    };
    int p0 = lb(r);
    ptr[r] = p0;
    if (inv_deg && r < N) {
        int deg = lb(r + 1) - p0;
        inv_deg[r] = 1.0f / (float)(deg > 1 ? deg : 1);
    }
}

int key_bits(int n) {
    int b = 1;
    while ((1ll << b) < (long long)n) ++b;
    return b;
}

size_t sort_temp_bytes(int E, int bits) {
    size_t bytes = 0;
    hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const int32_t*)nullptr, (int32_t*)nullptr,
                                       (const int32_t*)nullptr, (int32_t*)nullptr, E, 0, bits, 0);
    return bytes;
}

struct PrepWs {
    int32_t *row32, *col32, *ids, *iota, *keys_tmp;
    uint8_t* etype_in;
    void* sort_tmp;
    size_t sort_bytes;
};

size_t carve(PvsArena& a, int N, int E, PrepWs* w) {
    size_t e = (size_t)(E > 0 ? E : 1);
    PrepWs t;
    t.row32 = a.take<int32_t>(e);
    t.col32 = a.take<int32_t>(e);
    t.ids = a.take<int32_t>(e);
    t.iota = a.take<int32_t>(e);
    t.keys_tmp = a.take<int32_t>(e);
    t.etype_in = a.take<uint8_t>(e);
    t.sort_bytes = sort_temp_bytes(E > 0 ? E : 1, key_bits(N > 1 ? N : 2));
    t.sort_tmp = a.take<char>(t.sort_bytes);
    if (w) *w = t;
    return a.off;
}

}  // namespace

extern "C" size_t pvs_graph_prepare_workspace_bytes(int32_t n_nodes, int32_t n_edges) {
    PvsArena a(nullptr, 0);
    return carve(a, n_nodes, n_edges, nullptr) + 256;
}

extern "C" int pvs_graph_prepare(const int64_t* edge_index, const int64_t* edge_attr,
                                 int32_t n_edge_attr, int32_t N, int32_t E, int32_t* rowptr,
                                 int32_t* row, int32_t* col, uint8_t* etype, int32_t* perm,
                                 int32_t* colptr, int32_t* cedge, float* inv_deg, int32_t* status,
                                 void* workspace, size_t workspace_bytes, pvs_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    PVS_REQUIRE(N > 0 && E >= 0, "pvs_graph_prepare: bad sizes N=%d E=%d", N, E);
    PVS_REQUIRE(n_edge_attr >= 0 && n_edge_attr <= 255, "pvs_graph_prepare: bad n_edge_attr %d",
                n_edge_attr);
    PVS_REQUIRE(n_edge_attr == 0 || (edge_attr && etype), "pvs_graph_prepare: edge_attr/etype NULL");
    PVS_REQUIRE((colptr == nullptr) == (cedge == nullptr), "pvs_graph_prepare: colptr and cedge go together");
    PvsArena arena(workspace, workspace_bytes);
    PrepWs w;
    carve(arena, N, E, &w);
    PVS_REQUIRE(arena.ok(), "pvs_graph_prepare: workspace too small (%zu < %zu)", workspace_bytes,
                arena.off);
    PvsProfScope prof(stream, PVS_PROF_PREPARE);
    PVS_CHECK_HIP(hipMemsetAsync(status, 0, sizeof(int32_t), stream));
    const int bits = key_bits(N > 1 ? N : 2);
    const int T = 256;
    if (E > 0) {
        k_extract<<<(E + T - 1) / T, T, 0, stream>>>(edge_index, edge_attr, n_edge_attr, N, E,
                                                     w.row32, w.col32, w.ids, w.etype_in, status);
        PVS_CHECK_LAUNCH();
        size_t tb = w.sort_bytes;
        PVS_CHECK_HIP(hipcub::DeviceRadixSort::SortPairs(w.sort_tmp, tb, w.row32, row, w.ids, perm,
                                                         E, 0, bits, stream));
        k_gather_sorted<<<(E + T - 1) / T, T, 0, stream>>>(perm, w.col32, w.etype_in, E, col,
                                                           n_edge_attr ? etype : nullptr, w.iota);
        PVS_CHECK_LAUNCH();
        if (cedge) {   // by-column lists: only the backward reads them
            tb = w.sort_bytes;
            PVS_CHECK_HIP(hipcub::DeviceRadixSort::SortPairs(w.sort_tmp, tb, col, w.keys_tmp, w.iota,
                                                             cedge, E, 0, bits, stream));
        }
    }
    k_lower_bounds<<<(N + 1 + T - 1) / T, T, 0, stream>>>(row, E, N, rowptr, inv_deg);
    PVS_CHECK_LAUNCH();
    if (cedge) {
        k_lower_bounds<<<(N + 1 + T - 1) / T, T, 0, stream>>>(w.keys_tmp, E, N, colptr, nullptr);
        PVS_CHECK_LAUNCH();
    }
    return 0;
}

// ---- pvs_graph_prepare_runs: the same result for an edge list with the reference's layout, without the
// by-row radix sort. `generate_edges` (preprocessing.py:108-142) emits, per graph, the inter-molecular
// block and then the intra-molecular block, each row-major (np.where order): TWO ROW-SORTED RUNS per
// graph, and PyG collation concatenates the graphs. A stable sort by row of such a list is a two-way
// merge per graph, and a merge of sorted runs is a counting problem: the sorted position of edge e of run
// X with row r is  rowptr[r] + [X is the second run] * count_first(r) + (e - first edge of X with row r).
// The counts come from binary searches over the runs (4 per node), not from a pass over the edges.
// The layout is a CONTRACT of the caller (graph.edge_layout == 'generate_edges'), never inferred; it is
// verified on the device (at most one descent per graph, every offset inside its row's count, rows inside
// the graph's node range) and a violation sets status bit 4 -> the host raises.
namespace {

// graph of edge e: one binary search per workgroup (for its first edge, shared through LDS), then a
// short walk - consecutive edges belong to the same or the next few graphs
__device__ __forceinline__ int graph_of_edge(const int32_t* __restrict__ edge_ptr, int n_graphs, int e, int E,
                                             int edges_per_thread = 1) {
    __shared__ int g_first;
    if (threadIdx.x == 0) {
        const int e0 = min((int)(blockIdx.x * blockDim.x) * edges_per_thread, E - 1);
        int lo = 0, hi = n_graphs;          // last g with edge_ptr[g] <= e0
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (edge_ptr[mid] <= e0) lo = mid; else hi = mid;
        }
        g_first = lo;
    }
    __syncthreads();
    int g = g_first;
    while (g + 1 < n_graphs && edge_ptr[g + 1] <= e) ++g;
    return g;
}

// One thread per PAIR of consecutive edges. The reference's format costs 40 bytes per edge here (int64 COO + int64
// one-hot [E, 3]) against 9 written, so the pass is a stream of loads: VEC = 16-byte loads (two int64 per request:
// rows, columns and the pair's 2A one-hot words; needs E even and 16-byte aligned arrays - the launcher checks).
template <bool VEC>
__global__ void k_extract_runs(const int64_t* __restrict__ ei, const int64_t* __restrict__ ea, int A, int N, int E,
                               int n_graphs, const int32_t* __restrict__ node_ptr, const int32_t* __restrict__ edge_ptr,
                               int32_t* __restrict__ row32, int32_t* __restrict__ col32, uint8_t* __restrict__ etype_in,
                               int32_t* __restrict__ split, int32_t* __restrict__ status) {
    const int e0 = 2 * (blockIdx.x * blockDim.x + threadIdx.x);
    int g = graph_of_edge(edge_ptr, n_graphs, min(e0, E - 1), E, 2);
    if (e0 >= E) return;
    const int n_here = e0 + 1 < E ? 2 : 1;
    int64_t rr[2], cc[2];
    if (VEC) {
        const longlong2 r2 = *reinterpret_cast<const longlong2*>(ei + e0);
        const longlong2 c2 = *reinterpret_cast<const longlong2*>(ei + (size_t)E + e0);
        rr[0] = r2.x; rr[1] = r2.y; cc[0] = c2.x; cc[1] = c2.y;
    } else {
        rr[0] = ei[e0]; cc[0] = ei[(size_t)E + e0];
        rr[1] = n_here > 1 ? ei[e0 + 1] : 0; cc[1] = n_here > 1 ? ei[(size_t)E + e0 + 1] : 0;
    }
    int64_t prev = e0 > 0 ? ei[e0 - 1] : 0;      // (the neighbouring thread's line: a cache hit)
    int hot[2] = {-1, -1}, attr_bad[2] = {0, 0};
    if (A > 0) {
        int ones[2] = {0, 0};
        if (VEC) {       // the pair's 2A words as A 16-byte loads: word w belongs to edge w / A, class w % A
            const longlong2* src = reinterpret_cast<const longlong2*>(ea + (size_t)e0 * A);
            for (int q = 0; q < A; ++q) {
                const longlong2 v = src[q];
                const int w0 = 2 * q, w1 = 2 * q + 1;
                const int ed0 = w0 >= A, ed1 = w1 >= A;
                if (v.x == 1) { ones[ed0]++; hot[ed0] = w0 - ed0 * A; } else if (v.x != 0) attr_bad[ed0] = 1;
                if (v.y == 1) { ones[ed1]++; hot[ed1] = w1 - ed1 * A; } else if (v.y != 0) attr_bad[ed1] = 1;
            }
        } else {
            for (int i = 0; i < n_here; ++i)
                for (int a = 0; a < A; ++a) {
                    const int64_t v = ea[(size_t)(e0 + i) * A + a];
                    if (v == 1) { ones[i]++; hot[i] = a; } else if (v != 0) attr_bad[i] = 1;
                }
        }
        for (int i = 0; i < 2; ++i) if (ones[i] != 1) attr_bad[i] = 1;
    }
    int bad = 0;
    int32_t r32[2] = {0, 0}, c32[2] = {0, 0};
    uint8_t ty[2] = {0, 0};
    for (int i = 0; i < n_here; ++i) {
        const int e = e0 + i;
        while (g + 1 < n_graphs && edge_ptr[g + 1] <= e) ++g;
        int64_t r = rr[i], c = cc[i];
        if (r < 0 || r >= N || c < 0 || c >= N) { bad |= 1; r = 0; c = 0; }
        if (r < node_ptr[g] || r >= node_ptr[g + 1] || c < node_ptr[g] || c >= node_ptr[g + 1]) bad |= 4;
        // the tables themselves: they must cover exactly [0, E) and [0, N) (an edge list edited after collation with
        // stale per-graph counts would otherwise be placed outside its rows' slots); monotone segments follow from
        // the per-edge range checks above, because every edge is tested against the segment the walk assigns it
        if (e < edge_ptr[g] || e >= edge_ptr[g + 1]) bad |= 4;
        if (e == 0 && (edge_ptr[0] != 0 || edge_ptr[n_graphs] != E || node_ptr[0] != 0 || node_ptr[n_graphs] != N)) bad |= 4;
        if (e > edge_ptr[g] && prev > rr[i]) {         // a descent inside the graph: the second run starts here
            const int old = atomicMin(&split[g], e);
            if (old != edge_ptr[g + 1]) bad |= 4;          // more than one descent: not two sorted runs
        }
        prev = rr[i];
        r32[i] = (int32_t)r; c32[i] = (int32_t)c;
        if (A > 0) {
            if (attr_bad[i]) { bad |= 2; if (hot[i] < 0) hot[i] = 0; }
            ty[i] = (uint8_t)hot[i];
        }
    }
    if (n_here == 2) {
        *reinterpret_cast<int2*>(row32 + e0) = make_int2(r32[0], r32[1]);
        *reinterpret_cast<int2*>(col32 + e0) = make_int2(c32[0], c32[1]);
        if (A > 0) *reinterpret_cast<uchar2*>(etype_in + e0) = make_uchar2(ty[0], ty[1]);
    } else {
        row32[e0] = r32[0]; col32[e0] = c32[0];
        if (A > 0) etype_in[e0] = ty[0];
    }
    if (bad) atomicOr(status, bad);
}

__global__ void k_iota(int E, int32_t* iota) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < E) iota[p] = p;
}

__global__ void k_init_split(int n_graphs, const int32_t* __restrict__ edge_ptr, int32_t* __restrict__ split,
                             int32_t* __restrict__ status) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g == 0) *status = 0;                           // (instead of a memset launch of its own)
    if (g < n_graphs) split[g] = edge_ptr[g + 1];      // no descent: the second run is empty
}

// per node: where its edges start in the first / second run of its graph, how many the first run has, degree
__global__ void k_run_starts(const int32_t* __restrict__ row32, int N, int E, int n_graphs,
                             const int32_t* __restrict__ status, const int32_t* __restrict__ node_ptr,
                             const int32_t* __restrict__ edge_ptr, const int32_t* __restrict__ split,
                             int32_t* __restrict__ start_a, int32_t* __restrict__ start_b, int32_t* __restrict__ cnt_a,
                             int32_t* __restrict__ rowptr, float* __restrict__ inv_deg) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    if (n == N - 1) rowptr[N] = E;
    if (*status & 4) {      // broken contract (all of it is known after k_extract_runs): a SAFE graph - every
        // edge a self-edge of node 0 - so that nothing downstream indexes out of bounds before the host raises
        start_a[n] = 0; start_b[n] = 0; cnt_a[n] = 0;
        rowptr[n] = n == 0 ? 0 : E;
        inv_deg[n] = 1.0f;
        return;
    }
    int lo = 0, hi = n_graphs;          // graph of node n
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (node_ptr[mid] <= n) lo = mid; else hi = mid;
    }
    const int g = lo;
    // first position in a run with row >= n and with row >= n + 1, for both runs: FOUR binary searches advanced in
    // lockstep (their loads are independent, so the node pays the latency of one descent, not of four in a row)
    const int a0 = edge_ptr[g], a1 = split[g], b1 = edge_ptr[g + 1];
    int f0 = a0, l0 = a1, f1 = a0, l1 = a1, f2 = a1, l2 = b1, f3 = a1, l3 = b1;
    while (f0 < l0 || f1 < l1 || f2 < l2 || f3 < l3) {
        const int m0 = (f0 + l0) >> 1, m1 = (f1 + l1) >> 1, m2 = (f2 + l2) >> 1, m3 = (f3 + l3) >> 1;
        const int v0 = f0 < l0 ? row32[m0] : 0, v1 = f1 < l1 ? row32[m1] : 0;
        const int v2 = f2 < l2 ? row32[m2] : 0, v3 = f3 < l3 ? row32[m3] : 0;
        if (f0 < l0) { if (v0 < n) f0 = m0 + 1; else l0 = m0; }
        if (f1 < l1) { if (v1 < n + 1) f1 = m1 + 1; else l1 = m1; }
        if (f2 < l2) { if (v2 < n) f2 = m2 + 1; else l2 = m2; }
        if (f3 < l3) { if (v3 < n + 1) f3 = m3 + 1; else l3 = m3; }
    }
    const int sa = f0, sa_next = f1, sb = f2, sb_next = f3;
    start_a[n] = sa;
    start_b[n] = sb;
    cnt_a[n] = sa_next - sa;
    const int d = (sa_next - sa) + (sb_next - sb);
    // rowptr WITHOUT a prefix sum (round 5): the graph's edges fill [edge_ptr[g], edge_ptr[g + 1]) of the sorted list, and
    // those in front of row n are the (sa - a0) edges of run A and the (sb - a1) edges of run B with a smaller row
    rowptr[n] = a0 + (sa - a0) + (sb - a1);
    inv_deg[n] = 1.0f / (float)(d > 1 ? d : 1);
}

__global__ void k_place_runs(const int32_t* __restrict__ row32, const int32_t* __restrict__ col32,
                             const uint8_t* __restrict__ etype_in, int E, int n_graphs,
                             const int32_t* __restrict__ edge_ptr, const int32_t* __restrict__ split,
                             const int32_t* __restrict__ rowptr, const int32_t* __restrict__ start_a,
                             const int32_t* __restrict__ start_b, const int32_t* __restrict__ cnt_a,
                             int32_t* __restrict__ row, int32_t* __restrict__ col, uint8_t* __restrict__ etype,
                             int32_t* __restrict__ perm, const int32_t* __restrict__ status) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int g = graph_of_edge(edge_ptr, n_graphs, min(e, E - 1), E);
    if (e >= E) return;
    if (*status & 4) {      // the safe graph of k_run_starts
        row[e] = 0; col[e] = 0; perm[e] = e;
        if (etype) etype[e] = 0;
        return;
    }
    // (two sorted runs per graph with rows inside the graph's range - verified by k_extract_runs - make the
    // positions below a bijection onto [0, E): every slot is written exactly once)
    const int r = row32[e];
    const bool second = e >= split[g];
    const int off = second ? e - start_b[r] : e - start_a[r];
    const int local = second ? cnt_a[r] + off : off;
    const int pos = rowptr[r] + local;
    row[pos] = r;
    col[pos] = col32[e];
    if (etype) etype[pos] = etype_in[e];
    perm[pos] = e;
}


// ---- by-column lists of a batch of disjoint graphs WITHOUT a sort ("counting transpose", round 3) ----------
// cedge = the sorted positions grouped by column, ascending inside a column (what a stable sort by column of the
// row-sorted list gives). The columns of a graph's edges lie inside the graph's node range (verified by
// k_extract_runs), so a column has at most max_graph_nodes distinct keys RELATIVE to its graph: a counting sort
// with one LDS table per wave. Every graph's positions are cut into `cpg` contiguous chunks, one wave per chunk:
//   pass 1  counts the chunk's edges per column                       (cnt[chunk][column], LDS atomics: integer
//           sums do not depend on their order)
//   totals  in-degree per node = sum over the graph's chunks, exclusive scan -> colptr
//   bases   cnt[chunk][column] <- colptr[column] + edges of that column in EARLIER chunks of the graph
//   pass 2  walks the chunk in position order, 64 positions per step: slot = running count of the column, which
//           starts at the chunk's base. Two lanes of one step with the same column (duplicate edges, two rows in one
//           step) take consecutive slots in LANE order - found by a byte table of "who wrote this column last" and
//           resolved group by group with ballots, so the order never depends on how the LDS serialises a conflict.
// Array for array the radix sort's output (tests/test_gpu_properties.py). Measured at cfg2 (10.2 M edges, 32 graphs,
// profiles/r03_ab_csc_counting_transpose.txt): pass 1 16 us, totals 8, scan 10, bases 11, pass 2 137 us = 182 us
// against 195 us for the two-pass pair sort + offsets. Pass 2 is bound by its 10 M scattered 4-byte stores: every
// store's bytes leave the L2 as a request of their own (stores write through: MI355X_MICROARCH.md, "all bytes leave L2
// every pass"), 650 MB of 64-byte requests at the fabric's rate - all of a graph's chunks on one XCD, or half as many
// graphs in flight per XCD, change nothing. PVS_CSC_SORT=1 selects the sort.
constexpr int kCscMaxCols = 4096;         // nodes per graph this path handles (one 16 KB + 4 KB table pair per wave)
constexpr int kCscWaves = 4;              // waves per workgroup
constexpr int kCscTargetChunks = 2048;    // chunks (= waves) per pass: 8 resident per CU
constexpr int kCscAhead = 8;              // steps of 64 positions whose columns are in flight together

struct CscChunk { int begin, end, node_lo, width, g; };

// The caller's edge_ptr / node_ptr are CLAMPED into [0, E] / [0, N] and made monotone here: k_extract_runs verifies the
// layout on the device and sets status bit 4, but these passes run before the host reads the status word, and a
// direct C-ABI caller may pass tables that do not even end at E. With consistent tables the clamps do nothing.
__device__ __forceinline__ CscChunk csc_chunk(int wid, int cpg, int stride, int n_graphs, int N, int E,
                                              const int32_t* __restrict__ node_ptr,
                                              const int32_t* __restrict__ edge_ptr) {
    CscChunk c;
    c.g = wid / cpg;
    const int k = wid - c.g * cpg;
    if (c.g >= n_graphs) { c.begin = c.end = 0; c.node_lo = 0; c.width = 0; return c; }
    const int e0 = min(max(edge_ptr[c.g], 0), E), e1 = min(max(edge_ptr[c.g + 1], e0), E);
    const int len = (((e1 - e0) + cpg - 1) / cpg + 63) & ~63;
    c.begin = min(e0 + k * len, e1);
    c.end = min(c.begin + len, e1);
    c.node_lo = min(max(node_ptr[c.g], 0), N);
    c.width = min(max(node_ptr[c.g + 1], c.node_lo), N) - c.node_lo;   // (the caller clamps it to the table and flags a graph that does not fit)
    return c;
}

// One wave walks its chunk in position order, 64 positions per step, and hands every position its slot: the running
// count of its column, which starts at tab[column]. Two lanes of one step with the same column (duplicate edges, two
// rows in one step) take consecutive slots in LANE order - found by a byte table of "who wrote this column last" and
// resolved group by group with ballots, so the order never depends on how the LDS serialises a conflict.
// T: int32 (global slots) or unsigned short (slots inside a tile's LDS list).
template <class T, class Sink>
__device__ __forceinline__ void csc_walk_chunk(const int32_t* __restrict__ col, const CscChunk& c, int lane, T* tab,
                                               unsigned char* tag, Sink&& sink) {
    // kCscAhead steps' columns are fetched together (one wave walks its chunk alone: a dependent load per step would
    // expose the whole memory latency 80 times per chunk)
    for (int q0 = c.begin; q0 < c.end; q0 += 64 * kCscAhead) {
        int keys[kCscAhead];
#pragma unroll
        for (int u = 0; u < kCscAhead; ++u)       // (unconditional loads of clamped positions: all in flight together)
            keys[u] = col[min(q0 + 64 * u + lane, c.end - 1)];
#pragma unroll
        for (int u = 0; u < kCscAhead; ++u) {
            // (a broken layout contract leaves columns outside the graph's range: clamped, so that every table access
            // and every slot stays in bounds; the host raises on the status word)
            const int k = min(max(keys[u] - c.node_lo, 0), c.width - 1);
            keys[u] = q0 + 64 * u + lane < c.end ? k : -1;
        }
#pragma unroll
        for (int u = 0; u < kCscAhead; ++u) {
            const int p = q0 + 64 * u + lane;
            const int key = keys[u];
            const bool live = key >= 0;
            if (q0 + 64 * u >= c.end) break;       // (wave-uniform)
            // lanes of this step that share a column: leader = the lowest lane, rank = lanes of the group below me
            int leader = lane, rank = 0, members = 1;
            if (live) tag[key] = (unsigned char)lane;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const bool lost = live && tag[key] != (unsigned char)lane;
            unsigned long long pending = __ballot(lost);
            while (pending) {
                const int l0 = __builtin_ctzll(pending);
                const int k0 = __builtin_amdgcn_readlane(key, l0);
                const unsigned long long grp = __ballot(live && key == k0);
                if (live && key == k0) {
                    leader = __builtin_ctzll(grp);
                    rank = __popcll(grp & ((1ull << lane) - 1ull));
                    members = __popcll(grp);
                }
                pending &= ~grp;
            }
            int slot = 0;
            if (live && rank == 0) {       // one lane per distinct column: no two lanes touch one counter
                slot = (int)tab[key];
                tab[key] = (T)(slot + members);
            }
            slot = __shfl(slot, leader, 64) + rank;
            if (live) sink(slot, p);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}

template <bool PLACE>
__global__ void __launch_bounds__(64 * kCscWaves)
k_csc_pass(const int32_t* __restrict__ col, int n_graphs, int N, int E, const int32_t* __restrict__ node_ptr,
           const int32_t* __restrict__ edge_ptr, int cpg, int stride, int32_t* __restrict__ cnt,
           int32_t* __restrict__ cedge, int32_t* __restrict__ status) {
    extern __shared__ int32_t csc_lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 shares an L2; speed only): all chunks of one
    // graph go to ONE XCD, so that the 4-byte slots scattered over the graph's part of cedge (1.3 MB at BASELINE
    // size) meet in one L2 and leave it as whole lines - spread over eight non-coherent L2s every slot was its own
    // masked write to memory (pass 2: 125 us for 10 M edges).
    const int bpg = cpg / kCscWaves;                   // workgroups per graph
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int g_of_block = (j / bpg) * 8 + xcd;
    const int wid = g_of_block * cpg + (j % bpg) * kCscWaves + wv;
    CscChunk c = csc_chunk(g_of_block < n_graphs ? wid : n_graphs * cpg, cpg, stride, n_graphs, N, E, node_ptr, edge_ptr);
    if (c.width > stride) {        // a graph larger than the caller's bound: contract violation, stay inside the table
        if (lane == 0) atomicOr(status, 4);
        c.width = stride;
    }
    if (c.width <= 0) return;
    if (PLACE && (*status & 4)) {      // broken contract: identity lists over the (clamped) chunk, colptr is all zeros
        for (int p = c.begin + lane; p < c.end; p += 64) cedge[p] = p;
        return;
    }
    int32_t* tab = csc_lds + wv * (stride + stride / 4);
    unsigned char* tag = reinterpret_cast<unsigned char*>(tab + stride);
    int32_t* mine = cnt + (size_t)wid * stride;
    for (int i = lane; i < c.width; i += 64) tab[i] = PLACE ? mine[i] : 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (PLACE) {
        csc_walk_chunk(col, c, lane, tab, tag, [&](int slot, int p) { cedge[min(max(slot, 0), E - 1)] = p; });   // (a slot outside [0, E) only under status bit 4)
        return;
    }
    // kCscAhead steps' columns are fetched together (one wave walks its chunk alone: a dependent load per step would
    // expose the whole memory latency 80 times per chunk)
    for (int q0 = c.begin; q0 < c.end; q0 += 64 * kCscAhead) {
        int keys[kCscAhead];
#pragma unroll
        for (int u = 0; u < kCscAhead; ++u)
            keys[u] = col[min(q0 + 64 * u + lane, c.end - 1)];
#pragma unroll
        for (int u = 0; u < kCscAhead; ++u) {
            if (q0 + 64 * u >= c.end) break;       // (wave-uniform)
            const int k = min(max(keys[u] - c.node_lo, 0), c.width - 1);
            if (q0 + 64 * u + lane < c.end) atomicAdd(&tab[k], 1);
        }
    }
    if (!PLACE) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int i = lane; i < c.width; i += 64) mine[i] = tab[i];
    }
}

// ---- pass 2 through LDS-sorted tiles (round 6) -------------------------------------------------------------------------
// k_csc_pass<true> stores every position at its slot as it walks: 10 M scattered 4-byte stores at BASELINE size, each of
// which leaves the L2 as a 64-byte write request of its own (650 MB of requests for 41 MB of data, 0.136 ms). Here a
// workgroup of NW waves takes NW CONSECUTIVE chunks of one graph (a "tile": ~40 k positions at BASELINE size), the waves
// walk their chunks exactly as before - same leader / rank resolution, so the same order inside a column - but place
// into a tile-local list in LDS (16-bit offsets from the tile's first position), ordered by column and, inside a column,
// by position. The list then leaves as one contiguous run per column (~19 entries = 76 bytes at BASELINE size) written
// by 32-lane halves: 12x fewer write requests. Everything a wave needs comes from the slots k_csc_colptr left in `cnt`
// (cnt[chunk][c] = the first cedge slot of column c's entries in that chunk): a column's entries of the tile go to
// [cnt[first chunk][c], cnt[first chunk of the next tile][c]), the chunks' shares follow from the differences.
// A tile longer than the LDS list (or than 16 bits) is placed the old way by the same workgroup.
template <int NW>
__global__ void __launch_bounds__(64 * NW)
k_csc_place_tiles(const int32_t* __restrict__ col, int n_graphs, int N, int E, const int32_t* __restrict__ node_ptr,
                  const int32_t* __restrict__ edge_ptr, int cpg, int stride, const int32_t* __restrict__ cnt,
                  const int32_t* __restrict__ colptr, int32_t* __restrict__ cedge, int32_t* __restrict__ status, int cap) {
    extern __shared__ int32_t csc_lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, tid = threadIdx.x;
    const int tpg = cpg / NW;                                  // tiles per graph
    const int g = blockIdx.x / tpg, tl = blockIdx.x - g * tpg;
    if (g >= n_graphs) return;
    const int k0 = g * cpg + tl * NW;                          // the tile's first chunk
    CscChunk c = csc_chunk(k0 + wv, cpg, stride, n_graphs, N, E, node_ptr, edge_ptr);
    if (c.width > stride) {
        if (lane == 0) atomicOr(status, 4);
        c.width = stride;
    }
    if (c.width <= 0) return;                                  // (workgroup-uniform: the width is the graph's)
    if (*status & 4) {                                         // broken contract: identity lists, as k_csc_pass
        for (int p = c.begin + lane; p < c.end; p += 64) cedge[p] = p;
        return;
    }
    const CscChunk first = csc_chunk(k0, cpg, stride, n_graphs, N, E, node_ptr, edge_ptr);
    const CscChunk last = csc_chunk(k0 + NW - 1, cpg, stride, n_graphs, N, E, node_ptr, edge_ptr);
    const int tile_begin = first.begin, tile_len = last.end - first.begin;
    const int width = c.width;
    if (tile_len > cap || tile_len > 65535) {
        // the old way: a table of global slots per wave (the LDS block is large enough: 5 bytes per column and wave)
        int32_t* tab = csc_lds + wv * (stride + stride / 4);
        unsigned char* tag = reinterpret_cast<unsigned char*>(tab + stride);
        const int32_t* mine = cnt + (size_t)(k0 + wv) * stride;
        for (int i = lane; i < width; i += 64) tab[i] = mine[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        csc_walk_chunk(col, c, lane, tab, tag, [&](int slot, int p) { cedge[min(max(slot, 0), E - 1)] = p; });
        return;
    }
    // LDS: lstart[stride + 64] | gbase[stride] | tab[NW][stride] (16 bit) | tag[NW][stride] (bytes) | list[cap] (16 bit)
    int32_t* lstart = csc_lds;
    int32_t* gbase = lstart + stride + 64;
    unsigned short* tabs = reinterpret_cast<unsigned short*>(gbase + stride);
    unsigned char* tags = reinterpret_cast<unsigned char*>(tabs + NW * stride);
    unsigned short* list = reinterpret_cast<unsigned short*>(tags + NW * stride);
    const bool last_tile = tl == tpg - 1;
    for (int i = tid; i < width; i += 64 * NW) {
        const int32_t* src = cnt + (size_t)k0 * stride + i;
        const int s0 = src[0];
        const int s_end = last_tile ? colptr[c.node_lo + i + 1] : src[(size_t)NW * stride];
        gbase[i] = s0;
        lstart[i] = s_end - s0;                                // the column's entries in this tile
#pragma unroll
        for (int k = 0; k < NW; ++k) tabs[k * stride + i] = (unsigned short)(src[(size_t)k * stride] - s0);   // ... in front of chunk k
    }
    __syncthreads();
    // exclusive scan of lstart[0 .. width): PER consecutive columns per thread, wave scan, carry over the waves
    {
        __shared__ int32_t wave_tot[NW];
        const int per = (width + 64 * NW - 1) / (64 * NW), a = tid * per;
        int sum = 0;
        for (int i = a; i < min(a + per, width); ++i) sum += lstart[i];
        int inc = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(inc, o, 64);
            if (lane >= o) inc += up;
        }
        if (lane == 63) wave_tot[wv] = inc;
        __syncthreads();
        int run = inc - sum;
        for (int k = 0; k < wv; ++k) run += wave_tot[k];
        for (int i = a; i < min(a + per, width); ++i) {
            const int v = lstart[i];
            lstart[i] = run;
            run += v;
        }
        if (tid == 64 * NW - 1) lstart[width] = run;          // (= tile_len with consistent tables)
    }
    __syncthreads();
    for (int i = tid; i < width; i += 64 * NW) {
        const int base = lstart[i];
#pragma unroll
        for (int k = 0; k < NW; ++k) tabs[k * stride + i] = (unsigned short)(tabs[k * stride + i] + base);
    }
    __syncthreads();
    csc_walk_chunk(col, c, lane, tabs + wv * stride, tags + wv * stride,
                   [&](int slot, int p) { list[min(max(slot, 0), cap - 1)] = (unsigned short)(p - tile_begin); });
    __syncthreads();
    // one run per column, written by 32-lane halves
    const int hw = tid >> 5, l32 = tid & 31;
    for (int cc = hw; cc < width; cc += 2 * NW) {
        const int a = lstart[cc], b = lstart[cc + 1], dst = gbase[cc];
        for (int i = a + l32; i < b; i += 32)
            cedge[min(max(dst + i - a, 0), E - 1)] = tile_begin + (int)list[min(i, cap - 1)];
    }
}

__device__ __forceinline__ int csc_graph_of_node(const int32_t* __restrict__ node_ptr, int n_graphs, int n) {
    int lo = 0, hi = n_graphs;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (node_ptr[mid] <= n) lo = mid; else hi = mid;
    }
    return lo;
}

int csc_chunks_per_graph(int n_graphs) {      // a multiple of the waves per workgroup
    int cpg = kCscTargetChunks / (n_graphs > 0 ? n_graphs : 1);
    cpg = (cpg / kCscWaves) * kCscWaves;
    if (cpg < kCscWaves) cpg = kCscWaves;
    if (cpg > 128) cpg = 128;
    return cpg;
}
int csc_blocks(int n_graphs, int cpg) { return ((n_graphs + 7) / 8) * (cpg / kCscWaves) * 8; }
int csc_stride(int max_graph_nodes) { return (max_graph_nodes + 63) & ~63; }
bool csc_counting_ok(int max_graph_nodes) { return max_graph_nodes > 0 && max_graph_nodes <= kCscMaxCols; }

struct RunsWs {
    int32_t *row32, *col32, *iota, *keys_tmp, *split, *start_a, *start_b, *cnt_a, *deg, *csc_cnt, *indeg;
    uint8_t* etype_in;
    void *sort_tmp, *scan_tmp;
    size_t sort_bytes, scan_bytes;
};

size_t carve_runs(PvsArena& a, int N, int E, int B, int max_graph_nodes, RunsWs* w) {
    const size_t e = (size_t)(E > 0 ? E : 1);
    RunsWs t;
    t.row32 = a.take<int32_t>(e);
    t.col32 = a.take<int32_t>(e);
    t.iota = a.take<int32_t>(e);
    t.keys_tmp = a.take<int32_t>(e);
    t.etype_in = a.take<uint8_t>(e);
    t.split = a.take<int32_t>((size_t)B + 1);
    t.start_a = a.take<int32_t>((size_t)N);
    t.start_b = a.take<int32_t>((size_t)N);
    t.cnt_a = a.take<int32_t>((size_t)N);
    t.deg = a.take<int32_t>((size_t)N + 1);
    t.sort_bytes = sort_temp_bytes(E > 0 ? E : 1, key_bits(N > 1 ? N : 2));
    t.sort_tmp = a.take<char>(t.sort_bytes);
    t.scan_bytes = 0;
    hipcub::DeviceScan::ExclusiveSum(nullptr, t.scan_bytes, (const int32_t*)nullptr, (int32_t*)nullptr, N + 1, 0);
    t.scan_tmp = a.take<char>(t.scan_bytes);
    t.csc_cnt = nullptr;
    t.indeg = nullptr;
    if (csc_counting_ok(max_graph_nodes)) {
        t.csc_cnt = a.take<int32_t>((size_t)B * csc_chunks_per_graph(B) * csc_stride(max_graph_nodes));
        t.indeg = a.take<int32_t>((size_t)N + 1);
    }
    if (w) *w = t;
    return a.off;
}

// indeg[n] = edges whose column is n (sum over the chunks of n's graph)
// (status bit 4 - a broken layout contract, final since k_extract_runs: every in-degree is 0, so colptr is all zeros
// and nothing downstream follows cedge; the host raises at its next poll of the status word)
__global__ void k_csc_totals(const int32_t* __restrict__ cnt, int N, int n_graphs, const int32_t* __restrict__ node_ptr,
                             int cpg, int stride, int32_t* __restrict__ indeg, const int32_t* __restrict__ status) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    if (*status & 4) { indeg[n] = 0; return; }
    const int g = csc_graph_of_node(node_ptr, n_graphs, n);
    const int c = n - node_ptr[g];
    int t0 = 0, t1 = 0, t2 = 0, t3 = 0;          // (cpg is a multiple of 4: independent loads in flight)
    if (c >= 0 && c < stride) {                  // (c < 0: node_ptr[0] > n, a broken table - status bit 4)
        const int32_t* src = cnt + (size_t)g * cpg * stride + c;
        for (int k = 0; k < cpg; k += 4) {
            t0 += src[(size_t)k * stride];
            t1 += src[(size_t)(k + 1) * stride];
            t2 += src[(size_t)(k + 2) * stride];
            t3 += src[(size_t)(k + 3) * stride];
        }
    }
    indeg[n] = (t0 + t1) + (t2 + t3);
}

// colptr and the per-chunk first slots WITHOUT a device-wide prefix sum (round 5; before: a library scan of the N + 1
// in-degrees = two launches, then k_csc_bases). A graph's columns hold exactly the graph's edges, which fill
// [edge_ptr[g], edge_ptr[g + 1]) of every sorted list, so colptr[n] = edge_ptr[g] + the in-degrees of the graph's columns
// in front of n: workgroup (bx, g) owns columns [256 bx, 256 bx + 256) of graph g (a graph has at most kCscMaxCols = 4096
// here), sums the in-degrees in front of its slice (at most 16 per thread), scans its own 256, and turns
// cnt[chunk][column] into the first cedge slot of that column's edges in that chunk, as k_csc_bases did.
__global__ void __launch_bounds__(256) k_csc_colptr(int32_t* __restrict__ cnt, const int32_t* __restrict__ indeg, int N, int E,
                                                    int n_graphs, const int32_t* __restrict__ node_ptr,
                                                    const int32_t* __restrict__ edge_ptr, int cpg, int stride,
                                                    int32_t* __restrict__ colptr, const int32_t* __restrict__ status) {
    __shared__ int32_t wave_tot[4], front_tot[4];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    if (*status & 4) {
        for (int i = (blockIdx.y * gridDim.x + blockIdx.x) * 256 + t; i <= N; i += gridDim.x * gridDim.y * 256) colptr[i] = 0;
        return;
    }
    // (grid.y is capped at 65535: batches of more graphs than that take several graphs per block row)
    for (int g = blockIdx.y; g < n_graphs; g += gridDim.y) {
        if (g == n_graphs - 1 && blockIdx.x == 0 && t == 0) colptr[N] = E;
        const int n0 = node_ptr[g], width = min(node_ptr[g + 1] - n0, stride);
        const int c0 = blockIdx.x * 256;
        if (c0 >= width) continue;             // (block-uniform)
        int front = 0;
        for (int c = t; c < c0; c += 256) front += indeg[n0 + c];
        const int c = c0 + t;
        const int mine = c < width ? indeg[n0 + c] : 0;
        int inc = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(inc, o, 64);
            if (lane >= o) inc += up;
            front += __shfl_xor(front, o, 64);
        }
        if (lane == 63) wave_tot[wv] = inc;
        if (lane == 0) front_tot[wv] = front;
        __syncthreads();
        int run = edge_ptr[g] + (front_tot[0] + front_tot[1]) + (front_tot[2] + front_tot[3]) + inc - mine;
        for (int k = 0; k < wv; ++k) run += wave_tot[k];
        if (c < width) {
            colptr[n0 + c] = run;
            // (cpg is a multiple of 4; four counts in flight per step - with the load and the store of one slot back to
            // back the compiler had to order every load behind the previous store: 64 dependent round trips per column,
            // 23.7 us at cfg2)
            int32_t* slot = cnt + (size_t)g * cpg * stride + c;
            for (int k = 0; k < cpg; k += 4) {
                const int v0 = slot[(size_t)k * stride], v1 = slot[(size_t)(k + 1) * stride];
                const int v2 = slot[(size_t)(k + 2) * stride], v3 = slot[(size_t)(k + 3) * stride];
                slot[(size_t)k * stride] = run;
                slot[(size_t)(k + 1) * stride] = run + v0;
                slot[(size_t)(k + 2) * stride] = run + v0 + v1;
                slot[(size_t)(k + 3) * stride] = run + v0 + v1 + v2;
                run += (v0 + v1) + (v2 + v3);
            }
        }
        __syncthreads();                       // the totals are rewritten by the next graph of this block row
    }
}

}  // namespace

extern "C" size_t pvs_graph_prepare_runs_workspace_bytes(int32_t n_nodes, int32_t n_edges, int32_t n_graphs,
                                                         int32_t max_graph_nodes) {
    PvsArena a(nullptr, 0);
    return carve_runs(a, n_nodes, n_edges, n_graphs, max_graph_nodes, nullptr) + 256;
}

extern "C" int pvs_graph_prepare_runs(const int64_t* edge_index, const int64_t* edge_attr, int32_t n_edge_attr,
                                      int32_t N, int32_t E, int32_t n_graphs, const int32_t* node_ptr,
                                      const int32_t* edge_ptr, int32_t* rowptr, int32_t* row, int32_t* col,
                                      uint8_t* etype, int32_t* perm, int32_t* colptr, int32_t* cedge, float* inv_deg,
                                      int32_t* status, int32_t max_graph_nodes, void* workspace,
                                      size_t workspace_bytes, pvs_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    PVS_REQUIRE(N > 0 && E >= 0 && n_graphs > 0, "pvs_graph_prepare_runs: bad sizes N=%d E=%d B=%d", N, E, n_graphs);
    PVS_REQUIRE(max_graph_nodes >= 0, "pvs_graph_prepare_runs: bad max_graph_nodes %d", max_graph_nodes);
    PVS_REQUIRE(n_edge_attr >= 0 && n_edge_attr <= 255, "pvs_graph_prepare_runs: bad n_edge_attr %d", n_edge_attr);
    PVS_REQUIRE(n_edge_attr == 0 || (edge_attr && etype), "pvs_graph_prepare_runs: edge_attr/etype NULL");
    PVS_REQUIRE(node_ptr && edge_ptr, "pvs_graph_prepare_runs: node_ptr / edge_ptr NULL");
    PVS_REQUIRE((colptr == nullptr) == (cedge == nullptr), "pvs_graph_prepare_runs: colptr and cedge go together");
    PvsArena arena(workspace, workspace_bytes);
    RunsWs w;
    carve_runs(arena, N, E, n_graphs, max_graph_nodes, &w);
    PVS_REQUIRE(arena.ok(), "pvs_graph_prepare_runs: workspace too small (%zu < %zu)", workspace_bytes, arena.off);
    PvsProfScope prof(stream, PVS_PROF_PREPARE);
    const int T = 256;
    k_init_split<<<(n_graphs + T - 1) / T > 0 ? (n_graphs + T - 1) / T : 1, T, 0, stream>>>(n_graphs, edge_ptr, w.split, status);
    PVS_CHECK_LAUNCH();
    if (E > 0) {
        const bool vec = (E & 1) == 0 && (((uintptr_t)edge_index | (uintptr_t)edge_attr) & 15) == 0;
        const int pairs = (E + 1) / 2;
        if (vec)
            k_extract_runs<true><<<(pairs + T - 1) / T, T, 0, stream>>>(edge_index, edge_attr, n_edge_attr, N, E, n_graphs,
                                                                        node_ptr, edge_ptr, w.row32, w.col32, w.etype_in,
                                                                        w.split, status);
        else
            k_extract_runs<false><<<(pairs + T - 1) / T, T, 0, stream>>>(edge_index, edge_attr, n_edge_attr, N, E, n_graphs,
                                                                         node_ptr, edge_ptr, w.row32, w.col32, w.etype_in,
                                                                         w.split, status);
        PVS_CHECK_LAUNCH();
    }
    k_run_starts<<<(N + T - 1) / T, T, 0, stream>>>(w.row32, N, E, n_graphs, status, node_ptr, edge_ptr, w.split, w.start_a,
                                                    w.start_b, w.cnt_a, rowptr, inv_deg);
    PVS_CHECK_LAUNCH();
    if (E > 0) {
        k_place_runs<<<(E + T - 1) / T, T, 0, stream>>>(w.row32, w.col32, w.etype_in, E, n_graphs, edge_ptr, w.split,
                                                        rowptr, w.start_a, w.start_b, w.cnt_a, row, col,
                                                        n_edge_attr ? etype : nullptr, perm, status);
        PVS_CHECK_LAUNCH();
    }
    if (!cedge) return 0;
    // by-column lists: only the backward reads them
    static const bool force_sort = [] { const char* e = getenv("PVS_CSC_SORT"); return e && e[0] == '1'; }();
    if (w.csc_cnt && !force_sort) {      // counting transpose (the caller bounds the graphs' sizes)
        const int cpg = csc_chunks_per_graph(n_graphs), stride = csc_stride(max_graph_nodes);
        const int waves = n_graphs * cpg, blocks = csc_blocks(n_graphs, cpg);
        const size_t lds = (size_t)kCscWaves * (stride + stride / 4) * sizeof(int32_t);
        if (lds > 48 * 1024) {
            PVS_CHECK_HIP(hipFuncSetAttribute((const void*)k_csc_pass<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            PVS_CHECK_HIP(hipFuncSetAttribute((const void*)k_csc_pass<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        }
        if (E > 0) {
            k_csc_pass<false><<<blocks, 64 * kCscWaves, lds, stream>>>(col, n_graphs, N, E, node_ptr, edge_ptr, cpg, stride,
                                                                        w.csc_cnt, nullptr, status);
            PVS_CHECK_LAUNCH();
        } else {
            PVS_CHECK_HIP(hipMemsetAsync(w.csc_cnt, 0, (size_t)waves * stride * sizeof(int32_t), stream));
        }
        k_csc_totals<<<(N + T - 1) / T, T, 0, stream>>>(w.csc_cnt, N, n_graphs, node_ptr, cpg, stride, w.indeg, status);
        PVS_CHECK_LAUNCH();
        k_csc_colptr<<<dim3((max_graph_nodes + 255) / 256, n_graphs < 65535 ? n_graphs : 65535), 256, 0, stream>>>(w.csc_cnt, w.indeg, N, E, n_graphs, node_ptr,
                                                                                         edge_ptr, cpg, stride, colptr, status);
        PVS_CHECK_LAUNCH();
        if (E > 0) {
            // placement through LDS-sorted tiles of NW consecutive chunks (round 6; PVS_CSC_TILES=0: the direct scatter)
            const char* tiles_env = getenv("PVS_CSC_TILES");
            const bool tiles_on = !(tiles_env && tiles_env[0] == '0');
            const int budget = 160 * 1024 - 256;              // (the CU's LDS less the kernel's static words)
            auto fixed_bytes = [&](int nw) { return (size_t)(stride + 64) * 4 + (size_t)stride * 4 + (size_t)nw * stride * 3; };
            int nw = (cpg % 8 == 0 && fixed_bytes(8) + 2 * 32768 <= (size_t)budget) ? 8 : 4;
            int cap = ((int)budget - (int)fixed_bytes(nw)) / 2;
            if (cap > 65535) cap = 65535;
            // worth it where a column's run in a tile is long enough to fill write requests: E / (N * tiles per graph)
            // entries on average - 20 at cfg2 (r = 10 A: 0.385 -> 0.33 ms of preparation per step), 5 at cfg3 (r = 6 A),
            // where the tables' set-up and the two extra barriers cost more than the shorter runs save (0.140 -> 0.156 ms:
            // profiles/r06_ab_csc_lds_sorted_tiles.txt)
            const bool forced = tiles_env && tiles_env[0] == '2';
            const double avg_run = (double)E / ((double)(N > 0 ? N : 1) * (double)(cpg / nw));
            if (tiles_on && (avg_run >= 10.0 || forced) && cap >= 8192 && (size_t)nw * (stride + stride / 4) * 4 <= (size_t)budget) {
                const int tiles = n_graphs * (cpg / nw);
                if (nw == 8) {
                    PVS_CHECK_HIP(hipFuncSetAttribute((const void*)k_csc_place_tiles<8>, hipFuncAttributeMaxDynamicSharedMemorySize, budget));
                    k_csc_place_tiles<8><<<tiles, 64 * 8, budget, stream>>>(col, n_graphs, N, E, node_ptr, edge_ptr, cpg, stride,
                                                                           w.csc_cnt, colptr, cedge, status, cap);
                } else {
                    PVS_CHECK_HIP(hipFuncSetAttribute((const void*)k_csc_place_tiles<4>, hipFuncAttributeMaxDynamicSharedMemorySize, budget));
                    k_csc_place_tiles<4><<<tiles, 64 * 4, budget, stream>>>(col, n_graphs, N, E, node_ptr, edge_ptr, cpg, stride,
                                                                           w.csc_cnt, colptr, cedge, status, cap);
                }
            } else {
                k_csc_pass<true><<<blocks, 64 * kCscWaves, lds, stream>>>(col, n_graphs, N, E, node_ptr, edge_ptr, cpg, stride,
                                                                           w.csc_cnt, cedge, status);
            }
            PVS_CHECK_LAUNCH();
        }
        return 0;
    }
    if (E > 0) {
        size_t tb = w.sort_bytes;
        const int bits = key_bits(N > 1 ? N : 2);
        k_iota<<<(E + 255) / 256, 256, 0, stream>>>(E, w.iota);
        PVS_CHECK_LAUNCH();
        PVS_CHECK_HIP(hipcub::DeviceRadixSort::SortPairs(w.sort_tmp, tb, col, w.keys_tmp, w.iota, cedge, E, 0,
                                                         bits, stream));
    }
    k_lower_bounds<<<(N + 1 + T - 1) / T, T, 0, stream>>>(w.keys_tmp, E, N, colptr, nullptr);
    PVS_CHECK_LAUNCH();
    return 0;
}

// CSC of an existing CSR (shared with the radius-graph builder): cedge = CSR positions grouped by
// column, stable; colptr = offsets.
namespace {
}  // namespace

size_t pvs_build_csc_workspace_bytes(int N, int E) {
    const size_t e = (size_t)(E > 0 ? E : 1);
    return 2 * pvs_align_up(e * sizeof(int32_t), 256) +
           pvs_align_up(sort_temp_bytes((int)e, key_bits(N > 1 ? N : 2)), 256) + 512;
}

int pvs_build_csc(hipStream_t stream, const int32_t* col, int E, int N, int32_t* colptr, int32_t* cedge,
                  void* workspace, size_t workspace_bytes) {
    PVS_REQUIRE(workspace_bytes >= pvs_build_csc_workspace_bytes(N, E) - 512, "build_csc: workspace too small");
    PvsArena a(workspace, workspace_bytes);
    const size_t e = (size_t)(E > 0 ? E : 1);
    int32_t* iota = a.take<int32_t>(e);
    int32_t* keys = a.take<int32_t>(e);
    const int bits = key_bits(N > 1 ? N : 2);
    size_t tb = sort_temp_bytes((int)e, bits);
    void* tmp = a.take<char>(tb);
    if (E > 0) {
        k_iota<<<(E + 255) / 256, 256, 0, stream>>>(E, iota);
        PVS_CHECK_LAUNCH();
        PVS_CHECK_HIP(hipcub::DeviceRadixSort::SortPairs(tmp, tb, col, keys, iota, cedge, E, 0, bits, stream));
    }
    k_lower_bounds<<<(N + 1 + 255) / 256, 256, 0, stream>>>(keys, E, N, colptr, nullptr);
    PVS_CHECK_LAUNCH();
    return 0;
}

// ---- the ligand-touching edges of a CSR (screening, see pvs_graph_filter_ligand_edges) ----
namespace {
// wave per row: count (FILL = false) or copy (FILL = true) the edges whose row or column is a ligand atom
template <bool FILL>
__global__ void __launch_bounds__(256)
k_filter_ligand(PvsGraph g, const uint8_t* __restrict__ bp, int capacity, int32_t* __restrict__ cnt,
                const int32_t* __restrict__ rowptr_out, int32_t* __restrict__ row_out,
                int32_t* __restrict__ col_out, uint8_t* __restrict__ etype_out, int32_t* __restrict__ status) {
    const int lane = threadIdx.x & 63;
    const int i = (blockIdx.x * 256 + threadIdx.x) >> 6;
    if (i >= g.n_nodes) return;
    const bool lig_row = bp[i] == 0;
    const int p0 = g.rowptr[i], p1 = g.rowptr[i + 1];
    int done = 0;
    const int base = FILL ? rowptr_out[i] : 0;
    if (FILL && rowptr_out[g.n_nodes] > capacity) {
        if (i == 0 && lane == 0) atomicOr(status, 4);
        return;
    }
    for (int pb = p0; pb < p1; pb += 64) {
        const int p = pb + lane;
        int c = 0;
        bool keep = false;
        if (p < p1) {
            c = g.col[p];
            keep = lig_row || bp[c] == 0;
        }
        const unsigned long long m = __ballot(keep);
        if (FILL && keep) {
            const int q = base + done + __popcll(m & ((1ull << lane) - 1ull));
            row_out[q] = i;
            col_out[q] = c;
            if (etype_out) etype_out[q] = g.etype[p];
        }
        done += __popcll(m);
    }
    if (!FILL && lane == 0) cnt[i] = done;
}
}  // namespace

extern "C" size_t pvs_graph_filter_workspace_bytes(int32_t N) {
    size_t sb = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, sb, (const int32_t*)nullptr, (int32_t*)nullptr, N + 1);
    return pvs_align_up(((size_t)N + 1) * sizeof(int32_t), 256) + pvs_align_up(sb, 256) + 512;
}

extern "C" int pvs_graph_filter_ligand_edges(const PvsGraph* full, const uint8_t* bp, int32_t capacity,
                                             int32_t* rowptr_out, int32_t* row_out, int32_t* col_out,
                                             uint8_t* etype_out, int32_t* status, void* workspace,
                                             size_t workspace_bytes, pvs_stream_t stream_) {
    hipStream_t s = (hipStream_t)stream_;
    PVS_REQUIRE(full && bp && rowptr_out && row_out && col_out && status && full->rowptr && full->col,
                "pvs_graph_filter_ligand_edges: NULL");
    PVS_REQUIRE(!full->n_edges_dev, "pvs_graph_filter_ligand_edges: needs a graph with a host-side edge count");
    PVS_REQUIRE(etype_out == nullptr || full->etype, "pvs_graph_filter_ligand_edges: graph has no edge classes");
    const int N = full->n_nodes;
    PVS_REQUIRE(workspace_bytes >= pvs_graph_filter_workspace_bytes(N) - 512, "filter workspace too small");
    PvsArena a(workspace, workspace_bytes);
    int32_t* cnt = a.take<int32_t>((size_t)N + 1);
    size_t sb = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, sb, (const int32_t*)nullptr, (int32_t*)nullptr, N + 1);
    void* tmp = a.take<char>(sb);
    PVS_CHECK_HIP(hipMemsetAsync(status, 0, sizeof(int32_t), s));
    PVS_CHECK_HIP(hipMemsetAsync(cnt + N, 0, sizeof(int32_t), s));
    const int blocks = (N + 3) / 4;
    k_filter_ligand<false><<<blocks, 256, 0, s>>>(*full, bp, capacity, cnt, nullptr, nullptr, nullptr, nullptr, status);
    PVS_CHECK_LAUNCH();
    PVS_CHECK_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, sb, cnt, rowptr_out, N + 1, s));
    k_filter_ligand<true><<<blocks, 256, 0, s>>>(*full, bp, capacity, nullptr, rowptr_out, row_out, col_out,
                                                 etype_out, status);
    PVS_CHECK_LAUNCH();
    return 0;
}

namespace {
template <bool TO_INPUT>
__global__ void k_permute_rows(const float* __restrict__ src, float* __restrict__ dst,
                               const int32_t* __restrict__ perm, int E, int width) {
    // one thread per float4 (or scalar tail) of a row
    long long total = (long long)E * width;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        int e = (int)(i / width), c = (int)(i % width);
        int o = perm[e];
        if (TO_INPUT) dst[(size_t)o * width + c] = src[i];
        else dst[i] = src[(size_t)o * width + c];
    }
}
}  // namespace

extern "C" int pvs_rows_to_input_order(const float* src, float* dst, const int32_t* perm,
                                       int32_t E, int32_t width, pvs_stream_t stream) {
    if (E <= 0) return 0;
    long long total = (long long)E * width;
    int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    k_permute_rows<true><<<blocks, 256, 0, (hipStream_t)stream>>>(src, dst, perm, E, width);
    PVS_CHECK_LAUNCH();
    return 0;
}

extern "C" int pvs_rows_to_sorted_order(const float* src, float* dst, const int32_t* perm,
                                        int32_t E, int32_t width, pvs_stream_t stream) {
    if (E <= 0) return 0;
    long long total = (long long)E * width;
    int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    k_permute_rows<false><<<blocks, 256, 0, (hipStream_t)stream>>>(src, dst, perm, E, width);
    PVS_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// unsorted_segment_sum / unsorted_segment_mean (egnn_satorras.py:332-347) as standalone operators:
// stable sort of the segment ids, then one wave per segment sums its rows in that fixed order
// (deterministic, no atomics). The layers do not call these (their sums are fused into the edge
// kernels); they exist for the module surface.
namespace {

__global__ void k_seg_extract(const int64_t* __restrict__ ids, int E, int N, int32_t* key, int32_t* iota,
                              int32_t* status) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    int64_t v = ids[e];
    if (v < 0 || v >= N) { atomicOr(status, 1); v = 0; }
    key[e] = (int32_t)v;
    iota[e] = e;
}

__global__ void __launch_bounds__(256)
k_segment_reduce(const float* __restrict__ data, const int32_t* __restrict__ perm,
                 const int32_t* __restrict__ ptr, int N, int C, int mean, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int n_waves = (gridDim.x * blockDim.x) >> 6;
    for (int n = wave; n < N; n += n_waves) {
        const int p0 = ptr[n], p1 = ptr[n + 1];
        const float scale = mean ? 1.0f / (float)(p1 - p0 > 1 ? p1 - p0 : 1) : 1.0f;
        for (int c = lane; c < C; c += 64) {
            float acc = 0.f;
            for (int p = p0; p < p1; ++p) acc += data[(size_t)perm[p] * C + c];
            out[(size_t)n * C + c] = acc * scale;
        }
    }
}

__global__ void k_segment_expand(const float* __restrict__ g_out, const int64_t* __restrict__ ids,
                                 const int32_t* __restrict__ ptr, long long total, int C, int N, int mean,
                                 float* __restrict__ g_data) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(i / C), c = (int)(i % C);
        const int64_t id = ids[e];
        const int n = id < 0 || id >= N ? 0 : (int)id;     // (as k_seg_extract clamps; the forward set status bit 0)
        float s = 1.f;
        if (mean) { const int cnt = ptr[n + 1] - ptr[n]; s = 1.0f / (float)(cnt > 1 ? cnt : 1); }
        g_data[i] = g_out[(size_t)n * C + c] * s;
    }
}

}  // namespace

extern "C" size_t pvs_segment_workspace_bytes(int32_t n_rows, int32_t n_segments) {
    const size_t e = (size_t)(n_rows > 0 ? n_rows : 1);
    const size_t sort_bytes = sort_temp_bytes((int)e, key_bits(n_segments > 1 ? n_segments : 2));
    return 4 * pvs_align_up(e * sizeof(int32_t), 256) + pvs_align_up((size_t)(n_segments + 1) * 4, 256) +
           pvs_align_up(sort_bytes, 256) + 1024;
}

// forward: out[n,:] = sum (or mean) of data[e,:] with ids[e] == n.  ptr_out (int32 [N+1], caller
// owned) receives the segment offsets and is what pvs_segment_reduce_bwd needs.
extern "C" int pvs_segment_reduce_fwd(const float* data, const int64_t* ids, int32_t E, int32_t C,
                                      int32_t N, int32_t mean, float* out, int32_t* ptr_out,
                                      int32_t* status, void* workspace, size_t workspace_bytes,
                                      pvs_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    PVS_REQUIRE(N > 0 && E >= 0 && C > 0, "pvs_segment_reduce_fwd: bad sizes");
    PVS_REQUIRE(workspace_bytes >= pvs_segment_workspace_bytes(E, N), "segment workspace too small");
    PvsArena a(workspace, workspace_bytes);
    const size_t e = (size_t)(E > 0 ? E : 1);
    int32_t* key = a.take<int32_t>(e);
    int32_t* iota = a.take<int32_t>(e);
    int32_t* key_s = a.take<int32_t>(e);
    int32_t* perm = a.take<int32_t>(e);
    const int bits = key_bits(N > 1 ? N : 2);
    size_t sort_bytes = sort_temp_bytes((int)e, bits);
    void* sort_tmp = a.take<char>(sort_bytes);
    PVS_CHECK_HIP(hipMemsetAsync(status, 0, sizeof(int32_t), stream));
    if (E > 0) {
        k_seg_extract<<<(E + 255) / 256, 256, 0, stream>>>(ids, E, N, key, iota, status);
        PVS_CHECK_LAUNCH();
        PVS_CHECK_HIP(hipcub::DeviceRadixSort::SortPairs(sort_tmp, sort_bytes, key, key_s, iota, perm, E,
                                                         0, bits, stream));
    }
    k_lower_bounds<<<(N + 1 + 255) / 256, 256, 0, stream>>>(key_s, E, N, ptr_out, nullptr);
    PVS_CHECK_LAUNCH();
    int blocks = (N + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    k_segment_reduce<<<blocks, 256, 0, stream>>>(data, perm, ptr_out, N, C, mean, out);
    PVS_CHECK_LAUNCH();
    return 0;
}

extern "C" int pvs_segment_reduce_bwd(const float* g_out, const int64_t* ids, const int32_t* ptr,
                                      int32_t E, int32_t C, int32_t N, int32_t mean, float* g_data,
                                      pvs_stream_t stream) {
    PVS_REQUIRE(N > 0 && C > 0, "pvs_segment_reduce_bwd: bad sizes");
    if (E <= 0) return 0;
    const long long total = (long long)E * C;
    int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    k_segment_expand<<<blocks, 256, 0, (hipStream_t)stream>>>(g_out, ids, ptr, total, C, N, mean, g_data);
    PVS_CHECK_LAUNCH();
    return 0;
}
