// pvs_radius_graph_*: protein-ligand radius graph straight from coordinates to the library's CSR/CSC
// (SURVEY.md §8f row 1). Restates generate_edges of the reference
// (/root/reference/point_vs/preprocessing/preprocessing.py:68-155, prune=False part):
//   distances = cdist(coords, coords)                       float64, sqrt(sum_k (a_k - b_k)^2)
//   inter block: pairs with 1e-7 < d < inter_radius and bp_i != bp_j        -> class 1
//   intra block: ALL pairs with 1e-7 < d < intra_radius (also inter-molecular ones, which therefore
//                appear twice - SURVEY Q6)                                   -> class 2 if both bp, else 0
//   edge order: inter block (np.where order: row-major) then intra block (row-major)
// and the int64 COO -> CSR step of pvs_graph_prepare in one go: the result is array-for-array what
// pvs_graph_prepare returns for the reference's edge list (stable by-row order = inter edges of the
// row with ascending col, then its intra edges; perm = position in the reference's order), without
// the [2,E] int64 + [E,3] int64 one-hot (40 B/edge) ever existing.
//
// Graphs are the whole-molecule blocks of a PyG batch (graph_ptr: node offsets); every row scans
// the nodes of its own graph in ascending order (brute force, O(n_g^2) distance tests per graph,
// exactly like cdist; n_g is a few thousand atoms), so columns come out sorted and no per-row sort
// is needed. Distances are evaluated in fp64 with the operation order of scipy's cdist and without
// FMA contraction, so the `<` decisions match the reference bit for bit.
#include "common.h"
#include "profile.h"
#include "radius_common.h"
#include <hipcub/hipcub.hpp>

int pvs_build_csc(hipStream_t stream, const int32_t* col, int E, int N, int32_t* colptr, int32_t* cedge,
                  void* workspace, size_t workspace_bytes);
size_t pvs_build_csc_workspace_bytes(int N, int E);

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = 4;
constexpr int kRowsPerWave = 16;
constexpr int kRowsPerBlock = kWaves * kRowsPerWave;

// block -> (graph, first row) and the per-graph offset of its neighbour bit masks
__global__ void k_block_table(const int32_t* __restrict__ gptr, int B, int32_t* __restrict__ blk_graph,
                              int32_t* __restrict__ blk_row0, int32_t* __restrict__ n_blocks_out,
                              long long* __restrict__ mask_off) {
    // one wave: graph g's blocks start at the prefix sum of ceil(n_g / kRowsPerBlock)
    const int lane = threadIdx.x;
    int nb_base = 0;
    long long mo_base = 0;
    for (int g0 = 0; g0 < B; g0 += 64) {
        const int g = g0 + lane;
        const int n = g < B ? gptr[g + 1] - gptr[g] : 0;
        int nb = (n + kRowsPerBlock - 1) / kRowsPerBlock;
        long long mw = (long long)n * ((n + 63) / 64);          // mask words (per kind) of the graph
        int nb_scan = nb;
        long long mw_scan = mw;
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(nb_scan, o, 64);
            const long long u = __shfl_up(mw_scan, o, 64);
            if (lane >= o) { nb_scan += t; mw_scan += u; }
        }
        if (g < B) {
            const int first = nb_base + nb_scan - nb;
            mask_off[g] = mo_base + mw_scan - mw;
            for (int k = 0; k < nb; ++k) {
                blk_graph[first + k] = g;
                blk_row0[first + k] = gptr[g] + k * kRowsPerBlock;
            }
        }
        nb_base += __shfl(nb_scan, 63, 64);
        mo_base += __shfl(mw_scan, 63, 64);
    }
    if (lane == 0) *n_blocks_out = nb_base;
}

// Brute-force pass: per (row, 64-column chunk of its graph) one 64-bit mask of inter edges and one
// of intra edges, plus the per-row counts. Block = kRowsPerBlock rows of one graph, wave =
// kRowsPerWave of them; the rows' coordinates sit in LDS as fp64, a lane keeps one column's
// coordinates in registers while the wave's rows go by.
__global__ void __launch_bounds__(kThreads)
k_radius_masks(const float* __restrict__ pos, const uint8_t* __restrict__ bp,
               const int32_t* __restrict__ gptr, const int32_t* __restrict__ blk_graph,
               const int32_t* __restrict__ blk_row0, const int32_t* __restrict__ n_blocks,
               const long long* __restrict__ mask_off, Radius r_inter, Radius r_intra, Radius r_zero,
               int pair_filter, unsigned long long* __restrict__ masks, int32_t* __restrict__ cnt_inter,
               int32_t* __restrict__ cnt_intra) {
    if ((int)blockIdx.x >= *n_blocks) return;
    __shared__ double rx[kRowsPerBlock], ry[kRowsPerBlock], rz[kRowsPerBlock];
    __shared__ int rbp[kRowsPerBlock];
    const int g = blk_graph[blockIdx.x];
    const int n0 = gptr[g], n1 = gptr[g + 1];
    const int n_chunks = (n1 - n0 + 63) / 64;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r_block = blk_row0[blockIdx.x];
    for (int t = threadIdx.x; t < kRowsPerBlock; t += kThreads) {
        const int i = min(r_block + t, n1 - 1);
        rx[t] = (double)pos[3 * i]; ry[t] = (double)pos[3 * i + 1]; rz[t] = (double)pos[3 * i + 2];
        rbp[t] = bp[i];
    }
    __syncthreads();
    const int r_first = r_block + wv * kRowsPerWave;
    const int n_rows = max(0, min(kRowsPerWave, n1 - r_first));
    int k_inter = 0, k_intra = 0;      // lane t (< kRowsPerWave) keeps the counts of row t
    const double far = r_inter.hi > r_intra.hi ? r_inter.hi : r_intra.hi;
    // masks of row i: [2][n_chunks] words at mask_off[g] * 2 + (i - n0) * 2 * n_chunks
    unsigned long long* mrow = masks + 2 * mask_off[g] + (size_t)(r_first - n0) * 2 * n_chunks;
    for (int c = 0; c < n_chunks; ++c) {
        const int j = n0 + 64 * c + lane;
        const bool jok = j < n1;
        const int jj = jok ? j : n1 - 1;
        const double xj = (double)pos[3 * jj], yj = (double)pos[3 * jj + 1], zj = (double)pos[3 * jj + 2];
        const int bpj = bp[jj];
        for (int t = 0; t < n_rows; ++t) {
            const int li = wv * kRowsPerWave + t;
            const double d0 = rx[li] - xj, d1 = ry[li] - yj, d2 = rz[li] - zj;
            double s = __dmul_rn(d0, d0);
            s = __dadd_rn(s, __dmul_rn(d1, d1));
            s = __dadd_rn(s, __dmul_rn(d2, d2));
            bool e_inter = false, e_intra = false;
            // pair_filter 1: only pairs that touch a ligand atom (bp == 0)
            const bool wanted = pair_filter == 0 || rbp[li] == 0 || bpj == 0;
            if (jok && wanted && s <= far && above(s, r_zero)) {       // 1e-7 < d, d < radius
                e_inter = rbp[li] != bpj && below(s, r_inter);
                e_intra = below(s, r_intra);
            }
            const unsigned long long m_inter = __ballot(e_inter), m_intra = __ballot(e_intra);
            if (lane == 0) {
                mrow[(size_t)t * 2 * n_chunks + c] = m_inter;
                mrow[(size_t)t * 2 * n_chunks + n_chunks + c] = m_intra;
            }
            if (lane == t) { k_inter += __popcll(m_inter); k_intra += __popcll(m_intra); }
        }
    }
    if (lane < n_rows) {
        cnt_inter[r_first + lane] = k_inter;
        cnt_intra[r_first + lane] = k_intra;
    }
}

// Expansion pass: one wave per row turns its masks into the CSR entries (all stores of a row go to
// one contiguous segment).
__global__ void __launch_bounds__(kThreads)
k_radius_fill(const uint8_t* __restrict__ bp, const int32_t* __restrict__ gptr, int n_graphs,
              const long long* __restrict__ mask_off, const unsigned long long* __restrict__ masks,
              const int32_t* __restrict__ rowptr, const int32_t* __restrict__ inter_ptr,
              const int32_t* __restrict__ intra_ptr, int32_t* __restrict__ row, int32_t* __restrict__ col,
              uint8_t* __restrict__ etype, int32_t* __restrict__ perm, float* __restrict__ inv_deg,
              int n_nodes) {
    const int lane = threadIdx.x & 63;
    const int i = (blockIdx.x * kThreads + threadIdx.x) >> 6;
    if (i >= n_nodes) return;
    // graph of the row: binary search over graph_ptr
    int lo = 0, hi = n_graphs;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (gptr[mid] <= i) lo = mid; else hi = mid;
    }
    const int g = lo, n0 = gptr[g], n1 = gptr[g + 1];
    const int n_chunks = (n1 - n0 + 63) / 64;
    const unsigned long long* mrow = masks + 2 * mask_off[g] + (size_t)(i - n0) * 2 * n_chunks;
    const int n_inter_total = inter_ptr[n_nodes];
    const int bpi = bp[i];
    const int seg0 = rowptr[i];
    const int n_int = inter_ptr[i + 1] - inter_ptr[i], n_itr = intra_ptr[i + 1] - intra_ptr[i];
    if (lane == 0) {
        const int deg = n_int + n_itr;
        inv_deg[i] = 1.0f / (float)(deg > 1 ? deg : 1);
    }
    for (int kind = 0; kind < 2; ++kind) {
        const int base = kind == 0 ? seg0 : seg0 + n_int;
        const int pbase = kind == 0 ? inter_ptr[i] : n_inter_total + intra_ptr[i];
        int done = 0;      // entries of this kind already written (wave-uniform)
        for (int c0 = 0; c0 < n_chunks; c0 += 64) {
            const int c = c0 + lane;
            unsigned long long m = c < n_chunks ? mrow[(size_t)kind * n_chunks + c] : 0ull;
            const int cnt = __popcll(m);
            int scan = cnt;
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(scan, o, 64);
                if (lane >= o) scan += t;
            }
            int k = done + scan - cnt;
            while (m) {
                const int bit = __builtin_ctzll(m);
                m &= m - 1ull;
                const int j = n0 + 64 * c + bit;
                const int p = base + k;
                row[p] = i;
                col[p] = j;
                etype[p] = kind == 0 ? 1 : ((bpi == 1 && bp[j] == 1) ? 2 : 0);
                perm[p] = pbase + k;
                ++k;
            }
            done += __shfl(scan, 63, 64);
        }
    }
}

__global__ void k_sum_counts(const int32_t* __restrict__ a, const int32_t* __restrict__ b, int N,
                             int32_t* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) out[i] = a[i] + b[i];
    if (i == N) out[N] = 0;
}

// One sweep of min-label propagation over the CSR (connected components by repeated sweeps until
// *changed stays 0; labels start as the node ids). Used by the host-side `prune` of generate_edges
// (preprocessing.py:139-151: keep the component of the first inter edge's row).
__global__ void k_min_label_step(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, int N,
                                 int32_t* labels, int32_t* changed) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    int m = labels[i];
    const int own = m;
    for (int p = rowptr[i]; p < rowptr[i + 1]; ++p) m = min(m, labels[col[p]]);
    m = min(m, labels[m]);     // pointer jump
    if (m < own) {
        labels[i] = m;
        *changed = 1;
    }
}

struct RgState {
    int32_t *blk_graph, *blk_row0, *n_blocks, *cnt_inter, *cnt_intra, *deg;
    long long* mask_off;
    unsigned long long* masks;
    void* scan_tmp;
    size_t scan_bytes;
};

int max_blocks(int N, int B) { return N / kRowsPerBlock + B + 1; }

size_t mask_words(int N, int max_graph_nodes) {      // both kinds
    return 2 * (size_t)N * (size_t)((max_graph_nodes + 63) / 64);
}

size_t carve_state(PvsArena& a, int N, int B, int max_graph_nodes, RgState* w) {
    RgState t;
    const int mb = max_blocks(N, B);
    t.blk_graph = a.take<int32_t>(mb);
    t.blk_row0 = a.take<int32_t>(mb);
    t.n_blocks = a.take<int32_t>(4);
    t.cnt_inter = a.take<int32_t>((size_t)N + 1);
    t.cnt_intra = a.take<int32_t>((size_t)N + 1);
    t.deg = a.take<int32_t>((size_t)N + 1);
    t.mask_off = a.take<long long>((size_t)B + 1);
    t.masks = a.take<unsigned long long>(mask_words(N, max_graph_nodes));
    size_t sb = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, sb, (const int32_t*)nullptr, (int32_t*)nullptr, N + 1);
    t.scan_bytes = sb;
    t.scan_tmp = a.take<char>(sb);
    if (w) *w = t;
    return a.off;
}

}  // namespace

// state: scratch that lives from _count to _fill (block table, per-row counts, neighbour bit masks:
// 2 * N * ceil(max_graph_nodes / 64) words). max_graph_nodes = size of the largest graph of the batch.
extern "C" size_t pvs_radius_graph_state_bytes(int32_t N, int32_t n_graphs, int32_t max_graph_nodes) {
    PvsArena a(nullptr, 0);
    return carve_state(a, N, n_graphs, max_graph_nodes, nullptr) + 256;
}

extern "C" size_t pvs_radius_graph_workspace_bytes(int32_t N, int32_t n_graphs, int32_t n_edges) {
    (void)n_graphs;
    return pvs_build_csc_workspace_bytes(N, n_edges) + 256;
}

// Step 1: neighbour masks, per-row edge counts and their prefix sums. After it rowptr[N] (device)
// = E: the caller reads that one int32 back to size the arrays of step 2.
extern "C" int pvs_radius_graph_count(const float* pos, const uint8_t* bp, const int32_t* graph_ptr,
                                      int32_t n_graphs, int32_t N, int32_t max_graph_nodes,
                                      double inter_radius, double intra_radius, int32_t pair_filter,
                                      int32_t* rowptr, int32_t* inter_ptr, int32_t* intra_ptr,
                                      void* state, size_t state_bytes, pvs_stream_t stream_) {
    hipStream_t s = (hipStream_t)stream_;
    PVS_REQUIRE(pos && bp && graph_ptr && rowptr && inter_ptr && intra_ptr && state, "pvs_radius_graph_count: NULL");
    PVS_REQUIRE(N > 0 && n_graphs > 0 && max_graph_nodes > 0, "pvs_radius_graph_count: bad sizes N=%d B=%d", N, n_graphs);
    PvsArena arena(state, state_bytes);
    RgState w;
    carve_state(arena, N, n_graphs, max_graph_nodes, &w);
    PVS_REQUIRE(arena.ok(), "pvs_radius_graph_count: state too small (%zu < %zu)", state_bytes, arena.off);
    PvsProfScope prof(s, PVS_PROF_PREPARE);
    const int mb = max_blocks(N, n_graphs);
    k_block_table<<<1, 64, 0, s>>>(graph_ptr, n_graphs, w.blk_graph, w.blk_row0, w.n_blocks, w.mask_off);
    PVS_CHECK_LAUNCH();
    k_radius_masks<<<mb, kThreads, 0, s>>>(pos, bp, graph_ptr, w.blk_graph, w.blk_row0, w.n_blocks, w.mask_off,
                                           make_radius(inter_radius), make_radius(intra_radius),
                                           make_radius(1e-7), pair_filter, w.masks, w.cnt_inter, w.cnt_intra);
    PVS_CHECK_LAUNCH();
    PVS_CHECK_HIP(hipMemsetAsync(w.cnt_inter + N, 0, sizeof(int32_t), s));
    PVS_CHECK_HIP(hipMemsetAsync(w.cnt_intra + N, 0, sizeof(int32_t), s));
    k_sum_counts<<<(N + 1 + 255) / 256, 256, 0, s>>>(w.cnt_inter, w.cnt_intra, N, w.deg);
    PVS_CHECK_LAUNCH();
    size_t sb = w.scan_bytes;
    PVS_CHECK_HIP(hipcub::DeviceScan::ExclusiveSum(w.scan_tmp, sb, w.cnt_inter, inter_ptr, N + 1, s));
    sb = w.scan_bytes;
    PVS_CHECK_HIP(hipcub::DeviceScan::ExclusiveSum(w.scan_tmp, sb, w.cnt_intra, intra_ptr, N + 1, s));
    sb = w.scan_bytes;
    PVS_CHECK_HIP(hipcub::DeviceScan::ExclusiveSum(w.scan_tmp, sb, w.deg, rowptr, N + 1, s));
    return 0;
}

// Step 2: the arrays of PvsGraph (n_edges = rowptr[N] as read back by the caller); `state` as left
// by _count for the same inputs.
extern "C" int pvs_radius_graph_fill(const uint8_t* bp, const int32_t* graph_ptr, int32_t n_graphs, int32_t N,
                                     int32_t max_graph_nodes, int32_t E, const int32_t* rowptr,
                                     const int32_t* inter_ptr, const int32_t* intra_ptr,
                                     int32_t* row, int32_t* col, uint8_t* etype, int32_t* perm,
                                     int32_t* colptr, int32_t* cedge, float* inv_deg,
                                     const void* state, size_t state_bytes,
                                     void* workspace, size_t workspace_bytes, pvs_stream_t stream_) {
    hipStream_t s = (hipStream_t)stream_;
    PVS_REQUIRE(bp && graph_ptr && rowptr && inter_ptr && intra_ptr && row && col && etype && perm && inv_deg &&
                state, "pvs_radius_graph_fill: NULL");
    PVS_REQUIRE((colptr == nullptr) == (cedge == nullptr), "pvs_radius_graph_fill: colptr and cedge go together");
    PVS_REQUIRE(N > 0 && n_graphs > 0 && E >= 0, "pvs_radius_graph_fill: bad sizes");
    PvsArena arena(const_cast<void*>(state), state_bytes);
    RgState w;
    carve_state(arena, N, n_graphs, max_graph_nodes, &w);
    PVS_REQUIRE(arena.ok(), "pvs_radius_graph_fill: state too small");
    PvsProfScope prof(s, PVS_PROF_PREPARE);
    k_radius_fill<<<(N + kWaves - 1) / kWaves, kThreads, 0, s>>>(bp, graph_ptr, n_graphs, w.mask_off, w.masks,
                                                                rowptr, inter_ptr, intra_ptr, row, col, etype,
                                                                perm, inv_deg, N);
    PVS_CHECK_LAUNCH();
    if (!cedge) return 0;     // forward-only use: the by-column lists are only read by the backward
    return pvs_build_csc(s, col, E, N, colptr, cedge, workspace, workspace_bytes);
}

// labels [N] int32 (initialised to 0..N-1 by the caller), changed: device int32 set to 1 when any
// label dropped in this sweep.
extern "C" int pvs_graph_min_label_step(const int32_t* rowptr, const int32_t* col, int32_t N, int32_t* labels,
                                        int32_t* changed, pvs_stream_t stream) {
    PVS_REQUIRE(rowptr && col && labels && changed && N > 0, "pvs_graph_min_label_step: bad arguments");
    k_min_label_step<<<(N + 255) / 256, 256, 0, (hipStream_t)stream>>>(rowptr, col, N, labels, changed);
    PVS_CHECK_LAUNCH();
    return 0;
}
