// pvs_screen_graph_build: the radius graphs of B rigid poses of ONE ligand against ONE receptor
// (virtual screening) without testing the receptor-receptor pairs again: they
// are the same for every pose and come in as a template CSR (built once with pvs_radius_graph_*).
// Node layout per pose: the n_lig ligand atoms first, then the n_rec receptor atoms (so that inside
// a row "ligand columns then receptor columns" IS ascending column order). Same edges, classes and
// in-row order as generate_edges (/root/reference/point_vs/preprocessing/preprocessing.py:68-155):
//   inter block (class 1): ligand<->receptor pairs with 1e-7 < d < inter_radius
//   intra block: all pairs with 1e-7 < d < intra_radius: ligand-ligand and ligand-receptor (class 0),
//                receptor-receptor (class 2, from the template)
// Only the B * n_lig * n_rec ligand-receptor and B * n_lig^2 ligand-ligand distances are evaluated
// (fp64, cdist operation order). Two CSRs come out, both with the edge count left on the device
// (their rowptr[N]; PvsGraph.n_edges_dev): the full graph, and its ligand-touching edges only (what
// the first layer runs over when the receptor-receptor sums are cached, pvs_egnn_layer_fwd_partial).
#include "common.h"
#include "profile.h"
#include "radius_common.h"
#include <hipcub/hipcub.hpp>

namespace {

typedef unsigned long long u64;

// wave per (pose, ligand atom): contact masks against the receptor (64 atoms per word) and the ligand
__global__ void __launch_bounds__(256)
k_contacts(const float* __restrict__ lig_pos, const float* __restrict__ rec_pos, int B, int n_lig, int n_rec,
           Radius r_inter, Radius r_intra, Radius r_zero, u64* __restrict__ m_inter, u64* __restrict__ m_intra,
           u64* __restrict__ m_ll) {
    const int lane = threadIdx.x & 63;
    const int w = (blockIdx.x * 256 + threadIdx.x) >> 6;
    if (w >= B * n_lig) return;
    const int p = w / n_lig, a = w - p * n_lig;
    const int n_chunks = (n_rec + 63) / 64;
    const float* la = lig_pos + ((size_t)p * n_lig + a) * 3;
    const double xa = la[0], ya = la[1], za = la[2];
    for (int c = 0; c < n_chunks; ++c) {
        const int i = 64 * c + lane;
        bool ei = false, ea = false;
        if (i < n_rec) {
            const double s = pvs_sqdist(xa, ya, za, (double)rec_pos[3 * i], (double)rec_pos[3 * i + 1],
                                        (double)rec_pos[3 * i + 2]);
            if (above(s, r_zero)) {
                ei = below(s, r_inter);
                ea = below(s, r_intra);
            }
        }
        const u64 bi = __ballot(ei), ba = __ballot(ea);
        if (lane == 0) {
            m_inter[(size_t)w * n_chunks + c] = bi;
            m_intra[(size_t)w * n_chunks + c] = ba;
        }
    }
    bool ell = false;
    if (lane < n_lig) {
        const float* lb = lig_pos + ((size_t)p * n_lig + lane) * 3;
        const double s = pvs_sqdist(xa, ya, za, (double)lb[0], (double)lb[1], (double)lb[2]);
        ell = above(s, r_zero) && below(s, r_intra);
    }
    const u64 bl = __ballot(ell);
    if (lane == 0) m_ll[w] = bl;
}

__device__ __forceinline__ int row_popc(const u64* __restrict__ m, int n_chunks) {
    int c = 0;
    for (int k = 0; k < n_chunks; ++k) c += __popcll(m[k]);
    return c;
}

// thread per node: degree in the full graph and in the ligand-touching subgraph
__global__ void k_degrees(const u64* __restrict__ m_inter, const u64* __restrict__ m_intra,
                          const u64* __restrict__ m_ll, const int32_t* __restrict__ rr_rowptr, int B, int n_lig,
                          int n_rec, int32_t* __restrict__ deg, int32_t* __restrict__ deg_l) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = n_lig + n_rec, N = B * n;
    if (g > N) return;
    if (g == N) { deg[g] = 0; deg_l[g] = 0; return; }
    const int p = g / n, local = g - p * n;
    const int n_chunks = (n_rec + 63) / 64;
    if (local < n_lig) {
        const size_t w = (size_t)p * n_lig + local;
        const int d = row_popc(m_inter + w * n_chunks, n_chunks) + __popcll(m_ll[w]) +
                      row_popc(m_intra + w * n_chunks, n_chunks);
        deg[g] = d;
        deg_l[g] = d;
    } else {
        const int i = local - n_lig, c = i >> 6;
        const u64 bit = 1ull << (i & 63);
        int nl = 0;
        for (int a = 0; a < n_lig; ++a) {
            const size_t w = (size_t)p * n_lig + a;
            nl += (m_inter[w * n_chunks + c] & bit) ? 1 : 0;
            nl += (m_intra[w * n_chunks + c] & bit) ? 1 : 0;
        }
        deg_l[g] = nl;
        deg[g] = nl + (rr_rowptr[i + 1] - rr_rowptr[i]);
    }
}

struct OutCsr {
    const int32_t* rowptr;
    int32_t *row, *col;
    uint8_t* etype;
    int capacity;
};

// wave per row: writes the row's segment of the full CSR and of the ligand-touching CSR
__global__ void __launch_bounds__(256)
k_fill(const u64* __restrict__ m_inter, const u64* __restrict__ m_intra, const u64* __restrict__ m_ll,
       const int32_t* __restrict__ rr_rowptr, const int32_t* __restrict__ rr_col, int B, int n_lig, int n_rec,
       OutCsr full, OutCsr lig, float* __restrict__ inv_deg, int32_t* __restrict__ status) {
    const int lane = threadIdx.x & 63;
    const int g = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int n = n_lig + n_rec, N = B * n;
    if (g >= N) return;
    if (full.rowptr[N] > full.capacity || lig.rowptr[N] > lig.capacity) {
        if (g == 0 && lane == 0) atomicOr(status, 4);
        return;
    }
    const int p = g / n, local = g - p * n, node0 = p * n;
    const int n_chunks = (n_rec + 63) / 64;
    const u64 lower = (1ull << lane) - 1ull;
    int pf = full.rowptr[g], pl = lig.rowptr[g];
    if (lane == 0) {
        const int d = full.rowptr[g + 1] - pf;
        inv_deg[g] = 1.0f / (float)(d > 1 ? d : 1);
    }
    auto emit = [&](int off, int column, int cls, bool also_lig) {
        full.row[pf + off] = g; full.col[pf + off] = column; full.etype[pf + off] = (uint8_t)cls;
        if (also_lig) { lig.row[pl + off] = g; lig.col[pl + off] = column; lig.etype[pl + off] = (uint8_t)cls; }
    };
    // one mask row (n_chunks words) expanded by the wave: lane c takes word c, offsets by a wave scan
    auto expand_words = [&](const u64* __restrict__ m, int col0, int cls) {
        int done = 0;
        for (int c0 = 0; c0 < n_chunks; c0 += 64) {
            const int c = c0 + lane;
            u64 word = c < n_chunks ? m[c] : 0ull;
            const int cnt = __popcll(word);
            int scan = cnt;
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(scan, o, 64);
                if (lane >= o) scan += t;
            }
            int k = done + scan - cnt;
            while (word) {
                const int bit = __builtin_ctzll(word);
                word &= word - 1ull;
                emit(k, col0 + 64 * c + bit, cls, true);
                ++k;
            }
            done += __shfl(scan, 63, 64);
        }
        pf += done;
        pl += done;
    };
    if (local < n_lig) {
        const size_t w = (size_t)p * n_lig + local;
        expand_words(m_inter + w * n_chunks, node0 + n_lig, 1);             // inter block: receptor atoms
        {                                                                     // intra block: ligand atoms ...
            const u64 mll = m_ll[w];
            const bool on = (mll >> lane) & 1ull;
            if (on) emit(__popcll(mll & lower), node0 + lane, 0, true);
            const int c = __popcll(mll);
            pf += c; pl += c;
        }
        expand_words(m_intra + w * n_chunks, node0 + n_lig, 0);              // ... then receptor atoms
    } else {
        const int i = local - n_lig, c = i >> 6;
        const u64 bit = 1ull << (i & 63);
        for (int kind = 0; kind < 2; ++kind) {                                // inter block, then intra: ligand atoms
            const u64* m = kind == 0 ? m_inter : m_intra;
            const bool on = lane < n_lig && (m[((size_t)p * n_lig + lane) * n_chunks + c] & bit);
            const u64 b = __ballot(on);
            if (on) emit(__popcll(b & lower), node0 + lane, kind == 0 ? 1 : 0, true);
            const int cnt = __popcll(b);
            pf += cnt; pl += cnt;
        }
        const int r0 = rr_rowptr[i], r1 = rr_rowptr[i + 1];                   // intra block: receptor atoms
        for (int k = lane; k < r1 - r0; k += 64) emit(k, node0 + n_lig + rr_col[r0 + k], 2, false);
    }
}

struct ScreenState {
    u64 *m_inter, *m_intra, *m_ll;
    int32_t *deg, *deg_l;
    void* scan_tmp;
    size_t scan_bytes;
};

size_t carve_screen(PvsArena& a, int B, int n_lig, int n_rec, ScreenState* out) {
    ScreenState t;
    const size_t n_chunks = (size_t)(n_rec + 63) / 64, rows = (size_t)B * n_lig;
    const int N = B * (n_lig + n_rec);
    t.m_inter = a.take<u64>(rows * n_chunks);
    t.m_intra = a.take<u64>(rows * n_chunks);
    t.m_ll = a.take<u64>(rows);
    t.deg = a.take<int32_t>((size_t)N + 1);
    t.deg_l = a.take<int32_t>((size_t)N + 1);
    size_t sb = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, sb, (const int32_t*)nullptr, (int32_t*)nullptr, N + 1);
    t.scan_bytes = sb;
    t.scan_tmp = a.take<char>(sb);
    if (out) *out = t;
    return a.off;
}

}  // namespace

extern "C" size_t pvs_screen_graph_state_bytes(int32_t B, int32_t n_lig, int32_t n_rec) {
    PvsArena a(nullptr, 0);
    return carve_screen(a, B, n_lig, n_rec, nullptr) + 256;
}

extern "C" int pvs_screen_graph_build(const float* lig_pos, const float* rec_pos, const int32_t* rr_rowptr,
                                      const int32_t* rr_col, int32_t B, int32_t n_lig, int32_t n_rec,
                                      double inter_radius, double intra_radius, int32_t capacity,
                                      int32_t capacity_lig, int32_t* rowptr, int32_t* row, int32_t* col,
                                      uint8_t* etype, float* inv_deg, int32_t* rowptr_lig, int32_t* row_lig,
                                      int32_t* col_lig, uint8_t* etype_lig, int32_t* status, void* state,
                                      size_t state_bytes, pvs_stream_t stream_) {
    hipStream_t s = (hipStream_t)stream_;
    PVS_REQUIRE(lig_pos && rec_pos && rr_rowptr && rr_col && rowptr && row && col && etype && inv_deg &&
                rowptr_lig && row_lig && col_lig && etype_lig && status && state, "pvs_screen_graph_build: NULL");
    PVS_REQUIRE(B > 0 && n_lig > 0 && n_lig <= 64 && n_rec > 0, "pvs_screen_graph_build: needs 1..64 ligand atoms "
                "(got %d) and a receptor", n_lig);
    PvsArena arena(state, state_bytes);
    ScreenState w;
    carve_screen(arena, B, n_lig, n_rec, &w);
    PVS_REQUIRE(arena.ok(), "pvs_screen_graph_build: state too small (%zu < %zu)", state_bytes, arena.off);
    PvsProfScope prof(s, PVS_PROF_PREPARE);
    const int N = B * (n_lig + n_rec);
    PVS_CHECK_HIP(hipMemsetAsync(status, 0, sizeof(int32_t), s));
    k_contacts<<<(B * n_lig + 3) / 4, 256, 0, s>>>(lig_pos, rec_pos, B, n_lig, n_rec, make_radius(inter_radius),
                                                   make_radius(intra_radius), make_radius(1e-7), w.m_inter,
                                                   w.m_intra, w.m_ll);
    PVS_CHECK_LAUNCH();
    k_degrees<<<(N + 1 + 255) / 256, 256, 0, s>>>(w.m_inter, w.m_intra, w.m_ll, rr_rowptr, B, n_lig, n_rec, w.deg,
                                                  w.deg_l);
    PVS_CHECK_LAUNCH();
    size_t sb = w.scan_bytes;
    PVS_CHECK_HIP(hipcub::DeviceScan::ExclusiveSum(w.scan_tmp, sb, w.deg, rowptr, N + 1, s));
    sb = w.scan_bytes;
    PVS_CHECK_HIP(hipcub::DeviceScan::ExclusiveSum(w.scan_tmp, sb, w.deg_l, rowptr_lig, N + 1, s));
    OutCsr full{rowptr, row, col, etype, capacity}, lig{rowptr_lig, row_lig, col_lig, etype_lig, capacity_lig};
    k_fill<<<(N + 3) / 4, 256, 0, s>>>(w.m_inter, w.m_intra, w.m_ll, rr_rowptr, rr_col, B, n_lig, n_rec, full, lig,
                                       inv_deg, status);
    PVS_CHECK_LAUNCH();
    return 0;
}
