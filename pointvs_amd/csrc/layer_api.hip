// C ABI of one EGNNLayer forward / backward: sequences the kernels on the caller's stream.
// Reference: EGNNLayer.forward, /root/reference/point_vs/models/geometric/egnn_satorras.py:189-206.
#include "common.h"
#include "dense_ops.h"
#include "edge_kernels.h"
#include "node_ops.h"
#include "profile.h"

#include <stdlib.h>

static constexpr uint32_t kFwdRawXsumFlag = 1u << 23;   // == kFwdRawXsum of edge_mfma_common.h

// PVS_EGNN_KERNELS=generic forces the generic (VALU/LDS) edge kernels everywhere: used by the
// tests to cross-check the MFMA path against the generic one on the same inputs.
static bool pvs_use_mfma() {
    const char* v = getenv("PVS_EGNN_KERNELS");
    return !(v && v[0] == 'g');
}

// PVS_ABLATE=<hex bits>: timing-only switches of the MFMA edge kernels (tools/ablate.py)
static uint32_t pvs_ablate_bits() {
    const char* v = getenv("PVS_ABLATE");
    return v ? (uint32_t)strtoul(v, nullptr, 16) << 24 : 0u;
}

namespace {

static size_t edge_slab_capacity(int H, int E) {
    (void)E;
    const size_t cap = (size_t)pvs_edge_bwd_mfma_max_blocks(H);
    return cap < 512 ? 512 : cap;     // one slab per workgroup of the edge backward
}

struct Dims {
    int N, E, H, A, ld1, off_rho, off_q;
    bool perm;
};

Dims make_dims(const PvsLayerDesc* d, const PvsGraph* g) {
    Dims m;
    m.N = g->n_nodes; m.E = g->n_edges; m.H = d->hidden; m.A = d->n_edge_attr;
    m.perm = d->flags & PVS_PERM_INVARIANT;
    m.off_rho = m.perm ? m.H : 2 * m.H;
    m.off_q = m.perm ? 0 : m.H;
    m.ld1 = m.off_rho + 1 + m.A;
    return m;
}

PvsEdgeW make_edge_w(const Dims& m, const PvsLayerParams* p) {
    PvsEdgeW w;
    w.w1 = p->edge_w1; w.ld1 = m.ld1; w.off_rho = m.off_rho;
    w.w2 = p->edge_w2; w.b2 = p->edge_b2;
    w.wc1 = p->coord_w1; w.bc1 = p->coord_b1; w.wc2 = p->coord_w2;
    w.wa = p->att_w; w.ba = p->att_b; w.edge_gate = p->edge_gate; w.n_attr = m.A;
    return w;
}

PvsNodeW make_node_w(const PvsLayerDesc* d, const PvsLayerParams* p) {
    PvsNodeW w;
    const bool gn = d->flags & PVS_GRAPHNORM, na = d->flags & PVS_NODE_ATTENTION;
    w.gn_w = gn ? p->gn_weight : nullptr;
    w.gn_b = gn ? p->gn_bias : nullptr;
    w.gn_ms = gn ? p->gn_mean_scale : nullptr;
    w.natt_w = na ? p->node_att_w : nullptr;
    w.natt_b = na ? p->node_att_b : nullptr;
    w.node_gate = p->node_gate;
    return w;
}

int check_desc(const PvsLayerDesc* d, const PvsGraph* g, const PvsLayerParams* p, bool allow_dev_count = false) {
    PVS_REQUIRE(d && g && p, "NULL descriptor/graph/params");
    // 16 / 32 / 64: every kernel family; 128 (the wide layer: 64 < hidden <= 128 zero-padded by the caller): the MFMA
    // edge kernels only (up to 3 edge classes, no PVS_EGNN_KERNELS=generic)
    PVS_REQUIRE(pvs_edge_v0_supported(d->hidden) ||
                    (d->hidden == 128 && pvs_use_mfma() && pvs_edge_bwd_mfma_supported(128, d->flags, d->n_edge_attr)),
                "hidden size %d unsupported by this build (16, 32, 64; 128 on the MFMA kernels with <= 3 edge classes)",
                d->hidden);
    PVS_REQUIRE(d->n_edge_attr >= 0 && d->n_edge_attr <= PVS_MAX_EDGE_ATTR,
                "n_edge_attr %d unsupported (0..%d)", d->n_edge_attr, PVS_MAX_EDGE_ATTR);
    PVS_REQUIRE(!((d->flags & PVS_GATED_RESIDUAL) && (d->flags & PVS_REZERO)),
                "gated_residual and rezero are incompatible");
    PVS_REQUIRE(g->n_nodes > 0 && g->n_edges >= 0, "bad graph sizes");
    PVS_REQUIRE(allow_dev_count || !g->n_edges_dev, "a graph with a device-side edge count is only accepted by "
                "pvs_egnn_layer_edge_sums / pvs_egnn_layer_fwd_partial");
    PVS_REQUIRE(d->n_edge_attr == 0 || g->etype, "graph has no edge types but layer expects %d",
                d->n_edge_attr);
    PVS_REQUIRE(p->edge_w1 && p->edge_b1 && p->edge_w2 && p->edge_b2 && p->node_w1 && p->node_b1 &&
                p->node_w2 && p->node_b2, "missing edge/node MLP parameters");
    if (d->flags & PVS_UPDATE_COORDS)
        PVS_REQUIRE(p->coord_w1 && p->coord_b1 && p->coord_w2, "missing coord MLP parameters");
    if (d->flags & PVS_EDGE_ATTENTION) PVS_REQUIRE(p->att_w && p->att_b, "missing att_mlp");
    if (d->flags & PVS_NODE_ATTENTION)
        PVS_REQUIRE(p->node_att_w && p->node_att_b, "missing node_att_mlp");
    if (d->flags & PVS_GRAPHNORM)
        PVS_REQUIRE(p->gn_weight && p->gn_bias && p->gn_mean_scale, "missing graphnorm parameters");
    if ((d->flags & PVS_RESIDUAL) && (d->flags & (PVS_REZERO | PVS_GATED_RESIDUAL)))
        PVS_REQUIRE(p->node_gate, "missing node_gate_parameter");
    return 0;
}

struct FwdWs {
    float *PQ, *y1, *u, *o, *smax, *ssum, *shift, *slabs, *m_scratch;
};

size_t carve_fwd(PvsArena& a, const Dims& m, FwdWs* w) {
    const size_t NH = (size_t)m.N * m.H;
    FwdWs t;
    t.PQ = a.take<float>(2 * NH);
    t.y1 = a.take<float>(NH);
    t.u = a.take<float>(NH);
    t.o = a.take<float>(NH);
    t.smax = a.take<float>(m.N);
    t.ssum = a.take<float>(m.N);
    t.shift = a.take<float>(m.H);
    t.slabs = a.take<float>((size_t)pvs_colreduce_blocks(m.N) * m.H);
    // H = 128: the edge forward is two launches that hand the messages over through memory
    t.m_scratch = a.take<float>(m.H > 64 ? (size_t)(m.E > 0 ? m.E : 1) * m.H : 4);
    if (w) *w = t;
    return a.off;
}

struct BwdWs {
    float *PQ, *y1, *u, *o, *g_o, *g_u, *gM, *t1, *tg, *gl, *gxagg, *softD, *gPQ, *gz1, *gd, *gx_row;
    float *eslabs, *gsum, *dslabs, *S1, *S2, *coefs, *gvec, *nslabs, *nsum, *wslabs, *wpair;
};

size_t carve_bwd(PvsArena& a, const Dims& m, BwdWs* w) {
    const size_t NH = (size_t)m.N * m.H;
    const PvsSlabLayout L = pvs_slab_layout(m.H);
    BwdWs t;
    t.PQ = a.take<float>(2 * NH);
    t.y1 = a.take<float>(NH);
    t.u = a.take<float>(NH);
    t.o = a.take<float>(NH);
    t.g_o = a.take<float>(NH);
    t.g_u = a.take<float>(NH);
    t.gM = a.take<float>(NH);
    t.t1 = a.take<float>(NH);
    t.tg = a.take<float>(NH);
    t.gl = a.take<float>(m.N);
    t.gxagg = a.take<float>(3 * (size_t)m.N);
    t.softD = a.take<float>(m.N);
    t.gPQ = a.take<float>(2 * NH);
    t.gz1 = a.take<float>((size_t)(m.E > 0 ? m.E : 1) * m.H);
    t.gd = a.take<float>(4 * (size_t)(m.E > 0 ? m.E : 1));
    t.gx_row = a.take<float>(3 * (size_t)m.N);
    t.eslabs = a.take<float>(edge_slab_capacity(m.H, m.E) * L.total);
    t.gsum = a.take<float>(L.total);
    t.dslabs = a.take<float>((size_t)pvs_reduce_blocks(m.N) * m.H * m.H);
    t.S1 = a.take<float>(m.H);
    t.S2 = a.take<float>(m.H);
    t.coefs = a.take<float>(4 * (size_t)m.H);
    t.gvec = a.take<float>(m.H);
    t.nslabs = a.take<float>((size_t)2048 * 4 * m.H);      // one [4H] slab per workgroup of the column gather
    t.nsum = a.take<float>(4 * (size_t)m.H);
    t.wslabs = a.take<float>(pvs_node_wgrads_supported(m.H) ? pvs_node_wgrads_slab_floats(m.N, m.H) : 4);
    // H = 128: coord_mlp.0's weight staged per launch for the team backward (split A operands, or the fp32 pair)
    t.wpair = a.take<float>(m.H > 64 ? pvs_edge_bwd_wide_scratch_floats() + 2 * (size_t)m.H * m.H : 4);
    if (w) *w = t;
    return a.off;
}

// scatter the reduced edge-kernel slab into the parameter gradients
// The first `node_blocks` workgroups (if any) reduce the node-level weight-gradient slabs, 32 entries each, straight
// into the gradient tensors (no reduced copy in between, no launch of its own); the others scatter the edge sums.
__global__ void __launch_bounds__(256)
k_finalize_edge_grads(const float* __restrict__ gsum, PvsSlabLayout L, int H, int A,
                      int ld1, int off_rho, PvsLayerGrads gr, int has_coord,
                      int has_att, int has_gate, const float* __restrict__ node_gsum,
                      PvsNodeWgradOut node_out, PvsNodeWgradSlabs node_slabs, int node_blocks) {
    if ((int)blockIdx.x < node_blocks) {
        __shared__ float part[8][33];
        pvs_node_wgrads_reduce_scatter32(node_slabs, node_out, H, 32 * (int)blockIdx.x, part);
        return;
    }
    const int tid = ((int)blockIdx.x - node_blocks) * blockDim.x + threadIdx.x;
    const int stride = ((int)gridDim.x - node_blocks) * blockDim.x;
    // (the node-level weight gradients' reduced sums, when they were left for this kernel to scatter)
    if (node_gsum) pvs_node_wgrads_scatter(node_gsum, node_out, H, tid, stride);
    for (int i = tid; i < H * H; i += stride) {
        if (gr.edge_w2) gr.edge_w2[i] = gsum[L.w2 + i];
        if (has_coord && gr.coord_w1) gr.coord_w1[i] = gsum[L.wc1 + i];
    }
    for (int c = tid; c < H; c += stride) {
        if (gr.edge_b2) gr.edge_b2[c] = gsum[L.b2 + c];
        if (has_coord && gr.coord_b1) gr.coord_b1[c] = gsum[L.bc1 + c];
        if (has_coord && gr.coord_w2) gr.coord_w2[c] = gsum[L.wc2 + c];
        if (has_att && gr.att_w) gr.att_w[c] = gsum[L.wa + c];
        if (gr.edge_w1) {
            gr.edge_w1[(size_t)c * ld1 + off_rho] = gsum[L.wrho + c];
            for (int t = 0; t < A; ++t)
                gr.edge_w1[(size_t)c * ld1 + off_rho + 1 + t] = gsum[L.wattr + t * H + c];
        }
    }
    if (tid == 0) {
        // (softmax attention, has_att == 2: a softmax does not see a shift of its logits, so the gradient of the logit
        // bias is zero IDENTICALLY; the sum of the per-edge logit gradients only leaves its rounding residue there)
        if (has_att && gr.att_b) gr.att_b[0] = has_att == 2 ? 0.f : gsum[L.ba];
        if (has_gate && gr.edge_gate) gr.edge_gate[0] = gsum[L.gate];
    }
}

#define PVS_TRY(call)             \
    do {                          \
        int _rc = (call);         \
        if (_rc) return _rc;      \
    } while (0)

// y1 = [h | Magg] Wn1^T + bn1 ; (graphnorm stats) ; u = SiLU(GN(y1)) ; o = u Wn2^T + bn2 ; h_out
// = node_out(o, h). Without GraphNorm the SiLU rides on the first product's epilogue, and the plain
// (un-gated, no node attention) output stage on the second's: two launches instead of four.
int node_mlp_forward(hipStream_t s, const Dims& m, const PvsLayerDesc* d, const PvsLayerParams* p,
                     const PvsNodeW& nw, const float* h, const float* Magg, float* y1, float* u,
                     float* o, float* stats, bool compute_stats, float* shift_tmp, float* slabs,
                     float* h_out, float* node_att_out) {
    const int H = m.H;
    const uint32_t F = d->flags;
    const bool can_epi = pvs_linear_epilogue_supported(H, H, H, H, H, H, y1, h, Magg) &&
                         pvs_linear_epilogue_supported(H, H, 0, H, 0, H, o, u, nullptr) &&
                         (((uintptr_t)h_out | (uintptr_t)u) & 15) == 0;
    const bool fuse_silu = can_epi && !(F & PVS_GRAPHNORM);
    const bool gated = (F & PVS_RESIDUAL) && (F & (PVS_REZERO | PVS_GATED_RESIDUAL));
    const bool fuse_out = can_epi && !(F & PVS_NODE_ATTENTION) && !gated;
    // (The A/B switches are read from the environment at EVERY call on purpose - ADVICE r03 suggested statics: the
    // test suite and tools/abenv.py flip them inside one process, e.g. the exact-fp32 family against the split products
    // on the same tensors. A getenv is ~0.2 us of host time; a layer call enqueues 6-9 launches of >= 3 us each.)
    const bool split_small = getenv("PVS_EGNN_SPLIT_SMALL") != nullptr;     // (the launches apart, for A/B)
    if (fuse_silu && !gated && !split_small && p->node_b1 && p->node_b2 &&
        pvs_node_mlp_fused_supported(H, h, Magg, y1, h_out)) {
        // no GraphNorm, no rezero / gated residual: the whole chain y1 -> u -> o -> (node gate) -> h_out in one launch
        // (dense_ops.hip: k_node_mlp_fwd)
        const bool natt = F & PVS_NODE_ATTENTION;
        return pvs_launch_node_mlp_fwd(s, H, m.N, h, Magg, p->node_w1, p->node_b1, p->node_w2, p->node_b2,
                                       (F & PVS_RESIDUAL) != 0, natt ? nw.natt_w : nullptr, natt ? nw.natt_b : nullptr,
                                       d->att_act, y1, u, o, h_out, node_att_out);
    }
    PVS_TRY(pvs_launch_linear(s, y1, H, h, H, p->node_w1, 2 * H, 1, p->node_b1, Magg, H,
                              p->node_w1 + H, 2 * H, 1, m.N, H, H, H, false,
                              fuse_silu ? PVS_EPI_SILU_OUT : PVS_EPI_NONE, nullptr, 0, u, H));
    if ((F & PVS_GRAPHNORM) && compute_stats)
        PVS_TRY(pvs_graphnorm_stats(s, y1, p->gn_mean_scale, m.N, H, stats, shift_tmp, slabs));
    if (!fuse_silu) PVS_TRY(pvs_node_tail_fwd(s, y1, stats, nw, m.N, H, u));
    PVS_TRY(pvs_launch_linear(s, o, H, u, H, p->node_w2, H, 1, p->node_b2, nullptr, 0, nullptr, 0, 0,
                              m.N, H, 0, H, false,
                              !fuse_out ? PVS_EPI_NONE : (F & PVS_RESIDUAL) ? PVS_EPI_ADD_OUT : PVS_EPI_COPY_OUT,
                              h, H, h_out, H));
    if (!fuse_out) PVS_TRY(pvs_node_out_fwd(s, H, o, h, nw, F, d->att_act, m.N, h_out, node_att_out));
    return 0;
}

// P and Q of a layer as ONE launch where the MFMA linear takes the shape (H = 32, 64): two groups of workgroups, one per
// 32- or 64-column half of the output, the second on the Q slice of edge_mlp.0's weight.
// `init` (MFMA edge forward only): the edge kernel never flushes rows without edges, so Magg = 0 and x_out = x (0 for
// raw sums) are written first - by a third group of workgroups of that launch, else by the edge launcher's own small
// kernel (io->init_done stays false).
int node_pre_forward(hipStream_t s, const Dims& m, const PvsLayerParams* p, const float* h,
                     float* PQ, PvsEdgeFwdIO* init = nullptr, uint32_t init_flags = 0) {
    const int H = m.H;
    const bool split_small = getenv("PVS_EGNN_SPLIT_SMALL") != nullptr;     // (the launches apart, for A/B)
    PvsLinearExt e;
    if (init) {
        e.zero_rows = init->Magg; e.zero_w = H; e.zero_ld = H;
        if (init_flags & kFwdRawXsumFlag) e.zero3 = init->x_out;
        else if (init_flags & PVS_UPDATE_COORDS) { e.copy3_src = init->x; e.copy3_dst = init->x_out; }
    }
    const bool side_ok = init && !split_small && ((uintptr_t)init->Magg & 15) == 0;
    if (H == 32 && !split_small && pvs_linear_epilogue_supported(2 * H, H, 0, H, 0, 2 * H, PQ, h, nullptr) &&
        (!init || side_ok)) {
        // P | Q in one launch: 64 outputs, the second column block (group) on the Q slice of edge_mlp.0's weight, no bias
        e.w_shift1 = (long long)m.off_q - 32LL * m.ld1;
        e.bias_blocks = 1;
        // (as two groups of one-column-block workgroups, the clears as a third group: 12 us against 15-16 for
        // workgroups twice as long that also do the clears)
        e.groups = 2; e.side_group = init ? 1 : 0;
        PVS_TRY(pvs_launch_linear(s, PQ, 2 * H, h, H, p->edge_w1, m.ld1, 1, p->edge_b1, nullptr, 0, nullptr, 0, 0, m.N,
                                  H, 0, 2 * H, false, PVS_EPI_NONE, nullptr, 0, nullptr, 0, &e));
        if (init) init->init_done = true;
        return 0;
    }
    if (H == 64 && !split_small && pvs_linear_epilogue_supported(2 * H, H, 0, H, 0, H, PQ, h, nullptr)) {
        // H = 64: P and Q as two groups of workgroups of ONE launch (64 outputs each; the second group on the Q slice of
        // the weight, no bias), the forward's clears as a third group (as side jobs of the P workgroups they cost 11 us
        // against 6 for a launch of their own, profiles/r03_ab_small_launch_folding.txt).
        PvsLinearExt g2 = side_ok ? e : PvsLinearExt{};      // (with the forward's clears as a third group of their own)
        g2.groups = 2; g2.shift_block = 2; g2.bias_blocks = 2; g2.side_group = side_ok ? 1 : 0;
        g2.w_shift1 = (long long)m.off_q - 64LL * m.ld1;
        PVS_TRY(pvs_launch_linear(s, PQ, 2 * H, h, H, p->edge_w1, m.ld1, 1, p->edge_b1, nullptr, 0, nullptr, 0, 0, m.N, H, 0,
                                  2 * H, false, PVS_EPI_NONE, nullptr, 0, nullptr, 0, &g2));
        if (side_ok) init->init_done = true;
        return 0;
    }
    // P = W1[:, 0:H] h + b1 (row part), Q = W1[:, off_q:off_q+H] h (col part)
    PVS_TRY(pvs_launch_linear(s, PQ, 2 * H, h, H, p->edge_w1, m.ld1, 1, p->edge_b1, nullptr, 0,
                              nullptr, 0, 0, m.N, H, 0, H, false));
    PVS_TRY(pvs_launch_linear(s, PQ + H, 2 * H, h, H, p->edge_w1 + m.off_q, m.ld1, 1, nullptr,
                              nullptr, 0, nullptr, 0, 0, m.N, H, 0, H, false));
    return 0;
}

}  // namespace

extern "C" size_t pvs_egnn_layer_saved_floats(const PvsLayerDesc* d, int32_t N, int32_t E) {
    (void)E;
    // Magg [N,H] | graphnorm stats [2H] | PQ [N,2H] | y1 [N,H] | o [N,H] | u [N,H]: the node-level forward is
    // kept for the backward (N rows: small) instead of being recomputed
    // (+ u = SiLU(GN(y1)) [N,H]: one small launch less in the backward)
    return 6 * (size_t)N * d->hidden + 2 * (size_t)d->hidden;
}

extern "C" size_t pvs_egnn_layer_workspace_bytes(const PvsLayerDesc* d, int32_t N, int32_t E,
                                                 int32_t backward) {
    PvsGraph g{};
    g.n_nodes = N; g.n_edges = E;
    Dims m = make_dims(d, &g);
    PvsArena a(nullptr, 0);
    if (backward == 2)   // pvs_egnn_layer_edge_sums / _fwd_partial: forward + a per-edge attention scratch
        return carve_fwd(a, m, nullptr) + pvs_align_up((size_t)(E > 0 ? E : 1) * sizeof(float), 256) + 512;
    return (backward ? carve_bwd(a, m, nullptr) : carve_fwd(a, m, nullptr)) + 256;
}

extern "C" int pvs_egnn_layer_fwd(const PvsLayerDesc* d, const PvsGraph* g, const PvsLayerParams* p,
                                  const float* h, const float* x, const float* m_prev, float* h_out,
                                  float* x_out, float* m_out, float* att_out, float* node_att_out,
                                  float* saved, void* workspace, size_t workspace_bytes,
                                  pvs_stream_t stream_) {
    hipStream_t s = (hipStream_t)stream_;
    PVS_TRY(check_desc(d, g, p, /*allow_dev_count=*/true));
    PVS_REQUIRE(!g->n_edges_dev || (pvs_use_mfma() && pvs_edge_mfma_supported(d->hidden, d->flags) && !m_out),
                "pvs_egnn_layer_fwd: a graph with a device-side edge count needs the MFMA edge kernel and m_out = NULL");
    PVS_REQUIRE(h && x && h_out && x_out && saved, "pvs_egnn_layer_fwd: NULL tensor");
    PVS_REQUIRE(x_out != x, "pvs_egnn_layer_fwd: x_out must not alias x");
    const bool eatt = d->flags & PVS_EDGE_ATTENTION;
    PVS_REQUIRE(!eatt || att_out, "pvs_egnn_layer_fwd: att_out required with edge attention");
    const Dims m = make_dims(d, g);
    PvsArena arena(workspace, workspace_bytes);
    FwdWs w;
    carve_fwd(arena, m, &w);
    PVS_REQUIRE(arena.ok(), "pvs_egnn_layer_fwd: workspace too small (%zu < %zu)", workspace_bytes,
                arena.off);
    const int H = m.H;
    float* Magg = saved;
    float* stats = saved + (size_t)m.N * H;
    float* sPQ = stats + 2 * H;
    float* sy1 = sPQ + 2 * (size_t)m.N * H;
    float* so = sy1 + (size_t)m.N * H;
    const PvsEdgeW ew = make_edge_w(m, p);
    const PvsNodeW nw = make_node_w(d, p);

    PvsEdgeFwdIO io;
    io.PQ = sPQ; io.x = x; io.m_prev = m_prev; io.Magg = Magg; io.x_out = x_out; io.m_out = m_out;
    io.att_out = att_out; io.smax = w.smax; io.ssum = w.ssum; io.m_scratch = w.m_scratch;
    const bool mfma_fwd = pvs_use_mfma() && pvs_edge_mfma_supported(H, d->flags);
    PVS_TRY(node_pre_forward(s, m, p, h, sPQ, mfma_fwd ? &io : nullptr, d->flags));
    if (mfma_fwd)
        PVS_TRY(pvs_launch_edge_fwd_mfma(s, H, *g, ew, d->flags | pvs_ablate_bits(), d->att_act, io));
    else
        PVS_TRY(pvs_launch_edge_fwd_v0(s, H, *g, ew, d->flags, d->att_act, io));
    if (!(d->flags & PVS_UPDATE_COORDS))
        PVS_CHECK_HIP(hipMemcpyAsync(x_out, x, sizeof(float) * 3 * (size_t)m.N,
                                     hipMemcpyDeviceToDevice, s));
    PVS_TRY(node_mlp_forward(s, m, d, p, nw, h, Magg, sy1, so + (size_t)m.N * H, so, stats, true, w.shift, w.slabs,
                             h_out, node_att_out));
    return 0;
}

// ---- forward with part of every row's edges given as precomputed sums (screening: the
// receptor-receptor messages of the first layer do not depend on the ligand pose) ----
namespace {

__global__ void k_combine_partial(float* __restrict__ Magg, float* __restrict__ x_out, const float* __restrict__ x,
                                  const float* __restrict__ base_magg, const float* __restrict__ base_xsum,
                                  const float* __restrict__ base_deg, const int32_t* __restrict__ rowptr,
                                  int N, int H, int upd) {
    const int qpr = H / 4;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = t / qpr, q = t - n * qpr;
    if (n >= N) return;
    float4* mp = reinterpret_cast<float4*>(Magg + (size_t)n * H + 4 * q);
    const float4 b = *reinterpret_cast<const float4*>(base_magg + (size_t)n * H + 4 * q);
    float4 m = *mp;
    m.x += b.x; m.y += b.y; m.z += b.z; m.w += b.w;
    *mp = m;
    if (q == 0) {
        if (upd) {
            const float deg = base_deg[n] + (float)(rowptr[n + 1] - rowptr[n]);
            const float inv = 1.0f / (deg > 1.f ? deg : 1.f);
#pragma unroll
            for (int c = 0; c < 3; ++c)
                x_out[3 * n + c] = x[3 * n + c] + (base_xsum[3 * n + c] + x_out[3 * n + c]) * inv;
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) x_out[3 * n + c] = x[3 * n + c];
        }
    }
}

int partial_checks(const PvsLayerDesc* d, const PvsGraph* g, const PvsLayerParams* p) {
    PVS_TRY(check_desc(d, g, p, true));
    PVS_REQUIRE(pvs_use_mfma() && pvs_edge_mfma_supported(d->hidden, d->flags),
                "partial-sum forward needs the MFMA edge kernel (H = 32 or 64)");
    PVS_REQUIRE(!((d->flags & PVS_EDGE_ATTENTION) && (d->flags & PVS_SOFTMAX_ATT)),
                "partial-sum forward: softmax attention is not supported (row sums of two edge sets do not add)");
    PVS_REQUIRE(!(d->flags & PVS_EDGE_RESIDUAL), "partial-sum forward: edge_residual layers are not supported");
    return 0;
}

}  // namespace

extern "C" int pvs_egnn_layer_edge_sums(const PvsLayerDesc* d, const PvsGraph* g, const PvsLayerParams* p,
                                        const float* h, const float* x, float* magg, float* xsum,
                                        void* workspace, size_t workspace_bytes, pvs_stream_t stream_) {
    hipStream_t s = (hipStream_t)stream_;
    PVS_TRY(partial_checks(d, g, p));
    PVS_REQUIRE(h && x && magg && xsum, "pvs_egnn_layer_edge_sums: NULL tensor");
    const Dims m = make_dims(d, g);
    PvsArena arena(workspace, workspace_bytes);
    FwdWs w;
    carve_fwd(arena, m, &w);
    float* att = arena.take<float>((size_t)(m.E > 0 ? m.E : 1));
    PVS_REQUIRE(arena.ok(), "pvs_egnn_layer_edge_sums: workspace too small (%zu < %zu)", workspace_bytes, arena.off);
    const PvsEdgeW ew = make_edge_w(m, p);
    PvsEdgeFwdIO io;
    io.PQ = w.PQ; io.x = x; io.m_prev = nullptr; io.Magg = magg; io.x_out = xsum; io.m_out = nullptr;
    io.att_out = att; io.smax = w.smax; io.ssum = w.ssum; io.m_scratch = w.m_scratch;
    PVS_TRY(node_pre_forward(s, m, p, h, w.PQ, &io, d->flags | kFwdRawXsumFlag));
    PVS_TRY(pvs_launch_edge_fwd_mfma(s, m.H, *g, ew, d->flags | kFwdRawXsumFlag, d->att_act, io));
    if (!(d->flags & PVS_UPDATE_COORDS))
        PVS_CHECK_HIP(hipMemsetAsync(xsum, 0, sizeof(float) * 3 * (size_t)m.N, s));
    return 0;
}

extern "C" int pvs_egnn_layer_fwd_partial(const PvsLayerDesc* d, const PvsGraph* g, const PvsLayerParams* p,
                                          const float* h, const float* x, const float* base_magg,
                                          const float* base_xsum, const float* base_deg, float* h_out,
                                          float* x_out, float* node_att_out, float* saved, void* workspace,
                                          size_t workspace_bytes, pvs_stream_t stream_) {
    hipStream_t s = (hipStream_t)stream_;
    PVS_TRY(partial_checks(d, g, p));
    PVS_REQUIRE(h && x && base_magg && base_xsum && base_deg && h_out && x_out && saved,
                "pvs_egnn_layer_fwd_partial: NULL tensor");
    PVS_REQUIRE(x_out != x, "pvs_egnn_layer_fwd_partial: x_out must not alias x");
    const Dims m = make_dims(d, g);
    PvsArena arena(workspace, workspace_bytes);
    FwdWs w;
    carve_fwd(arena, m, &w);
    float* att = arena.take<float>((size_t)(m.E > 0 ? m.E : 1));
    PVS_REQUIRE(arena.ok(), "pvs_egnn_layer_fwd_partial: workspace too small (%zu < %zu)", workspace_bytes, arena.off);
    const int H = m.H;
    float* Magg = saved;
    float* stats = saved + (size_t)m.N * H;
    float* sPQ = stats + 2 * H;
    float* sy1 = sPQ + 2 * (size_t)m.N * H;
    float* so = sy1 + (size_t)m.N * H;
    const PvsEdgeW ew = make_edge_w(m, p);
    const PvsNodeW nw = make_node_w(d, p);
    PvsEdgeFwdIO io;
    io.PQ = sPQ; io.x = x; io.m_prev = nullptr; io.Magg = Magg; io.x_out = x_out; io.m_out = nullptr;
    io.att_out = att; io.smax = w.smax; io.ssum = w.ssum; io.m_scratch = w.m_scratch;
    PVS_TRY(node_pre_forward(s, m, p, h, sPQ, &io, d->flags | kFwdRawXsumFlag));
    pvs_prof_set_fwd_tag(PVS_PROF_EDGE_FWD_PARTIAL);      // timed apart from the full-graph layers (bench.py)
    const int rc_partial = pvs_launch_edge_fwd_mfma(s, H, *g, ew, d->flags | kFwdRawXsumFlag, d->att_act, io);
    pvs_prof_set_fwd_tag(PVS_PROF_EDGE_FWD);
    PVS_TRY(rc_partial);
    const long long threads = (long long)m.N * (H / 4);
    k_combine_partial<<<(int)((threads + 255) / 256), 256, 0, s>>>(Magg, x_out, x, base_magg, base_xsum, base_deg,
                                                                   g->rowptr, m.N, H,
                                                                   (d->flags & PVS_UPDATE_COORDS) ? 1 : 0);
    PVS_CHECK_LAUNCH();
    PVS_TRY(node_mlp_forward(s, m, d, p, nw, h, Magg, sy1, so + (size_t)m.N * H, so, stats, true, w.shift, w.slabs,
                             h_out, node_att_out));
    return 0;
}

extern "C" int pvs_egnn_layer_bwd(const PvsLayerDesc* d, const PvsGraph* g, const PvsLayerParams* p,
                                  const float* h, const float* x, const float* m_prev,
                                  const float* att, const float* saved, const float* g_h_out,
                                  const float* g_x_out, const float* g_m_out, float* g_h, float* g_x,
                                  float* g_m_prev, const PvsLayerGrads* gr_, void* workspace,
                                  size_t workspace_bytes, pvs_stream_t stream_) {
    hipStream_t s = (hipStream_t)stream_;
    PVS_TRY(check_desc(d, g, p));
    PVS_REQUIRE(h && x && saved && g_h_out && g_h && gr_, "pvs_egnn_layer_bwd: NULL tensor");
    PVS_REQUIRE(g->n_edges == 0 || (g->colptr && g->cedge),
                "pvs_egnn_layer_bwd: graph was built without the by-column lists (forward-only)");
    const uint32_t F = d->flags;
    const bool eatt = F & PVS_EDGE_ATTENTION, soft = F & PVS_SOFTMAX_ATT;
    const bool eres = (F & PVS_EDGE_RESIDUAL) && m_prev;
    PVS_REQUIRE(!eatt || att, "pvs_egnn_layer_bwd: att required with edge attention");
    PVS_REQUIRE(!eres || g_m_prev, "pvs_egnn_layer_bwd: g_m_prev required with edge residual");
    const PvsLayerGrads gr = *gr_;
    const Dims m = make_dims(d, g);
    PvsArena arena(workspace, workspace_bytes);
    BwdWs w;
    carve_bwd(arena, m, &w);
    PVS_REQUIRE(arena.ok(), "pvs_egnn_layer_bwd: workspace too small (%zu < %zu)", workspace_bytes,
                arena.off);
    const int H = m.H, N = m.N;
    const float* Magg = saved;
    float* stats = const_cast<float*>(saved) + (size_t)N * H;   // read-only here
    const float* sPQ = stats + 2 * H;
    const float* sy1 = sPQ + 2 * (size_t)N * H;
    const float* so = sy1 + (size_t)N * H;
    const PvsEdgeW ew = make_edge_w(m, p);
    const PvsNodeW nw = make_node_w(d, p);
    const bool gn = F & PVS_GRAPHNORM, natt = F & PVS_NODE_ATTENTION;
    const bool gates = (F & PVS_RESIDUAL) && (F & (PVS_REZERO | PVS_GATED_RESIDUAL));
    const bool coord_bwd = (F & PVS_UPDATE_COORDS) && g_x_out;
    // all node-level weight gradients in one pass at the end (H = 32, 64, the usual full set of grads)
    const bool fused_wgrads = pvs_node_wgrads_supported(H) && gr.node_w2 && gr.node_w1 && gr.edge_w1 &&
                              !getenv("PVS_EGNN_SPLIT_WGRADS");

    // ---- node-level forward: PQ, y1, o were kept by the forward; u = SiLU(GN(y1)) is elementwise ----
    const float* su = so + (size_t)N * H;      // u = SiLU(GN(y1)) kept by the forward

    // ---- node_model backward ----
    const bool split_small = getenv("PVS_EGNN_SPLIT_SMALL") != nullptr;     // (the launches apart, for A/B)
    const bool mfma_bwd = pvs_use_mfma() && pvs_edge_bwd_mfma_supported(H, F, m.A);
    // Layers without GraphNorm and without rezero / gated residual: the output stage (node gate, residual), g_y1 and
    // [g_h | gM] as ONE launch that also carries the per-node preparation of the edge backward (dense_ops.hip:
    // k_node_mlp_bwd). Without a node gate g_o is g_h_out itself and nothing is written for it.
    const bool chain_bwd = !gn && !gates && !split_small && pvs_node_mlp_fused_supported(H, g_h_out, sy1, w.g_u, g_h) &&
                           (((uintptr_t)w.gM | (uintptr_t)w.gPQ | (uintptr_t)so | (uintptr_t)w.g_o | (uintptr_t)w.t1) & 15) == 0;
    // (no residual, no node gate: g_o = g_h_out and the residual's part of g_h is zero - nothing to launch either way)
    const bool plain_out = !(F & PVS_RESIDUAL) && !natt && !split_small;
    const float* g_o = (plain_out || (chain_bwd && !natt)) ? g_h_out : w.g_o;
    if (!plain_out && !chain_bwd)
        PVS_TRY(pvs_node_out_bwd(s, H, g_h_out, so, h, nw, F, d->att_act, N, w.g_o, g_h, w.gl, w.t1,
                                 w.tg));
    PvsLinearExt prep;
    if (mfma_bwd) { prep.zero_rows = w.gPQ; prep.zero_w = H; prep.zero_ld = 2 * H; prep.zero3 = w.gx_row; }
    if (coord_bwd) { prep.scale3_src = g_x_out; prep.scale3_by = g->inv_deg; prep.scale3_dst = w.gxagg; }
    if (chain_bwd)
        PVS_TRY(pvs_launch_node_mlp_bwd(s, H, N, g_h_out, so, sy1, p->node_w1, p->node_w2, (F & PVS_RESIDUAL) != 0,
                                        natt ? nw.natt_w : nullptr, natt ? nw.natt_b : nullptr, d->att_act, w.g_o, w.t1,
                                        w.gl, w.g_u, g_h, w.gM, &prep));
    // (the node gate's weight gradients - column sums of t1, sum of gl - ride on the fused weight-gradient pass when
    // it runs: two column reductions and their slab reductions less per layer)
    const bool gate_in_wgrads = natt && fused_wgrads && !split_small && gr.node_att_w && gr.node_att_b;
    if (natt && !gate_in_wgrads) {
        if (gr.node_att_w)
            PVS_TRY(pvs_launch_colreduce(s, PVS_COL_SUM_A, gr.node_att_w, w.t1, H, nullptr, 0, nullptr,
                                         N, H, 1.f, w.dslabs, false));
        if (gr.node_att_b)
            PVS_TRY(pvs_launch_colreduce(s, PVS_COL_SUM_A, gr.node_att_b, w.gl, 1, nullptr, 0, nullptr,
                                         N, 1, 1.f, w.dslabs, false));
    }
    if (gates && gr.node_gate) {
        PVS_TRY(pvs_launch_colreduce(s, PVS_COL_SUM_A, w.gvec, w.tg, H, nullptr, 0, nullptr, N, H, 1.f,
                                     w.dslabs, false));
        PVS_TRY(pvs_sum_vec(s, w.gvec, H, gr.node_gate));
    }
    // o = u Wn2^T + bn2
    // (without GraphNorm g_y1 = g_u * SiLU'(y1) rides on this product's epilogue)
    const bool fuse_tail_bwd = !gn && pvs_linear_epilogue_supported(H, H, 0, H, 0, H, w.g_u, g_o, nullptr) &&
                               ((uintptr_t)sy1 & 15) == 0;
    if (!chain_bwd)
        PVS_TRY(pvs_launch_linear(s, w.g_u, H, g_o, H, p->node_w2, 1, H, nullptr, nullptr, 0, nullptr, 0,
                                  0, N, H, 0, H, false, fuse_tail_bwd ? PVS_EPI_MUL_SILU_GRAD : PVS_EPI_NONE, sy1, H,
                                  nullptr, 0));
    if (gr.node_w2 && !fused_wgrads)
        PVS_TRY(pvs_launch_tsgemm_tn(s, gr.node_w2, H, g_o, H, su, H, N, H, H, w.dslabs, false));
    if (gr.node_b2 && !fused_wgrads)
        PVS_TRY(pvs_launch_colreduce(s, PVS_COL_SUM_A, gr.node_b2, g_o, H, nullptr, 0, nullptr, N, H,
                                     1.f, w.dslabs, false));
    // u = SiLU(GN(y1)) ; g_u becomes g_yn then g_y1 in place
    if (!fuse_tail_bwd && !chain_bwd) PVS_TRY(pvs_node_tail_bwd1(s, w.g_u, sy1, stats, nw, N, H, w.g_u));
    if (gn) {
        PVS_TRY(pvs_launch_colreduce(s, PVS_COL_SUM_A, w.S1, w.g_u, H, nullptr, 0, nullptr, N, H, 1.f,
                                     w.dslabs, false));
        PVS_TRY(pvs_launch_colreduce(s, PVS_COL_SUM_AB, w.S2, w.g_u, H, sy1, H, nullptr, N, H, 1.f,
                                     w.dslabs, false));
        PVS_TRY(pvs_graphnorm_bwd_coefs(s, w.S1, w.S2, stats, nw, N, H, gr.gn_weight, gr.gn_bias,
                                        gr.gn_mean_scale, w.coefs));
        PVS_TRY(pvs_node_tail_bwd2(s, w.g_u, sy1, stats, nw, w.coefs, N, H, w.g_u));
    }
    float* g_y1 = w.g_u;
    // y1 = h Wn1[:, :H]^T + Magg Wn1[:, H:]^T + bn1
    // The per-node preparation of the edge backward (clear the row part of gPQ and gx_row - rows without edges are
    // never written by the MFMA edge backward -, g_x_out / deg) rides on these products as side jobs of the node rows
    // where the MFMA linear takes the shape; the softmax row dots need the finished gM and keep their own launch.
    const bool prep_side = !split_small && ((uintptr_t)w.gPQ & 15) == 0 &&
                           pvs_linear_epilogue_supported(H, H, 0, H, 0, H, w.gM, g_y1, nullptr);
    bool prep_done = false;
    if (chain_bwd) {
        prep_done = true;       // (launched above)
    } else if (H == 32 && prep_side && pvs_linear_epilogue_supported(H, H, 0, H, 0, 2 * H, g_h, g_y1, nullptr)) {
        // g_h (+)= and gM = in one launch: 64 outputs over the two halves of node_mlp.0's weight, the second column
        // block to gM
        prep.y1 = w.gM; prep.ldy1 = H; prep.acc1 = 0;
        PVS_TRY(pvs_launch_linear(s, g_h, H, g_y1, H, p->node_w1, 1, 2 * H, nullptr, nullptr, 0, nullptr, 0, 0, N, H, 0,
                                  2 * H, !plain_out, PVS_EPI_NONE, nullptr, 0, nullptr, 0, &prep));
        prep_done = true;
    } else {
        PVS_TRY(pvs_launch_linear(s, g_h, H, g_y1, H, p->node_w1, 1, 2 * H, nullptr, nullptr, 0, nullptr,
                                  0, 0, N, H, 0, H, !plain_out));
        PVS_TRY(pvs_launch_linear(s, w.gM, H, g_y1, H, p->node_w1 + H, 1, 2 * H, nullptr, nullptr, 0,
                                  nullptr, 0, 0, N, H, 0, H, false, PVS_EPI_NONE, nullptr, 0, nullptr, 0,
                                  prep_side ? &prep : nullptr));
        prep_done = prep_side;
    }
    if (gr.node_w1 && !fused_wgrads) {
        PVS_TRY(pvs_launch_tsgemm_tn(s, gr.node_w1, 2 * H, g_y1, H, h, H, N, H, H, w.dslabs, false));
        PVS_TRY(pvs_launch_tsgemm_tn(s, gr.node_w1 + H, 2 * H, g_y1, H, Magg, H, N, H, H, w.dslabs,
                                     false));
    }
    if (gr.node_b1 && !fused_wgrads)
        PVS_TRY(pvs_launch_colreduce(s, PVS_COL_SUM_A, gr.node_b1, g_y1, H, nullptr, 0, nullptr, N, H,
                                     1.f, w.dslabs, false));

    // ---- edge backward ----
    // (MFMA path: rows without edges are never written by the edge kernel: cleared here, unless done above)
    if (prep_done)
        PVS_TRY(pvs_prep_edge_bwd(s, g_x_out, g->inv_deg, Magg, w.gM, N, H, nullptr,
                                  (eatt && soft) ? w.softD : nullptr, nullptr, nullptr));
    else
        PVS_TRY(pvs_prep_edge_bwd(s, g_x_out, g->inv_deg, Magg, w.gM, N, H,
                                  coord_bwd ? w.gxagg : nullptr, (eatt && soft) ? w.softD : nullptr,
                                  mfma_bwd ? w.gPQ : nullptr, mfma_bwd ? w.gx_row : nullptr));
    PvsEdgeBwdIO io;
    io.PQ = sPQ; io.x = x; io.m_prev = m_prev; io.att = att; io.gM = w.gM;
    io.gxagg = coord_bwd ? w.gxagg : nullptr;
    io.softD = (eatt && soft) ? w.softD : nullptr;
    io.g_m_out = g_m_out; io.gPQ = w.gPQ; io.gz1 = w.gz1; io.gd = w.gd; io.gx_row = w.gx_row;
    io.g_m_prev = eres ? g_m_prev : nullptr; io.slabs = w.eslabs; io.wpair = w.wpair;
    int n_slabs = 0;
    const PvsSlabLayout L = pvs_slab_layout(H);
    int n_nslabs = 0;
    if (mfma_bwd) {
        const uint32_t Fk = F | pvs_ablate_bits();
        PVS_TRY(pvs_launch_edge_bwd_mfma(s, H, *g, ew, Fk, d->att_act, io, 0, m.E, &n_slabs));
        PVS_TRY(pvs_launch_node_gather(s, H, *g, true, w.gz1, w.gd, w.gx_row, g_x_out, w.gPQ, g_x,
                                       w.nslabs, 0, N, &n_nslabs));
    } else {
        PVS_TRY(pvs_launch_edge_bwd_v0(s, H, *g, ew, F, d->att_act, io, &n_slabs));
        PVS_TRY(pvs_launch_node_gather(s, H, *g, false, w.gz1, w.gd, w.gx_row, g_x_out, w.gPQ, g_x,
                                       w.nslabs, 0, N, &n_nslabs));
    }
    // g_wrho / g_wattr come from the node gather: its slab layout [wrho | wattr0 | wattr1 | wattr2]
    // == gsum[L.wrho .. L.wrho + 4H), which the edge slabs leave alone (one reduction for both). With the fused
    // weight-gradient pass below that reduction rides on ITS launch as extra workgroups (independent work), and the
    // pass's own slab reduction on the finalize launch: three launches at the end of a layer instead of five.
    PvsReduce2Args edge_red;
    edge_red.out_a = w.gsum; edge_red.slabs_a = w.eslabs; edge_red.n_a = n_slabs; edge_red.width_a = L.total;
    edge_red.skip_lo = L.wrho; edge_red.skip_hi = L.wrho + 4 * H;
    edge_red.out_b = w.gsum + L.wrho; edge_red.slabs_b = w.nslabs; edge_red.n_b = n_nslabs; edge_red.width_b = 4 * H;
    const bool tail_folded = mfma_bwd && fused_wgrads && !split_small;
    if (mfma_bwd && !tail_folded) {
        PVS_TRY(pvs_launch_reduce_slabs2(s, w.gsum, w.eslabs, n_slabs, L.total, L.wrho, L.wrho + 4 * H,
                                         w.gsum + L.wrho, w.nslabs, n_nslabs, 4 * H));
    } else if (!mfma_bwd) {
        PVS_TRY(pvs_launch_reduce_slabs(s, w.gsum, L.total, L.total, w.eslabs, n_slabs, L.total, false));
    }

    // ---- first edge-MLP layer at node level: P = W1a h + b1, Q = W1b h ----
    // g_h += g_P W1a + g_Q W1b: one launch with the two (input, weight) pairs
    // (with the folded tail it is a role of the weight-gradient launch below)
    const bool gh_folded = tail_folded && (((uintptr_t)g_h | (uintptr_t)w.gPQ) & 15) == 0;
    if (!gh_folded)
        PVS_TRY(pvs_launch_linear(s, g_h, H, w.gPQ, 2 * H, p->edge_w1, 1, m.ld1, nullptr, w.gPQ + H, 2 * H,
                                  p->edge_w1 + m.off_q, 1, m.ld1, N, H, H, H, true));
    const float* node_gsum = nullptr;
    PvsNodeWgradSlabs node_slabs;
    PvsNodeWgradOut node_out{};
    if (fused_wgrads) {
        PvsNodeWgradIn wi;
        wi.g_o = g_o; wi.g_y1 = g_y1; wi.gPQ = w.gPQ; wi.u = su; wi.h = h; wi.Magg = Magg;
        PvsNodeWgradOut wo;
        wo.node_w2 = gr.node_w2; wo.node_w1 = gr.node_w1; wo.edge_w1 = gr.edge_w1;
        wo.node_b2 = gr.node_b2; wo.node_b1 = gr.node_b1; wo.edge_b1 = gr.edge_b1;
        wo.ld1 = m.ld1; wo.off_q = m.off_q; wo.perm = m.perm ? 1 : 0;
        if (gate_in_wgrads) { wi.t1 = w.t1; wi.gl = w.gl; wo.natt_w = gr.node_att_w; wo.natt_b = gr.node_att_b; }
        PvsGhJob ghj;
        ghj.g_h = g_h; ghj.gPQ = w.gPQ; ghj.W1 = p->edge_w1; ghj.ld1 = m.ld1; ghj.off_q = m.off_q;
        if (tail_folded)
            PVS_TRY(pvs_launch_node_wgrads(s, H, N, wi, wo, w.wslabs, /*scatter=*/false, nullptr, &edge_red, &node_slabs,
                                           gh_folded ? &ghj : nullptr));
        else
            PVS_TRY(pvs_launch_node_wgrads(s, H, N, wi, wo, w.wslabs, /*scatter=*/false, &node_gsum));
        node_out = wo;
    }
    if (gr.edge_w1 && !fused_wgrads) {
        PVS_TRY(pvs_launch_tsgemm_tn(s, gr.edge_w1, m.ld1, w.gPQ, 2 * H, h, H, N, H, H, w.dslabs,
                                     false));
        PVS_TRY(pvs_launch_tsgemm_tn(s, gr.edge_w1 + m.off_q, m.ld1, w.gPQ + H, 2 * H, h, H, N, H, H,
                                     w.dslabs, m.perm));
    }
    if (gr.edge_b1 && !fused_wgrads)
        PVS_TRY(pvs_launch_colreduce(s, PVS_COL_SUM_A, gr.edge_b1, w.gPQ, 2 * H, nullptr, 0, nullptr,
                                     N, H, 1.f, w.dslabs, false));
    const int node_blocks = node_slabs.slabs ? (node_slabs.width + 31) / 32 : 0;
    k_finalize_edge_grads<<<node_blocks + (H * H / 256 > 4 ? H * H / 256 : 4), 256, 0, s>>>(
        w.gsum, L, H, m.A, m.ld1, m.off_rho, gr, coord_bwd ? 1 : 0, eatt ? (soft ? 2 : 1) : 0,
        (eres && (F & (PVS_REZERO | PVS_GATED_RESIDUAL))) ? 1 : 0, node_gsum, node_out, node_slabs, node_blocks);
    PVS_CHECK_LAUNCH();
    // GraphNorm: node_mlp.0.bias from the closed form of k_gn_bwd_coefs instead of the column sum of g_y1
    if (gn && gr.node_b1) PVS_TRY(pvs_copy_small(s, w.coefs + 3 * H, gr.node_b1, H));
    return 0;
}

// ---- the whole EGNNLayer stack as one call each way (include/pvs_egnn.h: pvs_egnn_stack_*) -------------------------------
// SartorrasEGNN.get_embeddings loops `for layer in self.layers` (egnn_satorras.py:325-328); at the reference's default shape
// (32 graphs of ~500 atoms, edge_radius 4 A) a training step is ~70 launches of a few microseconds and the HOST decides its
// length: one autograd node and one C call per layer cost more than the layer's kernels. These two entry points sequence
// the very same per-layer launches (pvs_egnn_layer_fwd / _bwd above, unchanged: results are bit for bit those of the
// per-layer calls) over caller-owned buffers that hold every layer's tensors at fixed strides.
namespace {

int check_stack(const PvsLayerDesc* descs, const PvsLayerParams* params, int32_t n_layers, const PvsGraph* g,
                const PvsStackStrides* st, const char* who) {
    PVS_REQUIRE(descs && params && g && st, "%s: NULL descriptor / parameter / graph / stride table", who);
    PVS_REQUIRE(n_layers >= 1 && n_layers <= 1024, "%s: n_layers %d out of range", who, n_layers);
    const int H = descs[0].hidden;
    for (int l = 0; l < n_layers; ++l) {
        PVS_REQUIRE(descs[l].hidden == H, "%s: layer %d has hidden size %d, layer 0 has %d (one width per stack)", who, l,
                    descs[l].hidden, H);
        PVS_REQUIRE(!(descs[l].flags & PVS_EDGE_RESIDUAL),
                    "%s: edge_residual layers hand [E,H] messages from layer to layer: use the per-layer calls", who);
    }
    const size_t N = (size_t)g->n_nodes, E = (size_t)(g->n_edges > 0 ? g->n_edges : 1);
    PVS_REQUIRE(st->h_mid >= (int64_t)(N * H) && st->x_mid >= (int64_t)(3 * N) && st->att >= (int64_t)E &&
                    st->node_att >= (int64_t)N &&
                    st->saved >= (int64_t)pvs_egnn_layer_saved_floats(&descs[0], g->n_nodes, g->n_edges),
                "%s: a stride is smaller than the tensor it separates", who);
    PVS_REQUIRE(((st->h_mid | st->x_mid | st->att | st->node_att | st->saved) & 3) == 0,
                "%s: strides must be multiples of 4 floats (16-byte rows)", who);
    return 0;
}

struct StackBwdScratch {
    float* gh[2];
    float* gx[2];
};

size_t carve_stack_bwd(PvsArena& a, int N, int H, StackBwdScratch* s) {
    StackBwdScratch t;
    for (int k = 0; k < 2; ++k) {
        t.gh[k] = a.take<float>((size_t)N * H);
        t.gx[k] = a.take<float>(3 * (size_t)N);
    }
    if (s) *s = t;
    return pvs_align_up(a.off, 256);
}

}  // namespace

extern "C" size_t pvs_egnn_stack_workspace_bytes(const PvsLayerDesc* descs, int32_t n_layers, int32_t N, int32_t E,
                                                 int32_t backward) {
    if (!descs || n_layers < 1) return 0;
    size_t layer = 0;
    for (int l = 0; l < n_layers; ++l) {
        const size_t b = pvs_egnn_layer_workspace_bytes(&descs[l], N, E, backward ? 1 : 0);
        if (b > layer) layer = b;
    }
    if (!backward) return layer;
    PvsArena a(nullptr, 0);
    return carve_stack_bwd(a, N, descs[0].hidden, nullptr) + layer;
}

extern "C" int pvs_egnn_stack_fwd(const PvsLayerDesc* descs, const PvsLayerParams* params, int32_t n_layers,
                                  const PvsGraph* g, const PvsStackStrides* st, const float* h0, const float* x0,
                                  float* h_mid, float* x_mid, float* h_out, float* x_out, float* att, float* node_att,
                                  float* saved, void* workspace, size_t workspace_bytes, pvs_stream_t stream) {
    PVS_TRY(check_stack(descs, params, n_layers, g, st, "pvs_egnn_stack_fwd"));
    PVS_REQUIRE(h0 && x0 && h_out && x_out && saved && (n_layers == 1 || (h_mid && x_mid)),
                "pvs_egnn_stack_fwd: NULL tensor");
    const float* h = h0;
    const float* x = x0;
    for (int l = 0; l < n_layers; ++l) {
        const bool last = l == n_layers - 1;
        float* ho = last ? h_out : h_mid + (size_t)l * st->h_mid;
        float* xo = last ? x_out : x_mid + (size_t)l * st->x_mid;
        PVS_REQUIRE(!(descs[l].flags & PVS_EDGE_ATTENTION) || att, "pvs_egnn_stack_fwd: att required (layer %d has edge attention)", l);
        PVS_TRY(pvs_egnn_layer_fwd(&descs[l], g, &params[l], h, x, nullptr, ho, xo, nullptr,
                                   att ? att + (size_t)l * st->att : nullptr,
                                   ((descs[l].flags & PVS_NODE_ATTENTION) && node_att) ? node_att + (size_t)l * st->node_att : nullptr,
                                   saved + (size_t)l * st->saved, workspace, workspace_bytes, stream));
        h = ho;
        x = xo;
    }
    return 0;
}

extern "C" int pvs_egnn_stack_bwd(const PvsLayerDesc* descs, const PvsLayerParams* params, int32_t n_layers,
                                  const PvsGraph* g, const PvsStackStrides* st, const float* h0, const float* x0,
                                  const float* h_mid, const float* x_mid, const float* att, const float* saved,
                                  const float* g_h_out, const float* g_x_out, float* g_h0, float* g_x0,
                                  const PvsLayerGrads* grads, void* workspace, size_t workspace_bytes,
                                  pvs_stream_t stream) {
    PVS_TRY(check_stack(descs, params, n_layers, g, st, "pvs_egnn_stack_bwd"));
    PVS_REQUIRE(h0 && x0 && saved && g_h_out && g_h0 && grads && (n_layers == 1 || (h_mid && x_mid)),
                "pvs_egnn_stack_bwd: NULL tensor");
    PvsArena arena(workspace, workspace_bytes);
    StackBwdScratch sc;
    const size_t own = carve_stack_bwd(arena, g->n_nodes, descs[0].hidden, &sc);
    PVS_REQUIRE(own <= workspace_bytes, "pvs_egnn_stack_bwd: workspace too small (%zu < %zu)", workspace_bytes, own);
    void* layer_ws = (char*)workspace + own;
    const size_t layer_ws_bytes = workspace_bytes - own;
    const float* gh = g_h_out;
    const float* gx = g_x_out;           // NULL: nothing reads the last layer's coordinates
    for (int l = n_layers - 1; l >= 0; --l) {
        const float* h = l ? h_mid + (size_t)(l - 1) * st->h_mid : h0;
        const float* x = l ? x_mid + (size_t)(l - 1) * st->x_mid : x0;
        float* gh_in = l ? sc.gh[l & 1] : g_h0;
        float* gx_in = l ? sc.gx[l & 1] : g_x0;          // (layer 0: NULL when the caller's coordinates need no gradient)
        PVS_TRY(pvs_egnn_layer_bwd(&descs[l], g, &params[l], h, x, nullptr, att ? att + (size_t)l * st->att : nullptr,
                                   saved + (size_t)l * st->saved, gh, gx, nullptr, gh_in, gx_in, nullptr, &grads[l],
                                   layer_ws, layer_ws_bytes, stream));
        gh = gh_in;
        gx = gx_in;
    }
    return 0;
}
