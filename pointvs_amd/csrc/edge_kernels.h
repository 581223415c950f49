// Per-edge kernels of the EGNN layer (internal interface between layer_api.hip and the kernels).
#pragma once
#include "common.h"

// Pointers the edge kernels need, by value in the kernel arguments.
struct PvsEdgeW {
    const float* w1;     // edge_mlp.0.weight [H, ld1]
    int ld1;             // (perm_inv ? H : 2H) + 1 + A
    int off_rho;         // column of the radial feature
    const float* w2;     // [H,H]
    const float* b2;     // [H]
    const float* wc1;    // [H,H]
    const float* bc1;    // [H]
    const float* wc2;    // [H]
    const float* wa;     // [H]  (edge attention) or NULL
    const float* ba;     // [1]
    const float* edge_gate;  // [1] or NULL
    int n_attr;          // A
};

struct PvsEdgeFwdIO {
    const float* PQ;      // [N,2H]: P = W1a h + b1 (row part), Q = W1b h (col part)
    const float* x;       // [N,3]
    const float* m_prev;  // [E,H] sorted or NULL
    float* Magg;          // [N,H]  sum_e att_e m_e over the row segment
    float* x_out;         // [N,3]
    float* m_out;         // [E,H] sorted or NULL
    float* att_out;       // [E] (sigmoid-type: activation; softmax: logits, normalised later)
    float* smax;          // [N] softmax running max  (softmax only)
    float* ssum;          // [N] softmax denominator  (softmax only)
    float* m_scratch;     // [E,H] (H = 128 without m_out: the two launches of the edge forward hand the messages over) or NULL
    bool init_done = false;   // Magg = 0 / x_out = x (or 0) already written by the caller (folded into the P/Q product)
};

struct PvsEdgeBwdIO {
    const float* PQ;
    const float* x;
    const float* m_prev;
    const float* att;      // [E] final attention values (edge attention)
    const float* gM;       // [N,H]  grad wrt Magg
    const float* gxagg;    // [N,3]  grad wrt the coordinate mean = g_x_out * inv_deg, or NULL
    const float* softD;    // [N]    Magg_i . gM_i (softmax attention) or NULL
    const float* g_m_out;  // [E,H] sorted or NULL
    float* gPQ;            // [N,2H]: row part written here (col part by the col gather)
    float* gz1;            // [E,H] sorted: grad wrt the first edge-MLP pre-activation
    float* gd;             // [E,4] sorted: grad wrt (x_row - x_col), and rho in .w
    float* gx_row;         // [N,3] row-side coordinate gradient
    float* g_m_prev;       // [E,H] sorted or NULL
    float* slabs;          // [blocks][slab_floats] per-block weight-gradient partials
    float* wpair;          // [2][H][H] scratch (H = 128: coord_mlp.0's weight and its transpose, staged per launch) or NULL
};

// weight-gradient slab layout (floats), shared by the kernels and the finaliser
#define PVS_MAX_EDGE_ATTR 8
struct PvsSlabLayout {
    int w2, wc1, b2, bc1, wc2, wa, wrho, wattr, ba, gate, total;
};
__host__ __device__ static inline PvsSlabLayout pvs_slab_layout(int H) {
    PvsSlabLayout L;
    int o = 0;
    L.w2 = o; o += H * H;
    L.wc1 = o; o += H * H;
    L.b2 = o; o += H;
    L.bc1 = o; o += H;
    L.wc2 = o; o += H;
    L.wa = o; o += H;
    L.wrho = o; o += H;
    L.wattr = o; o += PVS_MAX_EDGE_ATTR * H;
    L.ba = o; o += 1;
    L.gate = o; o += 1;
    L.total = (o + 3) / 4 * 4;
    return L;
}

int pvs_edge_mfma_supported(int H, uint32_t flags);
int pvs_launch_edge_fwd_mfma(hipStream_t s, int H, const PvsGraph& g, const PvsEdgeW& w, uint32_t flags,
                             int att_act, const PvsEdgeFwdIO& io);
int pvs_edge_bwd_mfma_supported(int H, uint32_t flags, int n_attr);
int pvs_edge_bwd_mfma_max_blocks(int H);
int pvs_launch_edge_bwd_mfma(hipStream_t s, int H, const PvsGraph& g, const PvsEdgeW& w, uint32_t flags,
                             int att_act, const PvsEdgeBwdIO& io, int e_lo, int e_hi, int* n_slabs);
// H = 32 backward with every product as three fp16 terms (f16x2 split with tile scales), both weight-gradient
// operands through transposing LDS reads (edge_bwd_f16.hip, round 3); same contract
int pvs_launch_edge_bwd_f16(hipStream_t s, int H, const PvsGraph& g, const PvsEdgeW& w, uint32_t flags, int att_act,
                            const PvsEdgeBwdIO& io, int e_lo, int e_hi, int* n_slabs);
// H = 128 backward (hidden sizes 65..128): a team of four waves per tile on the f16x2 arithmetic (edge_bwd_wide.hip);
// same contract, io.wpair = pvs_edge_bwd_wide_scratch_floats() floats of scratch
size_t pvs_edge_bwd_wide_scratch_floats();
int pvs_launch_edge_bwd_wide(hipStream_t s, int H, const PvsGraph& g, const PvsEdgeW& w, uint32_t flags, int att_act,
                             const PvsEdgeBwdIO& io, int e_lo, int e_hi, int* n_slabs);
// H = 64 backward, one wave per 16-edge tile with 16x16x32 chain products (edge_bwd_h64.hip); same contract
int pvs_launch_edge_bwd_h64(hipStream_t s, const PvsGraph& g, const PvsEdgeW& w, uint32_t flags, int att_act,
                            const PvsEdgeBwdIO& io, int e_lo, int e_hi, int* n_slabs);
int pvs_edge_v0_supported(int H);
int pvs_edge_v0_blocks(int N);
// softmax attention: att[e] (logit) -> exp(att[e] - smax[row]) / ssum[row] once the rows are complete
int pvs_launch_softmax_finalize(hipStream_t s, const PvsGraph& g, const float* smax, const float* ssum,
                                float* att);
int pvs_launch_edge_fwd_v0(hipStream_t s, int H, const PvsGraph& g, const PvsEdgeW& w, uint32_t flags,
                           int att_act, const PvsEdgeFwdIO& io);
int pvs_launch_edge_bwd_v0(hipStream_t s, int H, const PvsGraph& g, const PvsEdgeW& w, uint32_t flags,
                           int att_act, const PvsEdgeBwdIO& io, int* n_slabs);
// Column-side gather (see k_node_gather); with wsums also the g_wrho/g_wattr partial slabs
// ([n_slabs][4H]: wrho, wattr0..2) that the MFMA backward leaves to this kernel.
int pvs_node_gather_blocks(int N);
int pvs_launch_node_gather(hipStream_t s, int H, const PvsGraph& g, bool wsums, const float* gz1,
                           const float* gd4, const float* gx_row, const float* g_x_out, float* gPQ,
                           float* g_x, float* slabs, int n_lo, int n_hi, int* n_slabs);
