/*
 * pvs_egnn.h - C ABI of libpvs_egnn.so: the MI355X (gfx950) implementation of the PointVS EGNN
 * message-passing hot path (forward and backward).
 *
 * The reference has no native code at all: the boundary it offers for this path is
 * the Python nn.Module surface.  Each entry point below therefore names the reference Python
 * function whose body it replaces.  The host side that keeps the reference's class surface
 * (`pointvs_amd/egnn_satorras.py` etc.) binds these with ctypes; see INTEGRATION.md for the stub a
 * reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to caller-owned memory (torch allocations); the library
 *     never allocates, frees or synchronises; all work is enqueued on `stream` (hipGraph-capturable)
 *   - floating point is fp32, indices int32 inside the library (the int64 COO / int64 one-hot the
 *     reference hands over is converted once per batch by pvs_graph_prepare)
 *   - row-major, dense; "[E,H] sorted" means edge rows in the CSR order produced by
 *     pvs_graph_prepare (use pvs_rows_to_input_order to get the reference's input edge order)
 *   - return value: 0 = ok, <0 = error; pvs_last_error() gives the message (thread-local)
 *   - stateless and re-entrant
 */
#ifndef PVS_EGNN_H
#define PVS_EGNN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* pvs_stream_t; /* hipStream_t */

/* ---- layer flags: one bit per EGNNLayer.__init__ switch (egnn_satorras.py:26-46) ---- */
enum {
    PVS_RESIDUAL        = 1u << 0,
    PVS_EDGE_RESIDUAL   = 1u << 1,  /* applied only when m_prev != NULL (egnn_satorras.py:194) */
    PVS_EDGE_ATTENTION  = 1u << 2,
    PVS_NORMALIZE       = 1u << 3,
    PVS_TANH            = 1u << 4,
    PVS_GRAPHNORM       = 1u << 5,
    PVS_UPDATE_COORDS   = 1u << 6,
    PVS_PERM_INVARIANT  = 1u << 7,
    PVS_NODE_ATTENTION  = 1u << 8,
    PVS_GATED_RESIDUAL  = 1u << 9,
    PVS_REZERO          = 1u << 10,
    PVS_SOFTMAX_ATT     = 1u << 11
};

/* attention_activation_fn (egnn_satorras.py:66-71); IDENTITY is what softmax_attention selects */
enum { PVS_ACT_SIGMOID = 0, PVS_ACT_TANH = 1, PVS_ACT_RELU = 2, PVS_ACT_SILU = 3, PVS_ACT_IDENTITY = 4 };

typedef struct PvsLayerDesc {
    int32_t  hidden;       /* H = input_nf = hidden_nf = output_nf (always k,k,k in build_net). Built widths: 16, 32, 64
                            * (every kernel family) and 128 (MFMA kernels, <= 3 edge classes: the reference's --channels
                            * 65..128, parse_args.py:56, zero-padded by the caller); other sizes are padded up by the caller */
    int32_t  n_edge_attr;  /* A = edges_in_d: number of one-hot edge classes (0 = no edge_attr) */
    uint32_t flags;        /* PVS_* bits */
    int32_t  att_act;      /* PVS_ACT_* */
} PvsLayerDesc;

/* Graph in the library's layout, filled by pvs_graph_prepare. Rows = edge_index[0] (aggregation
 * target, egnn_satorras.py:135,171), cols = edge_index[1]. */
typedef struct PvsGraph {
    int32_t n_nodes;
    int32_t n_edges;
    const int32_t* rowptr;   /* [N+1] CSR offsets by row                                        */
    const int32_t* row;      /* [E]   row of each sorted edge                                   */
    const int32_t* col;      /* [E]   col of each sorted edge (stable order within a row)       */
    const uint8_t* etype;    /* [E]   one-hot class of each sorted edge (NULL when A == 0)      */
    const int32_t* perm;     /* [E]   sorted position -> input edge id                          */
    const int32_t* colptr;   /* [N+1] CSC offsets by col                                        */
    const int32_t* cedge;    /* [E]   sorted positions of the edges grouped by col (stable)     */
    const float*   inv_deg;  /* [N]   1 / max(deg_row, 1)  (unsorted_segment_mean's clamp)      */
    /* Optional DEVICE int32: the true edge count when it is only known on the device (a graph made
     * by pvs_graph_filter_ligand_edges); n_edges is then the capacity of the edge arrays. Only
     * pvs_egnn_layer_edge_sums / _fwd_partial accept such a graph. */
    const int32_t* n_edges_dev;
    /* Optional DEVICE int32 [n_graphs + 1]: first sorted edge of every graph of the batch (graphs are
     * contiguous node ranges, so contiguous edge ranges of the CSR), last entry = E; NULL / 0 when unknown.
     * Speed and values of the reference are unaffected by it; the fp16-split edge backward uses it to end its
     * 32-edge tiles at graph boundaries, because one tile shares one power-of-two scale and two graphs'
     * gradients can differ by many orders of magnitude ("f16x2").
     * ACCURACY CONTRACT for callers that leave it NULL: a tile of the H = 32 backward may then span two graphs, and the
     * smaller graph's per-edge gradients in that tile keep 22 - max(0, k - 14) bits when they are 2^-k below the
     * larger graph's (absolute error 2^-36 of the tile's largest value). pointvs_amd fills it wherever it knows the
     * batch's graphs: model forward, get_embeddings (from the batch vector), every graph prepared by
     * pvs_graph_prepare_runs (edge_ptr IS this table). */
    const int32_t* graph_eptr;
    int32_t n_graphs;
} PvsGraph;

/* Parameters of one EGNNLayer, torch nn.Linear layout W[out][in] (state_dict keys in comments). */
typedef struct PvsLayerParams {
    const float* edge_w1;   /* edge_mlp.0.weight  [H, (perm_inv?H:2H)+1+A] */
    const float* edge_b1;   /* edge_mlp.0.bias    [H]      */
    const float* edge_w2;   /* edge_mlp.2.weight  [H,H]    */
    const float* edge_b2;   /* edge_mlp.2.bias    [H]      */
    const float* coord_w1;  /* coord_mlp.0.weight [H,H]    */
    const float* coord_b1;  /* coord_mlp.0.bias   [H]      */
    const float* coord_w2;  /* coord_mlp.2.weight [1,H]    */
    const float* att_w;     /* att_mlp.0.weight   [1,H]    (edge attention) */
    const float* att_b;     /* att_mlp.0.bias     [1]      */
    const float* node_w1;   /* node_mlp.0.weight  [H,2H]   */
    const float* node_b1;   /* node_mlp.0.bias    [H]      */
    const float* node_w2;   /* node_mlp.3.weight  [H,H]    */
    const float* node_b2;   /* node_mlp.3.bias    [H]      */
    const float* gn_weight; /* node_mlp.1.weight  [H]      (graphnorm) */
    const float* gn_bias;   /* node_mlp.1.bias    [H]      */
    const float* gn_mean_scale; /* node_mlp.1.mean_scale [H] */
    const float* node_att_w;    /* node_att_mlp.0.weight [1,H] */
    const float* node_att_b;    /* node_att_mlp.0.bias   [1]   */
    const float* edge_gate;     /* edge_gate_parameter [1] */
    const float* node_gate;     /* node_gate_parameter [1] */
} PvsLayerParams;

/* Gradients, same shapes; every non-NULL member is OVERWRITTEN with this call's gradient. */
typedef struct PvsLayerGrads {
    float *edge_w1, *edge_b1, *edge_w2, *edge_b2;
    float *coord_w1, *coord_b1, *coord_w2;
    float *att_w, *att_b;
    float *node_w1, *node_b1, *node_w2, *node_b2;
    float *gn_weight, *gn_bias, *gn_mean_scale;
    float *node_att_w, *node_att_b;
    float *edge_gate, *node_gate;
} PvsLayerGrads;

const char* pvs_last_error(void);
int pvs_version(void);

/* ---------------------------------------------------------------------------------------------
 * Graph preparation: int64 COO (arbitrary order, duplicates allowed - SURVEY Q6) + int64 one-hot
 * edge_attr, as PygPointCloudDataset emits them (data_loaders.py:359-370), to CSR/CSC.
 * Replaces nothing in the reference (torch indexes the COO directly, egnn_satorras.py:190-193);
 * it is the once-per-batch price of atomics-free, deterministic segment reductions.
 *   edge_index [2,E] int64, edge_attr [E,A] int64 one-hot or NULL.
 *   status: device int32[1], 0 on success, bit0 = index out of range, bit1 = edge_attr row not
 *   one-hot; checked by the caller whenever it next synchronises.
 * Output arrays are the members of PvsGraph (caller-allocated with the sizes given there).
 * colptr / cedge may both be NULL for forward-only use (skips the second sort).
 */
size_t pvs_graph_prepare_workspace_bytes(int32_t n_nodes, int32_t n_edges);
int pvs_graph_prepare(const int64_t* edge_index, const int64_t* edge_attr, int32_t n_edge_attr,
                      int32_t n_nodes, int32_t n_edges,
                      int32_t* rowptr, int32_t* row, int32_t* col, uint8_t* etype, int32_t* perm,
                      int32_t* colptr, int32_t* cedge, float* inv_deg, int32_t* status,
                      void* workspace, size_t workspace_bytes, pvs_stream_t stream);

/* The same result as pvs_graph_prepare for an edge list with the layout of the reference's data loader,
 * without the by-row radix sort: per graph, generate_edges (preprocessing.py:108-142) emits the
 * inter-molecular block and then the intra-molecular block, each in row-major (np.where) order = two
 * row-sorted runs, and PyG collation (data_loaders.py:517-520) concatenates the graphs. node_ptr / edge_ptr
 * [n_graphs + 1] (DEVICE int32) give each graph's node and edge ranges. The layout is the caller's
 * contract; it is verified on the device and a violation sets bit 4 of *status (bits 1, 2 as in
 * pvs_graph_prepare). Outputs are array-for-array those of pvs_graph_prepare.
 * max_graph_nodes: an upper bound on the nodes of one graph, known to the caller (0 = unknown). With a bound of
 * at most 4096 the by-column lists are built by a counting transpose per graph (LDS tables over the graph's own
 * columns) instead of a radix sort of all edges by column; same arrays. A graph larger than the bound sets bit 4. */
size_t pvs_graph_prepare_runs_workspace_bytes(int32_t n_nodes, int32_t n_edges, int32_t n_graphs,
                                              int32_t max_graph_nodes);
int pvs_graph_prepare_runs(const int64_t* edge_index, const int64_t* edge_attr, int32_t n_edge_attr,
                           int32_t n_nodes, int32_t n_edges, int32_t n_graphs, const int32_t* node_ptr,
                           const int32_t* edge_ptr, int32_t* rowptr, int32_t* row, int32_t* col, uint8_t* etype,
                           int32_t* perm, int32_t* colptr, int32_t* cedge, float* inv_deg, int32_t* status,
                           int32_t max_graph_nodes, void* workspace, size_t workspace_bytes, pvs_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Radius graph straight from coordinates: generate_edges of the reference
 * (preprocessing.py:68-155, the prune=False body) fused with the COO -> CSR/CSC step, so the
 * int64 edge list and the int64 one-hot never exist. Same result, array for array, as
 * pvs_graph_prepare on the reference's own output for the same atoms:
 *   inter block: pairs with 1e-7 < d < inter_radius and bp_i != bp_j (class 1), row-major;
 *   intra block: all pairs with 1e-7 < d < intra_radius (class 2 if both bp == 1, else 0), row-major;
 *   perm = position of each CSR-sorted edge in [inter block | intra block] (the reference's order).
 * d is the float64 cdist distance (same operation order, no FMA): identical `<` decisions.
 *   pos [N,3] fp32, bp [N] uint8 (0 ligand / 1 receptor), graph_ptr [B+1] int32 node offsets of the
 *   batch's graphs (pairs are only formed inside a graph).
 * Two steps because E is data dependent: _count runs the O(n_g^2) distance tests once, leaves a
 * 64-bit neighbour mask per (row, 64-column chunk) in `state` and fills rowptr/inter_ptr/intra_ptr
 * ([N+1] each); the caller reads rowptr[N] = E back, allocates, and _fill expands the masks.
 * max_graph_nodes = node count of the largest graph of the batch (sizes the masks).
 * colptr / cedge may both be NULL (forward-only use: only the backward reads the by-column lists;
 * skips the radix sort); pvs_egnn_layer_bwd then refuses the graph.
 */
size_t pvs_radius_graph_state_bytes(int32_t n_nodes, int32_t n_graphs, int32_t max_graph_nodes);
size_t pvs_radius_graph_workspace_bytes(int32_t n_nodes, int32_t n_graphs, int32_t n_edges);
int pvs_radius_graph_count(const float* pos, const uint8_t* bp, const int32_t* graph_ptr, int32_t n_graphs,
                           int32_t n_nodes, int32_t max_graph_nodes, double inter_radius, double intra_radius,
                           int32_t pair_filter /* 0: every pair; 1: only pairs that touch a ligand atom (bp == 0) */,
                           int32_t* rowptr, int32_t* inter_ptr, int32_t* intra_ptr,
                           void* state, size_t state_bytes, pvs_stream_t stream);
int pvs_radius_graph_fill(const uint8_t* bp, const int32_t* graph_ptr, int32_t n_graphs, int32_t n_nodes,
                          int32_t max_graph_nodes, int32_t n_edges,
                          const int32_t* rowptr, const int32_t* inter_ptr, const int32_t* intra_ptr,
                          int32_t* row, int32_t* col, uint8_t* etype, int32_t* perm,
                          int32_t* colptr, int32_t* cedge, float* inv_deg,
                          const void* state, size_t state_bytes,
                          void* workspace, size_t workspace_bytes, pvs_stream_t stream);

/* One sweep of min-label propagation over a CSR (labels start as node ids; repeat until *changed
 * stays 0): connected components for the `prune` option of generate_edges
 * (preprocessing.py:139-151), driven from the host side (pointvs_amd/radius_graph.py). */
int pvs_graph_min_label_step(const int32_t* rowptr, const int32_t* col, int32_t n_nodes, int32_t* labels,
                             int32_t* changed, pvs_stream_t stream);

/* Edge dropout of SartorrasEGNN.get_embeddings (egnn_satorras.py:320-323: torch_geometric's
 * dropout_adj(edges, edge_attr, p, force_undirected=True, training)): of every pair only the copy with
 * row <= col is drawn (kept with probability 1 - p), the survivors' reverses are appended, attributes repeated.
 * Philox4x32-10 keyed on (seed, step), counter = edge id: reproducible, not bit-matched to torch's generator.
 * _mark: pos [E + 1] = exclusive scan of the keep flags (pos[E] = number of survivors K, read it on the host);
 * _fill: out_index int64 [2][2K], out_attr int64 [2K][A]. */
size_t pvs_dropout_adj_workspace_bytes(int32_t n_edges);
int pvs_dropout_adj_mark(const int64_t* edge_index, int32_t n_edges, float p, uint64_t seed, uint64_t step,
                         int32_t* pos, void* workspace, size_t workspace_bytes, pvs_stream_t stream);
int pvs_dropout_adj_fill(const int64_t* edge_index, const int64_t* edge_attr, int32_t n_edge_attr,
                         int32_t n_edges, const int32_t* pos, int32_t n_kept, int64_t* out_index,
                         int64_t* out_attr, pvs_stream_t stream);

/* dst[perm[e], :] = src[e, :]  (sorted -> input edge order), width floats per row.
 * Gives EGNNLayer.forward's 4th return value `edge_feat` and `att_val` in the reference's order. */
int pvs_rows_to_input_order(const float* src, float* dst, const int32_t* perm, int32_t n_edges,
                            int32_t width, pvs_stream_t stream);
/* dst[e, :] = src[perm[e], :]  (input -> sorted edge order): incoming edge_messages and their grads. */
int pvs_rows_to_sorted_order(const float* src, float* dst, const int32_t* perm, int32_t n_edges,
                             int32_t width, pvs_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * EGNNLayer.forward (egnn_satorras.py:189-206): coord2radial :178-187, edge_model :123-132,
 * edge-residual :194-202, coord_model :168-176 (+unsorted_segment_mean :340-347),
 * node_model :134-166 (+unsorted_segment_sum :332-337).
 *   h [N,H], x [N,3] inputs; m_prev [E,H] sorted or NULL (edge_messages of the previous layer)
 *   h_out [N,H], x_out [N,3] (never aliases x: the reference's in-place `coord += agg` is the
 *   caller's business), m_out [E,H] sorted or NULL (skip materialising edge_feat),
 *   att_out [E] sorted (required when PVS_EDGE_ATTENTION: also consumed by the backward),
 *   node_att_out [N] or NULL, saved: pvs_egnn_layer_saved_floats() floats kept for the backward.
 */
size_t pvs_egnn_layer_saved_floats(const PvsLayerDesc* desc, int32_t n_nodes, int32_t n_edges);
size_t pvs_egnn_layer_workspace_bytes(const PvsLayerDesc* desc, int32_t n_nodes, int32_t n_edges,
                                      int32_t backward);
int pvs_egnn_layer_fwd(const PvsLayerDesc* desc, const PvsGraph* graph, const PvsLayerParams* params,
                       const float* h, const float* x, const float* m_prev,
                       float* h_out, float* x_out, float* m_out, float* att_out,
                       float* node_att_out, float* saved,
                       void* workspace, size_t workspace_bytes, pvs_stream_t stream);

/* Forward-only pieces for virtual screening (`val`/inference.py loop of the
 * reference, point_neural_network_base.py:208-360): when many ligand poses are scored against one
 * receptor, the receptor-receptor messages of the FIRST layer (inputs: receptor features and
 * coordinates only) are the same for every pose.
 *   pvs_egnn_layer_edge_sums: the row-side sums of one layer's edge work over the edges of `graph`
 *     only: magg[i] = sum_j att_ij m_ij [N,H], xsum[i] = sum_j (x_i - x_j) s_ij [N,3] (no 1/deg).
 *   pvs_egnn_layer_fwd_partial: EGNNLayer.forward where `graph` holds only part of every row's edges
 *     and the rest enters as base_magg [N,H], base_xsum [N,3], base_deg [N] (edge counts as floats):
 *     M = base_magg + sums over graph, x' = x + (base_xsum + sums) / max(base_deg + deg_graph, 1).
 * No edge_residual, no softmax attention, H = 32 or 64; no backward. Workspace:
 * pvs_egnn_layer_workspace_bytes(desc, N, E, 2). */
int pvs_egnn_layer_edge_sums(const PvsLayerDesc* desc, const PvsGraph* graph, const PvsLayerParams* params,
                             const float* h, const float* x, float* magg, float* xsum,
                             void* workspace, size_t workspace_bytes, pvs_stream_t stream);
int pvs_egnn_layer_fwd_partial(const PvsLayerDesc* desc, const PvsGraph* graph, const PvsLayerParams* params,
                               const float* h, const float* x, const float* base_magg, const float* base_xsum,
                               const float* base_deg, float* h_out, float* x_out, float* node_att_out,
                               float* saved, void* workspace, size_t workspace_bytes, pvs_stream_t stream);

/* The edges of `full` that touch a ligand atom (bp == 0), as a CSR over the same nodes, without a
 * host round trip: rowptr_out [N+1] (its last entry = the edge count, pass it as
 * PvsGraph.n_edges_dev), row/col/etype_out with room for `capacity` edges (the call fails through
 * *status bit 2 if they do not fit). For pvs_egnn_layer_fwd_partial. */
size_t pvs_graph_filter_workspace_bytes(int32_t n_nodes);
int pvs_graph_filter_ligand_edges(const PvsGraph* full, const uint8_t* bp, int32_t capacity,
                                  int32_t* rowptr_out, int32_t* row_out, int32_t* col_out, uint8_t* etype_out,
                                  int32_t* status, void* workspace, size_t workspace_bytes, pvs_stream_t stream);

/* The radius graphs of B rigid poses of one ligand against one receptor without re-testing the
 * receptor-receptor pairs (they are pose independent: rr_rowptr [n_rec+1] / rr_col, receptor-local
 * ids, from pvs_radius_graph_* on the receptor alone). Node layout per pose: n_lig (<= 64) ligand
 * atoms first, then the n_rec receptor atoms; lig_pos [B,n_lig,3], rec_pos [n_rec,3]. Same edges,
 * classes and in-row order as generate_edges per pose. Outputs: the full CSR (rowptr [N+1], row,
 * col, etype with room for `capacity` edges, inv_deg [N]) and the CSR of its ligand-touching edges
 * (`_lig`), both with the edge count only on the device (rowptr[N]; pass it as PvsGraph.n_edges_dev:
 * forward-only use). *status bit 2 = a capacity was too small (nothing is written then). */
size_t pvs_screen_graph_state_bytes(int32_t n_poses, int32_t n_lig, int32_t n_rec);
int pvs_screen_graph_build(const float* lig_pos, const float* rec_pos, const int32_t* rr_rowptr,
                           const int32_t* rr_col, int32_t n_poses, int32_t n_lig, int32_t n_rec,
                           double inter_radius, double intra_radius, int32_t capacity, int32_t capacity_lig,
                           int32_t* rowptr, int32_t* row, int32_t* col, uint8_t* etype, float* inv_deg,
                           int32_t* rowptr_lig, int32_t* row_lig, int32_t* col_lig, uint8_t* etype_lig,
                           int32_t* status, void* state, size_t state_bytes, pvs_stream_t stream);

/* Backward of the above (what autograd replays for the reference, SURVEY.md §8a "Backward spec").
 *   g_h_out [N,H]; g_x_out [N,3] or NULL (=0: the last layer's x is unused);
 *   g_m_out [E,H] sorted or NULL (=0); att [E] sorted as written by the forward.
 *   g_h [N,H], g_x [N,3] (NULL to skip), g_m_prev [E,H] sorted (required iff edge residual applied)
 *   grads: members may be NULL to skip (coord_* MUST be NULL-safe: they are None for the last layer).
 */
int pvs_egnn_layer_bwd(const PvsLayerDesc* desc, const PvsGraph* graph, const PvsLayerParams* params,
                       const float* h, const float* x, const float* m_prev, const float* att,
                       const float* saved,
                       const float* g_h_out, const float* g_x_out, const float* g_m_out,
                       float* g_h, float* g_x, float* g_m_prev, const PvsLayerGrads* grads,
                       void* workspace, size_t workspace_bytes, pvs_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * The whole EGNNLayer stack of SartorrasEGNN.get_embeddings (egnn_satorras.py:325-328, `for layer in self.layers:
 * feats, coords, edge_attributes, edge_messages = layer(...)`) as ONE call each way. The same launches as n_layers calls of
 * pvs_egnn_layer_fwd / _bwd in a row (bit for bit the same results): what it removes is the host's share - one autograd
 * node, one ctypes call and ~15 allocations per layer and direction - which is what bounds a training step at the
 * reference's default shape (32 graphs of ~500 atoms, ~70 launches of a few microseconds: SURVEY.md §8f row 2).
 * All layers share one hidden size; layers with PVS_EDGE_RESIDUAL are refused (their [E,H] messages travel from layer to
 * layer: per-layer calls). Every layer's tensors sit in caller-owned buffers at the strides of PvsStackStrides (in
 * floats, multiples of 4):
 *   h_mid [n_layers-1][>= N*H], x_mid [n_layers-1][>= N*3]: outputs of layers 0 .. n_layers-2 (= inputs of 1 .. n_layers-1);
 *   h_out [N,H], x_out [N,3]: outputs of the last layer; att [n_layers][>= max(E,1)] (NULL unless a layer has edge
 *   attention), node_att [n_layers][>= N] (may be NULL), saved [n_layers][>= pvs_egnn_layer_saved_floats()].
 * Backward: g_h_out [N,H]; g_x_out [N,3] or NULL; g_h0 [N,H]; g_x0 [N,3] or NULL to skip; grads [n_layers]
 * (members NULL to skip, as in pvs_egnn_layer_bwd). Workspace: pvs_egnn_stack_workspace_bytes (the backward's holds the
 * two ping-pong gradient rows between layers).
 */
typedef struct PvsStackStrides {
    int64_t h_mid, x_mid, att, node_att, saved;
} PvsStackStrides;
size_t pvs_egnn_stack_workspace_bytes(const PvsLayerDesc* descs, int32_t n_layers, int32_t n_nodes, int32_t n_edges,
                                      int32_t backward);
int pvs_egnn_stack_fwd(const PvsLayerDesc* descs, const PvsLayerParams* params, int32_t n_layers,
                       const PvsGraph* graph, const PvsStackStrides* strides, const float* h0, const float* x0,
                       float* h_mid, float* x_mid, float* h_out, float* x_out, float* att, float* node_att,
                       float* saved, void* workspace, size_t workspace_bytes, pvs_stream_t stream);
int pvs_egnn_stack_bwd(const PvsLayerDesc* descs, const PvsLayerParams* params, int32_t n_layers,
                       const PvsGraph* graph, const PvsStackStrides* strides, const float* h0, const float* x0,
                       const float* h_mid, const float* x_mid, const float* att, const float* saved,
                       const float* g_h_out, const float* g_x_out, float* g_h0, float* g_x0,
                       const PvsLayerGrads* grads, void* workspace, size_t workspace_bytes, pvs_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * The thin callers either side of the layer stack.
 * y = x W^T + b: PygLinearPass.forward (pnn_geometric_base.py:83-94) and each nn.Linear of the
 * feats_linear_layers head (egnn_satorras.py:304-316, egnn_multitask.py:141-146).
 */
int pvs_linear_fwd(const float* x, const float* w, const float* b /*NULL if no bias*/, float* y,
                   int32_t n_rows, int32_t n_in, int32_t n_out, pvs_stream_t stream);
size_t pvs_linear_bwd_workspace_bytes(int32_t n_rows, int32_t n_in, int32_t n_out);
int pvs_linear_bwd(const float* x, const float* w, const float* g_y,
                   float* g_x /*NULL to skip*/, float* g_w, float* g_b /*NULL if no bias*/,
                   int32_t n_rows, int32_t n_in, int32_t n_out,
                   void* workspace, size_t workspace_bytes, pvs_stream_t stream);

/* global_mean_pool(feats, batch, size) (pnn_geometric_base.py:29-33, egnn_multitask.py:158-161):
 * nodes of one graph are contiguous (PyG collation); graph_ptr [B+1] int32 node offsets. */
int pvs_mean_pool_fwd(const float* h, const int32_t* graph_ptr, float* pooled,
                      int32_t n_graphs, int32_t width, pvs_stream_t stream);
int pvs_mean_pool_bwd(const float* g_pooled, const int32_t* graph_ptr, float* g_h,
                      int32_t n_graphs, int32_t n_nodes, int32_t width, pvs_stream_t stream);

/* global_mean_pool followed by the head's first Linear (pnn_geometric_base.py:29-36: `feats_linear_layers(global_mean_pool(
 * feats, batch))`; egnn_multitask.py:158-166) in one launch, and the backward of the pair in one launch: pooled [B, width]
 * is kept by the caller for the backward; y [B, n_out] = pooled W^T + b (b NULL: no bias). width <= 1024.
 * Backward: g_h [N, width] (NULL to skip), g_w [n_out, width], g_b [n_out] (NULL if no bias). */
int pvs_pool_head_fwd(const float* h, const int32_t* graph_ptr, const float* w, const float* b, float* pooled,
                      float* y, int32_t n_graphs, int32_t width, int32_t n_out, pvs_stream_t stream);
int pvs_pool_head_bwd(const float* g_y, const float* pooled, const float* w, const int32_t* graph_ptr,
                      float* g_h, float* g_w, float* g_b, int32_t n_graphs, int32_t n_nodes, int32_t width,
                      int32_t n_out, pvs_stream_t stream);

/* nn.BCEWithLogitsLoss() with its default mean reduction (point_neural_network_base.py:74, used at :365) over n logits:
 * loss[0] = mean(max(x, 0) - x t + log1p(exp(-|x|))), grad[i] = (sigmoid(x_i) - t_i) / n (what the backward scales by
 * the upstream gradient: pvs_scale_by_device_scalar, out[i] = a[i] * scalar[0] with the scalar in device memory). */
int pvs_bce_logits_fwd(const float* x, const float* target, int32_t n, float* loss, float* grad,
                       pvs_stream_t stream);
int pvs_scale_by_device_scalar(const float* a, const float* scalar, int32_t n, float* out, pvs_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * clip_grad_value_(params, clip) + torch.optim.Adam.step() (point_neural_network_base.py:421-422)
 * for all parameters in ONE launch. table: DEVICE array of n entries; the
 * gradients are clamped in place to [-clip, clip] (clip <= 0: no clamp), then
 *   g += weight_decay * p;  m += (1 - beta1) (g - m);  v = beta2 v + (1 - beta2) g^2;
 *   p -= (lr / bias_correction1) * m / (sqrt(v) / sqrt(bias_correction2) + eps)
 * (torch's non-amsgrad, non-maximize Adam, same operation order). bias_correction = 1 - beta^step
 * is passed by the host, which owns the step counter. */
typedef struct PvsAdamEntry {
    float* param;
    float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    int64_t numel;
} PvsAdamEntry;
int pvs_adam_clip_step(const PvsAdamEntry* table, int32_t n_tensors, double lr, double beta1, double beta2,
                       float eps, float weight_decay, double bias_correction1, double bias_correction2,
                       float clip, pvs_stream_t stream);
/* (lr, the betas and the bias corrections as DOUBLES since round 6: lr / bias_correction1, sqrt(bias_correction2), 1 - beta1
 * and 1 - beta2 are formed in double and rounded once, as torch forms the scalars it hands to lerp_ / addcmul_ / addcdiv_;
 * 1.f - 0.999f is 1.3e-5 below float(0.001).)
 * The same with the step count on the DEVICE: `step` points at one fp32 value, the number of this step (>= 1; the
 * caller advances it on the stream before the call), from which the kernel forms both bias corrections (in double, from the betas as doubles: `1 - beta ** step` as the host
 * form's caller evaluates it in Python: bit for bit the host form's update). This is what
 * torch.optim.Adam(capturable=True) does (adam.py `_multi_tensor_adam`, capturable branch: step tensors on the device) so
 * that a captured training step can be replayed; nothing about the launch depends on host state that changes per step. */
int pvs_adam_clip_step_dev(const PvsAdamEntry* table, int32_t n_tensors, double lr, double beta1, double beta2,
                           float eps, float weight_decay, const float* step, float clip, pvs_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * unsorted_segment_sum / unsorted_segment_mean (egnn_satorras.py:332-337 / :340-347) as standalone
 * operators (inside the layers these sums are fused into the edge kernels).
 *   data [E,C] fp32, ids [E] int64 in [0,N) -> out [N,C]; mean != 0 divides by max(count, 1).
 *   ptr_out [N+1] int32 (caller-owned) receives the segment offsets, needed by the backward
 *   g_data[e,:] = g_out[ids[e],:] (/ max(count,1)). status: device int32, bit0 = id out of range
 *   (the reference's scatter_add_ raises there; the kernels stay in bounds - such a row is summed into
 *   segment 0 by the forward and reads segment 0's gradient in the backward - and the HOST must read the
 *   status word and raise: pointvs_amd/functional.py does, at the latest in the backward).
 */
size_t pvs_segment_workspace_bytes(int32_t n_rows, int32_t n_segments);
int pvs_segment_reduce_fwd(const float* data, const int64_t* ids, int32_t n_rows, int32_t width,
                           int32_t n_segments, int32_t mean, float* out, int32_t* ptr_out,
                           int32_t* status, void* workspace, size_t workspace_bytes,
                           pvs_stream_t stream);
int pvs_segment_reduce_bwd(const float* g_out, const int64_t* ids, const int32_t* ptr, int32_t n_rows,
                           int32_t width, int32_t n_segments, int32_t mean, float* g_data, pvs_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Measurement hook (no reference counterpart): when enabled, the library brackets its dominant
 * kernels ("edge_fwd", "edge_bwd", "col_gather", "graph_prepare") with HIP events on the launch
 * stream; pvs_profile_read waits for those events and returns the summed kernel time. Used by
 * bench.py for the roofline figure; off by default (no events are created).
 * on: 0 = off; 1 = all four groups; otherwise a mask, bit (k + 1) = group k in the order above (an event pair costs the
 * stream a ~6 us bubble per launch, so bench.py brackets only the dominant kernel inside its timed region).
 */
int pvs_profile_enable(int on);
int pvs_profile_reset(void);
int pvs_profile_read(const char* kernel, double* total_ms, int64_t* launches);
/* The same records one by one, in launch order: ms[0 .. min(*launches, cap)) = the duration of each recorded launch of
 * the group, *launches = how many were recorded (bench.py: the launches of one step do different amounts of work - the
 * last layer's backward has no coordinate branch - and the line states the full-work average beside the overall one). */
int pvs_profile_read_each(const char* kernel, double* ms, int64_t cap, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* PVS_EGNN_H */
