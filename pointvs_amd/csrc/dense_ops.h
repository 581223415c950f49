// Node-level dense helpers (N rows x <=256 channels): small, deterministic, atomics-free.
#pragma once
#include "common.h"

// y[n*ldy + c] (=|+=) b[c] + sum_k x[n*ldx+k] * W[c*swc + k*swk] + sum_k2 x2[n*ldx2+k2] * W2[c*swc2 + k2*swk2]
int pvs_launch_linear(hipStream_t s, float* y, int ldy, const float* x, int ldx, const float* W,
                      int swc, int swk, const float* b, const float* x2, int ldx2, const float* W2,
                      int swc2, int swk2, int N, int K, int K2, int C, bool accumulate, int epi = 0,
                      const float* aux_in = nullptr, int ld_in = 0, float* aux_out = nullptr, int ld_out = 0,
                      const struct PvsLinearExt* ext = nullptr);
// Extras of the MFMA linear (shapes of pvs_linear_epilogue_supported only; launches folded into a product that deals
// the node rows to its lanes anyway):
struct PvsLinearExt {
    float* y1 = nullptr;          // column block 1 (outputs 32..63) -> y1[n*ldy1 + (c - 32)] instead of y[n*ldy + c]
    int ldy1 = 0;
    int acc1 = -1;                // accumulate flag of column block 1 (-1: the call's)
    long long w_shift1 = 0;       // added to the weight offsets of the column blocks from `shift_block` on (a second slice of one weight matrix)
    int shift_block = 1;
    int groups = 1;               // launch the product as `groups` groups of workgroups (grid.y), each with its own C / groups output columns
    int side_group = 0;           // 1: the side jobs below get a group of workgroups of their own (grid.y = groups) instead of riding on group 0
    int bias_blocks = 8;          // column blocks, from 0, that take the bias
    // per-row side jobs (valid rows):
    float* zero_rows = nullptr;   // zero_rows[n*zero_ld + 0 .. zero_w) = 0   (zero_w % 8 == 0, 16-byte aligned rows)
    int zero_w = 0, zero_ld = 0;
    float* zero3 = nullptr;       // zero3[3n .. 3n+3) = 0
    const float* copy3_src = nullptr;     // copy3_dst[3n + i] = copy3_src[3n + i]
    float* copy3_dst = nullptr;
    const float* scale3_src = nullptr;    // scale3_dst[3n + i] = scale3_src[3n + i] * scale3_by[n]
    const float* scale3_by = nullptr;
    float* scale3_dst = nullptr;
};
// elementwise epilogue on the product (only on the MFMA path: check first):
//   1: aux_out = SiLU(y)   2: aux_out = aux_in + y   3: y *= SiLU'(aux_in)   4: aux_out = y
enum { PVS_EPI_NONE = 0, PVS_EPI_SILU_OUT = 1, PVS_EPI_ADD_OUT = 2, PVS_EPI_MUL_SILU_GRAD = 3, PVS_EPI_COPY_OUT = 4 };
bool pvs_linear_epilogue_supported(int ldy, int ldx, int ldx2, int K, int K2, int C, const void* y,
                                   const void* x, const void* x2);

// The node MLP of a layer as one launch each way (H = 32, 64; layers without GraphNorm and without rezero / gated
// residual; node attention (natt_w != NULL) and the plain residual are part of the chain):
//   forward   y1 = [h | Magg] W1^T + b1, u = SiLU(y1), o = u W2^T + b2, a = act(natt_w . o + natt_b) (or 1),
//             h_out = (residual ? h : 0) + a o;  natt_out[n] = a
//   backward  g_o = g_hout a + g_l natt_w with g_l = act'(l) (g_hout . o)  [written with t1 = g_l o, gl = g_l when
//             gated; = g_hout otherwise, nothing written], g_y1 = (g_o W2) * SiLU'(y1),
//             g_h = (residual ? g_hout : 0) + g_y1 W1[:, :H], gM = g_y1 W1[:, H:]  (+ the row side jobs of `ext`)
bool pvs_node_mlp_fused_supported(int H, const void* a, const void* b, const void* c, const void* d);
int pvs_launch_node_mlp_fwd(hipStream_t s, int H, int N, const float* h, const float* Magg, const float* W1,
                            const float* b1, const float* W2, const float* b2, bool residual, const float* natt_w,
                            const float* natt_b, int att_act, float* y1, float* u, float* o, float* h_out,
                            float* natt_out);
int pvs_launch_node_mlp_bwd(hipStream_t s, int H, int N, const float* g_hout, const float* o, const float* y1,
                            const float* W1, const float* W2, bool residual, const float* natt_w, const float* natt_b,
                            int att_act, float* g_o, float* t1, float* gl, float* g_y1, float* g_h, float* gM,
                            const PvsLinearExt* ext);

// number of float slabs a column reduction / tsgemm over N rows needs: slabs * width floats
int pvs_reduce_blocks(int N);
int pvs_colreduce_blocks(int N);   // slabs of pvs_launch_colreduce: [pvs_colreduce_blocks(N)][C] (<= 4 x pvs_reduce_blocks)

// out[c*ldo + k] (=|+=) sum_n A[n*lda + c] * B[n*ldb + k]   (weight gradients), C*K <= 8192
// colsum_out != NULL (narrow right operand only, pvs_tsgemm_colsum_supported): colsum_out[c] = sum_n A[n*lda + c] out of
// the same pass (a ones column in the zero padding of B)
bool pvs_tsgemm_colsum_supported(int N, int C, int K);
int pvs_launch_tsgemm_tn(hipStream_t s, float* out, int ldo, const float* A, int lda, const float* B,
                         int ldb, int N, int C, int K, float* slabs, bool accumulate, float* colsum_out = nullptr);

enum { PVS_COL_SUM_A = 0, PVS_COL_SUM_AB = 1, PVS_COL_SUMSQ_SHIFT = 2 };
// out[c] (=|+=) scale * sum_n f(A[n*lda+c], B[n*ldb+c] | shift[c])
int pvs_launch_colreduce(hipStream_t s, int mode, float* out, const float* A, int lda, const float* B,
                         int ldb, const float* shift, int N, int C, float scale, float* slabs,
                         bool accumulate);

// out[map(o)] (=|+=) sum_g slabs[g*width + o], o < width; map(o) = (o / inner) * ldo + o % inner
int pvs_launch_reduce_slabs(hipStream_t s, float* out, int ldo, int inner, const float* slabs,
                            int n_slabs, int width, bool accumulate);

// Sum over slabs of 32 consecutive outputs by one 256-thread workgroup (the one reduction order of every slab
// reduction in the library): thread (sl = tid >> 5, ol = tid & 31) adds slabs sl, sl + 8, ... of output o0 + ol in four
// independent partial sums; the eight partials are then added in order. Valid in the threads with sl == 0.
__device__ __forceinline__ float pvs_slab_sum32(const float* __restrict__ slabs, int n_slabs, int width, int o,
                                                float (*part)[33]) {
    const int ol = threadIdx.x & 31, sl = threadIdx.x >> 5;
    float s = 0.f;
    if (o < width) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int gidx = sl;
        for (; gidx + 24 < n_slabs; gidx += 32) {
            a0 += slabs[(size_t)gidx * width + o];
            a1 += slabs[(size_t)(gidx + 8) * width + o];
            a2 += slabs[(size_t)(gidx + 16) * width + o];
            a3 += slabs[(size_t)(gidx + 24) * width + o];
        }
        for (; gidx < n_slabs; gidx += 8) a0 += slabs[(size_t)gidx * width + o];
        s = (a0 + a1) + (a2 + a3);
    }
    __syncthreads();            // (part may still be read from a previous call)
    part[sl][ol] = s;
    __syncthreads();
    float t = 0.f;
    if (sl == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) t += part[k][ol];
    }
    return t;
}

// the two-set slab reduction of a layer's backward as a job description (pvs_launch_reduce_slabs2, or extra workgroups
// of the weight-gradient pass: independent work in one launch)
struct PvsReduce2Args {
    float* out_a = nullptr;
    const float* slabs_a = nullptr;
    int n_a = 0, width_a = 0, skip_lo = 0, skip_hi = 0;
    float* out_b = nullptr;
    const float* slabs_b = nullptr;
    int n_b = 0, width_b = 0;
    __host__ __device__ int blocks_a() const { return (width_a + 31) / 32; }
    __host__ __device__ int blocks() const { return out_a ? (width_a + 31) / 32 + (width_b + 31) / 32 : 0; }
};
// workgroup `block` (of args.blocks()) of that reduction
__device__ __forceinline__ void pvs_reduce2_block(const PvsReduce2Args& a, int block, float (*part)[33]) {
    const bool is_a = block < a.blocks_a();
    const float* slabs = is_a ? a.slabs_a : a.slabs_b;
    float* out = is_a ? a.out_a : a.out_b;
    const int n_slabs = is_a ? a.n_a : a.n_b, width = is_a ? a.width_a : a.width_b;
    const int o = (block - (is_a ? 0 : a.blocks_a())) * 32 + (threadIdx.x & 31);
    const float t = pvs_slab_sum32(slabs, n_slabs, width, o, part);
    if ((threadIdx.x >> 5) == 0 && o < width && !(is_a && o >= a.skip_lo && o < a.skip_hi)) out[o] = t;
}

// out_a[o] = sum_g slabs_a[g*width_a + o] except o in [skip_lo, skip_hi); out_b[o] = sum_g slabs_b[g*width_b + o]
int pvs_launch_reduce_slabs2(hipStream_t s, float* out_a, const float* slabs_a, int n_a, int width_a, int skip_lo,
                             int skip_hi, float* out_b, const float* slabs_b, int n_b, int width_b);

// All node-level weight gradients of one EGNNLayer backward in one pass over the N rows (H = 32, 64):
//   node_w2 = g_o^T u, node_w1 = g_y1^T [h | Magg], edge_w1[:, P/Q columns] = [gP | gQ]^T h and the
//   bias gradients node_b2 / node_b1 / edge_b1 = column sums of g_o / g_y1 / gP.
// Replaces five tsgemm + three colreduce launches (and their slab reductions) by one product
// kernel, one slab reduction and one scatter. slabs: pvs_node_wgrads_slab_floats(N, H) floats.
struct PvsNodeWgradIn {
    const float *g_o, *g_y1, *gPQ;   // [N,H], [N,H], [N,2H] (P part | Q part)
    const float *u, *h, *Magg;       // [N,H] each
    const float *t1 = nullptr;       // [N,H] g_l * o of the node gate, or NULL
    const float *gl = nullptr;       // [N]   g_l of the node gate, or NULL
};
struct PvsNodeWgradOut {
    float *node_w2, *node_w1, *edge_w1;     // required
    float *node_b2, *node_b1, *edge_b1;     // NULL to skip
    int ld1, off_q, perm;                   // edge_w1 row stride, column of the Q block, P/Q share columns
    float *natt_w = nullptr, *natt_b = nullptr;   // node gate: column sums of t1 [H] and the sum of gl [1], or NULL
};
// reduced sums layout (per 32x32 sub-block (bo, bi) of the H x H products): 5 products x [32][32],
// then 5 column-sum vectors x [32] (valid for bi == 0): three biases, the node gate's weight, its bias (element 0 of bo == 0)
#define PVS_WG_PRODUCTS 5
#define PVS_WG_BIAS 5
#define PVS_WG_SLAB (PVS_WG_PRODUCTS * 1024 + PVS_WG_BIAS * 32)
// scatter of the reduced sums into the gradient tensors (device side, for fusing into another small kernel)
__device__ __forceinline__ void pvs_node_wgrads_scatter(const float* __restrict__ gsum, const PvsNodeWgradOut& out,
                                                        int H, int tid, int stride);
int pvs_node_wgrads_supported(int H);
size_t pvs_node_wgrads_slab_floats(int N, int H);
// scatter = false: stop after the slab reduction; *gsum_out then points at the reduced sums for a later
// pvs_node_wgrads_scatter
// extra != NULL: the launch also carries that (independent) two-set slab reduction as additional workgroups.
// slabs_out != NULL: stop after the product kernel; *slabs_out describes the per-row-block partials for a kernel
// that reduces and scatters them itself (pvs_node_wgrads_reduce_scatter32).
struct PvsNodeWgradSlabs { const float* slabs = nullptr; int n_slabs = 0, width = 0; };
// gh_job != NULL: the launch also computes g_h[n, :] += gPQ[n, 0:H] W1[:, 0:H] + gPQ[n, H:2H] W1[:, off_q:off_q+H]
// (W1 = edge_mlp.0's weight, row stride ld1) - the product of this point of the backward that does not depend on the
// weight gradients - as further workgroups.
struct PvsGhJob { float* g_h = nullptr; const float* gPQ = nullptr; const float* W1 = nullptr; int ld1 = 0, off_q = 0, blocks = 0; };
int pvs_launch_node_wgrads(hipStream_t s, int H, int N, const PvsNodeWgradIn& in, const PvsNodeWgradOut& out,
                           float* slabs, bool scatter = true, const float** gsum_out = nullptr,
                           const PvsReduce2Args* extra = nullptr, PvsNodeWgradSlabs* slabs_out = nullptr,
                           const PvsGhJob* gh_job = nullptr);

__device__ __forceinline__ void pvs_node_wgrads_scatter(const float* __restrict__ gsum, const PvsNodeWgradOut& out,
                                                        int H, int tid, int stride) {
    const int HB = H / 32;
    for (int i = tid; i < H * H; i += stride) {
        const int c = i / H, k = i % H;
        const float* b = gsum + (size_t)((c >> 5) * HB + (k >> 5)) * PVS_WG_SLAB + (c & 31) * 32 + (k & 31);
        out.node_w2[(size_t)c * H + k] = b[0];
        out.node_w1[(size_t)c * 2 * H + k] = b[1024];
        out.node_w1[(size_t)c * 2 * H + H + k] = b[2 * 1024];
        if (out.perm) {
            out.edge_w1[(size_t)c * out.ld1 + k] = b[3 * 1024] + b[4 * 1024];
        } else {
            out.edge_w1[(size_t)c * out.ld1 + k] = b[3 * 1024];
            out.edge_w1[(size_t)c * out.ld1 + out.off_q + k] = b[4 * 1024];
        }
    }
    for (int c = tid; c < H; c += stride) {
        const float* b = gsum + (size_t)((c >> 5) * HB) * PVS_WG_SLAB + PVS_WG_PRODUCTS * 1024 + (c & 31);
        if (out.node_b2) out.node_b2[c] = b[0];
        if (out.node_b1) out.node_b1[c] = b[32];
        if (out.edge_b1) out.edge_b1[c] = b[64];
        if (out.natt_w) out.natt_w[c] = b[96];
        if (out.natt_b && c == 0) out.natt_b[0] = b[128];
    }
}

// Reduce 32 consecutive entries [o0, o0 + 32) of the node weight-gradient slabs and write them where
// pvs_node_wgrads_scatter would (same sums, same order: the scatter is a permutation, except that with shared P / Q
// columns two entries add up). 256 threads; o0 % 32 == 0, so the 32 entries share product and sub-block.
__device__ __forceinline__ void pvs_node_wgrads_reduce_scatter32(const PvsNodeWgradSlabs& ns, const PvsNodeWgradOut& out,
                                                                 int H, int o0, float (*part)[33]) {
    const int HB = H / 32;
    const int sub = o0 / PVS_WG_SLAB, r0 = o0 - sub * PVS_WG_SLAB;      // block-uniform
    const int bo = sub / HB, bi = sub - bo * HB;
    const int ol = threadIdx.x & 31;
    const bool writer = (threadIdx.x >> 5) == 0;
    if (o0 >= ns.width) return;
    if (r0 < PVS_WG_PRODUCTS * 1024) {
        const int prod = r0 / 1024;
        if (prod == 4 && out.perm) return;                               // added into the P entry below
        float t = pvs_slab_sum32(ns.slabs, ns.n_slabs, ns.width, o0 + ol, part);
        if (prod == 3 && out.perm) t += pvs_slab_sum32(ns.slabs, ns.n_slabs, ns.width, o0 + 1024 + ol, part);
        if (!writer) return;
        const int c = 32 * bo + (r0 % 1024) / 32, k = 32 * bi + ol;
        if (prod == 0) out.node_w2[(size_t)c * H + k] = t;
        else if (prod == 1) out.node_w1[(size_t)c * 2 * H + k] = t;
        else if (prod == 2) out.node_w1[(size_t)c * 2 * H + H + k] = t;
        else if (prod == 3) out.edge_w1[(size_t)c * out.ld1 + k] = t;
        else out.edge_w1[(size_t)c * out.ld1 + out.off_q + k] = t;
        return;
    }
    if (bi != 0) return;                                                 // column-sum vectors live in the bi == 0 slabs
    const int vec = (r0 - PVS_WG_PRODUCTS * 1024) / 32;
    const float t = pvs_slab_sum32(ns.slabs, ns.n_slabs, ns.width, o0 + ol, part);
    if (!writer) return;
    const int c = 32 * bo + ol;
    if (vec == 0) { if (out.node_b2) out.node_b2[c] = t; }
    else if (vec == 1) { if (out.node_b1) out.node_b1[c] = t; }
    else if (vec == 2) { if (out.edge_b1) out.edge_b1[c] = t; }
    else if (vec == 3) { if (out.natt_w) out.natt_w[c] = t; }
    else if (vec == 4) { if (out.natt_b && c == 0) out.natt_b[0] = t; }
}
