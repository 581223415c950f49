// Node-level pieces of EGNNLayer.node_model (egnn_satorras.py:150-166) that are not plain linears.
#pragma once
#include "common.h"

struct PvsNodeW {
    const float *gn_w, *gn_b, *gn_ms;   // graphnorm (NULL when off)
    const float *natt_w, *natt_b;       // node attention (NULL when off)
    const float* node_gate;             // rezero / gated residual (NULL otherwise)
};

// stats[0:H] = column mean of y1, stats[H:2H] = column mean of (y1 - mean*mean_scale)^2
int pvs_graphnorm_stats(hipStream_t s, const float* y1, const float* gn_ms, int N, int H, float* stats,
                        float* shift_tmp, float* slabs);
// u = SiLU(GN(y1))  (GN skipped when w.gn_w == NULL)
int pvs_node_tail_fwd(hipStream_t s, const float* y1, const float* stats, const PvsNodeW& w, int N,
                      int H, float* u);
// h_out = residual(h, o * node_att(o)); node_att_out optional
int pvs_node_out_fwd(hipStream_t s, int H, const float* o, const float* h, const PvsNodeW& w,
                     uint32_t flags, int att_act, int N, float* h_out, float* natt_out);
// backward of pvs_node_out_fwd: g_o (wrt o), g_h (residual branch, overwritten), gl[N] (wrt the
// node-attention logit), t1[N,H] = gl*o, tg[N,H] = per-element gate-gradient contributions
int pvs_node_out_bwd(hipStream_t s, int H, const float* g_hout, const float* o, const float* h,
                     const PvsNodeW& w, uint32_t flags, int att_act, int N, float* g_o, float* g_h,
                     float* gl, float* t1, float* tg);
// g_yn = g_u * SiLU'(yn)
int pvs_node_tail_bwd1(hipStream_t s, const float* g_u, const float* y1, const float* stats,
                       const PvsNodeW& w, int N, int H, float* g_yn);
// graphnorm backward: S1 = sum g_yn, S2 = sum g_yn*y1 (both [H]) -> param grads + coefs[4H]
// (coefs[3H..4H) = the gradient of the bias in front of the norm in closed form)
int pvs_graphnorm_bwd_coefs(hipStream_t s, const float* S1, const float* S2, const float* stats,
                            const PvsNodeW& w, int N, int H, float* g_w, float* g_b, float* g_ms,
                            float* coefs);
// g_y1 = g_yn*coefA + (y1 - shift)*coefB + coefC   (in place on g_yn allowed)
int pvs_node_tail_bwd2(hipStream_t s, const float* g_yn, const float* y1, const float* stats,
                       const PvsNodeW& w, const float* coefs, int N, int H, float* g_y1);
// gxagg = g_x_out * inv_deg (if g_x_out), softD = rowdot(Magg, gM) (if softD)
int pvs_prep_edge_bwd(hipStream_t s, const float* g_x_out, const float* inv_deg, const float* Magg,
                      const float* gM, int N, int H, float* gxagg, float* softD, float* zero_gPQ,
                      float* zero_gx_row);
// out[0] = sum_i v[i]
int pvs_copy_small(hipStream_t s, const float* src, float* dst, int n);
int pvs_sum_vec(hipStream_t s, const float* v, int n, float* out);
