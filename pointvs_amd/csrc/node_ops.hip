#include "node_ops.h"

namespace {

constexpr float kGnEps = 1e-5f;   // torch_geometric GraphNorm default eps

__global__ void k_gn_shift(const float* mean, const float* ms, int H, float* shift) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < H) shift[c] = mean[c] * ms[c];
}

__device__ __forceinline__ float gn_apply(float y, int c, const float* stats, const PvsNodeW& w, int H) {
    if (!w.gn_w) return y;
    const float shift = stats[c] * w.gn_ms[c];
    const float rstd = 1.0f / sqrtf(stats[H + c] + kGnEps);
    return w.gn_w[c] * (y - shift) * rstd + w.gn_b[c];
}

__global__ void k_node_tail_fwd(const float* __restrict__ y1, const float* __restrict__ stats,
                                PvsNodeW w, int N, int H, float* __restrict__ u) {
    long long total = (long long)N * H;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        int c = (int)(i % H);
        u[i] = pvs_silu(gn_apply(y1[i], c, stats, w, H));
    }
}

template <int H>
__global__ void __launch_bounds__(256)
k_node_out_fwd(const float* __restrict__ o, const float* __restrict__ h, PvsNodeW w, uint32_t flags,
               int att_act, int N, float* __restrict__ h_out, float* __restrict__ natt_out) {
    constexpr int NPW = 64 / H;
    const int lane = threadIdx.x & 63;
    const int c = lane % H, sub = lane / H;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int n_waves = (gridDim.x * blockDim.x) >> 6;
    const bool natt = flags & PVS_NODE_ATTENTION;
    const float wna = natt ? w.natt_w[c] : 0.f;
    const float bna = natt ? w.natt_b[0] : 0.f;
    float gate = 1.f;
    if ((flags & PVS_RESIDUAL) && (flags & (PVS_REZERO | PVS_GATED_RESIDUAL))) {
        gate = w.node_gate[0];
        if (flags & PVS_GATED_RESIDUAL) gate = fmaxf(gate, 0.f);
    }
    for (int nb = wave * NPW; nb < N; nb += n_waves * NPW) {
        const int n = nb + sub;
        const bool valid = n < N;
        const int nn = valid ? n : N - 1;
        float ov = o[(size_t)nn * H + c];
        if (natt) {
            const float l = pvs_group_sum<H>(wna * ov) + bna;
            const float a = pvs_att_act(att_act, l);
            ov *= a;
            if (natt_out && valid && c == 0) natt_out[n] = a;
        }
        float r = ov;
        if (flags & PVS_RESIDUAL) {
            const float hv = h[(size_t)nn * H + c];
            if (flags & PVS_REZERO) r = hv + gate * ov;
            else if (flags & PVS_GATED_RESIDUAL) r = gate * ov + (1.f - gate) * hv;
            else r = hv + ov;
        }
        if (valid) h_out[(size_t)n * H + c] = r;
    }
}

template <int H>
__global__ void __launch_bounds__(256)
k_node_out_bwd(const float* __restrict__ g_hout, const float* __restrict__ o,
               const float* __restrict__ h, PvsNodeW w, uint32_t flags, int att_act, int N,
               float* __restrict__ g_o, float* __restrict__ g_h, float* __restrict__ gl,
               float* __restrict__ t1, float* __restrict__ tg) {
    constexpr int NPW = 64 / H;
    const int lane = threadIdx.x & 63;
    const int c = lane % H, sub = lane / H;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int n_waves = (gridDim.x * blockDim.x) >> 6;
    const bool natt = flags & PVS_NODE_ATTENTION;
    const bool res = flags & PVS_RESIDUAL;
    const float wna = natt ? w.natt_w[c] : 0.f;
    const float bna = natt ? w.natt_b[0] : 0.f;
    float gate_raw = 0.f, gate = 1.f;
    const bool gated = res && (flags & PVS_GATED_RESIDUAL), rez = res && (flags & PVS_REZERO);
    if (gated || rez) {
        gate_raw = w.node_gate[0];
        gate = gated ? fmaxf(gate_raw, 0.f) : gate_raw;
    }
    for (int nb = wave * NPW; nb < N; nb += n_waves * NPW) {
        const int n = nb + sub;
        const bool valid = n < N;
        const int nn = valid ? n : N - 1;
        const size_t idx = (size_t)nn * H + c;
        const float go = g_hout[idx];
        const float ov = o[idx];
        float l = 0.f, a = 1.f;
        if (natt) {
            l = pvs_group_sum<H>(wna * ov) + bna;
            a = pvs_att_act(att_act, l);
        }
        const float o2 = ov * a;    // value entering the residual
        // residual
        float g_o2 = go, g_hres = 0.f, g_gate_el = 0.f;
        if (res) {
            if (rez) { g_o2 = gate * go; g_hres = go; g_gate_el = go * o2; }
            else if (gated) {
                g_o2 = gate * go; g_hres = (1.f - gate) * go;
                g_gate_el = gate_raw > 0.f ? go * (o2 - h[idx]) : 0.f;
            } else { g_hres = go; }
        }
        // node attention: o2 = ov * a(l), l = wna . ov + bna
        float g_ov = g_o2, g_l = 0.f;
        if (natt) {
            const float dot = pvs_group_sum<H>(g_o2 * ov);
            g_l = pvs_att_act_grad(att_act, l, a) * dot;
            g_ov = g_o2 * a + g_l * wna;
        }
        if (valid) {
            g_o[idx] = g_ov;
            g_h[idx] = g_hres;
            if (natt) { t1[idx] = g_l * ov; if (c == 0) gl[n] = g_l; }
            if (gated || rez) tg[idx] = g_gate_el;
        }
    }
}

// H = 64 * K (K >= 2): one wave per node, lane owns channels lane + 64 k. Same arithmetic as the kernels above.
template <int K>
__global__ void __launch_bounds__(256)
k_node_out_fwd_wide(const float* __restrict__ o, const float* __restrict__ h, PvsNodeW w, uint32_t flags,
                    int att_act, int N, float* __restrict__ h_out, float* __restrict__ natt_out) {
    constexpr int H = 64 * K;
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int n_waves = (gridDim.x * blockDim.x) >> 6;
    const bool natt = flags & PVS_NODE_ATTENTION;
    const float bna = natt ? w.natt_b[0] : 0.f;
    float gate = 1.f;
    if ((flags & PVS_RESIDUAL) && (flags & (PVS_REZERO | PVS_GATED_RESIDUAL))) {
        gate = w.node_gate[0];
        if (flags & PVS_GATED_RESIDUAL) gate = fmaxf(gate, 0.f);
    }
    for (int n = wave; n < N; n += n_waves) {
        float ov[K];
        float part = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            ov[k] = o[(size_t)n * H + lane + 64 * k];
            if (natt) part = fmaf(w.natt_w[lane + 64 * k], ov[k], part);
        }
        float a = 1.f;
        if (natt) {
            a = pvs_att_act(att_act, pvs_group_sum<64>(part) + bna);
            if (natt_out && lane == 0) natt_out[n] = a;
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float v = ov[k] * a;
            float r = v;
            if (flags & PVS_RESIDUAL) {
                const float hv = h[(size_t)n * H + lane + 64 * k];
                if (flags & PVS_REZERO) r = hv + gate * v;
                else if (flags & PVS_GATED_RESIDUAL) r = gate * v + (1.f - gate) * hv;
                else r = hv + v;
            }
            h_out[(size_t)n * H + lane + 64 * k] = r;
        }
    }
}

template <int K>
__global__ void __launch_bounds__(256)
k_node_out_bwd_wide(const float* __restrict__ g_hout, const float* __restrict__ o,
                    const float* __restrict__ h, PvsNodeW w, uint32_t flags, int att_act, int N,
                    float* __restrict__ g_o, float* __restrict__ g_h, float* __restrict__ gl,
                    float* __restrict__ t1, float* __restrict__ tg) {
    constexpr int H = 64 * K;
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int n_waves = (gridDim.x * blockDim.x) >> 6;
    const bool natt = flags & PVS_NODE_ATTENTION;
    const bool res = flags & PVS_RESIDUAL;
    const float bna = natt ? w.natt_b[0] : 0.f;
    float gate_raw = 0.f, gate = 1.f;
    const bool gated = res && (flags & PVS_GATED_RESIDUAL), rez = res && (flags & PVS_REZERO);
    if (gated || rez) {
        gate_raw = w.node_gate[0];
        gate = gated ? fmaxf(gate_raw, 0.f) : gate_raw;
    }
    for (int n = wave; n < N; n += n_waves) {
        float go[K], ov[K], wna[K];
        float part = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const size_t idx = (size_t)n * H + lane + 64 * k;
            go[k] = g_hout[idx];
            ov[k] = o[idx];
            wna[k] = natt ? w.natt_w[lane + 64 * k] : 0.f;
            part = fmaf(wna[k], ov[k], part);
        }
        float l = 0.f, a = 1.f;
        if (natt) {
            l = pvs_group_sum<64>(part) + bna;
            a = pvs_att_act(att_act, l);
        }
        float g_o2[K], g_hres[K], g_gate_el[K];
        float dpart = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float o2 = ov[k] * a;
            g_o2[k] = go[k]; g_hres[k] = 0.f; g_gate_el[k] = 0.f;
            if (res) {
                if (rez) { g_o2[k] = gate * go[k]; g_hres[k] = go[k]; g_gate_el[k] = go[k] * o2; }
                else if (gated) {
                    g_o2[k] = gate * go[k]; g_hres[k] = (1.f - gate) * go[k];
                    g_gate_el[k] = gate_raw > 0.f ? go[k] * (o2 - h[(size_t)n * H + lane + 64 * k]) : 0.f;
                } else { g_hres[k] = go[k]; }
            }
            dpart = fmaf(g_o2[k], ov[k], dpart);
        }
        float g_l = 0.f;
        if (natt) g_l = pvs_att_act_grad(att_act, l, a) * pvs_group_sum<64>(dpart);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const size_t idx = (size_t)n * H + lane + 64 * k;
            g_o[idx] = natt ? g_o2[k] * a + g_l * wna[k] : g_o2[k];
            g_h[idx] = g_hres[k];
            if (natt) t1[idx] = g_l * ov[k];
            if (gated || rez) tg[idx] = g_gate_el[k];
        }
        if (natt && lane == 0) gl[n] = g_l;
    }
}

__global__ void k_node_tail_bwd1(const float* __restrict__ g_u, const float* __restrict__ y1,
                                 const float* __restrict__ stats, PvsNodeW w, int N, int H,
                                 float* __restrict__ g_yn) {
    long long total = (long long)N * H;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        int c = (int)(i % H);
        const float yn = gn_apply(y1[i], c, stats, w, H);
        g_yn[i] = g_u[i] * pvs_silu_grad(yn, pvs_sigmoid(yn));
    }
}

// yn = gw*cc*rstd + gb, cc = y1 - ms*mean, var = mean(cc^2)
__global__ void k_gn_bwd_coefs(const float* S1, const float* S2raw, const float* stats, PvsNodeW w,
                               int N, int H, float* g_w, float* g_b, float* g_ms, float* coefs) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= H) return;
    const float mean = stats[c], var = stats[H + c];
    const float ms = w.gn_ms[c], gw = w.gn_w[c];
    const float shift = mean * ms;
    const float rstd = 1.0f / sqrtf(var + kGnEps);
    const float s1 = S1[c];                  // sum g_yn
    const float s2 = S2raw[c] - shift * s1;  // sum g_yn * cc
    const float g_var = -0.5f * gw * rstd * rstd * rstd * s2;
    const float sum_cc = (float)N * (mean - shift);
    const float sum_gc = gw * rstd * s1 + g_var * 2.0f * sum_cc / (float)N;
    const float g_shift = -sum_gc;
    if (g_w) g_w[c] = s2 * rstd;
    if (g_b) g_b[c] = s1;
    if (g_ms) g_ms[c] = g_shift * mean;
    const float g_mean = g_shift * ms;
    coefs[c] = gw * rstd;
    coefs[H + c] = 2.0f * g_var / (float)N;
    coefs[2 * H + c] = g_mean / (float)N;
    // The gradient of the bias in front of the norm (node_mlp.0.bias) = sum_n g_y1[n] in closed form: the three terms of
    // k_node_tail_bwd2 summed over the nodes are  gw rstd s1 + (2 g_var / N) sum_cc + g_mean = sum_gc (1 - ms).  At the
    // initial mean_scale = 1 it is zero IDENTICALLY (the norm removes any shift of its input); the column sum of N
    // rounded g_y1 values leaves 1e-6 of their magnitude there instead (round 4: strict per-tensor parity).
    coefs[3 * H + c] = sum_gc * (1.0f - ms);
}

__global__ void k_copy_small(const float* __restrict__ src, float* __restrict__ dst, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

__global__ void k_node_tail_bwd2(const float* __restrict__ g_yn, const float* __restrict__ y1,
                                 const float* __restrict__ stats, PvsNodeW w,
                                 const float* __restrict__ coefs, int N, int H,
                                 float* __restrict__ g_y1) {
    long long total = (long long)N * H;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        int c = (int)(i % H);
        const float shift = stats[c] * w.gn_ms[c];
        g_y1[i] = g_yn[i] * coefs[c] + (y1[i] - shift) * coefs[H + c] + coefs[2 * H + c];
    }
}

// thread = (node, 16-byte quad of its H channels): quad 0 does the per-node work, every thread
// clears its piece of the row part of gPQ (rows without edges are never written by the MFMA edge
// backward), so no separate memsets are needed
__global__ void k_prep_edge_bwd(const float* __restrict__ g_x_out, const float* __restrict__ inv_deg,
                                const float* __restrict__ Magg, const float* __restrict__ gM, int N,
                                int H, float* __restrict__ gxagg, float* __restrict__ softD,
                                float* __restrict__ zero_gPQ, float* __restrict__ zero_gx_row) {
    const int qpr = H / 4;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = t / qpr, q = t - n * qpr;
    if (n >= N) return;
    if (zero_gPQ) *reinterpret_cast<float4*>(zero_gPQ + (size_t)n * 2 * H + 4 * q) = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q != 0) return;
    if (zero_gx_row) { zero_gx_row[3 * n] = 0.f; zero_gx_row[3 * n + 1] = 0.f; zero_gx_row[3 * n + 2] = 0.f; }
    if (gxagg) {
        const float inv = inv_deg[n];
        gxagg[3 * n] = g_x_out[3 * n] * inv;
        gxagg[3 * n + 1] = g_x_out[3 * n + 1] * inv;
        gxagg[3 * n + 2] = g_x_out[3 * n + 2] * inv;
    }
    if (softD) {
        float d = 0.f;
        for (int c = 0; c < H; ++c) d = fmaf(Magg[(size_t)n * H + c], gM[(size_t)n * H + c], d);
        softD[n] = d;
    }
}

__global__ void k_sum_vec(const float* v, int n, float* out) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < n; ++i) s += v[i];
        out[0] = s;
    }
}

int ew_grid(long long total) {
    long long b = (total + 255) / 256;
    if (b < 1) b = 1;
    if (b > 4096) b = 4096;
    return (int)b;
}

}  // namespace

#include "dense_ops.h"

int pvs_graphnorm_stats(hipStream_t s, const float* y1, const float* gn_ms, int N, int H, float* stats,
                        float* shift_tmp, float* slabs) {
    int rc = pvs_launch_colreduce(s, PVS_COL_SUM_A, stats, y1, H, nullptr, 0, nullptr, N, H,
                                  1.0f / (float)N, slabs, false);
    if (rc) return rc;
    k_gn_shift<<<(H + 63) / 64, 64, 0, s>>>(stats, gn_ms, H, shift_tmp);
    PVS_CHECK_LAUNCH();
    return pvs_launch_colreduce(s, PVS_COL_SUMSQ_SHIFT, stats + H, y1, H, nullptr, 0, shift_tmp, N, H,
                                1.0f / (float)N, slabs, false);
}

int pvs_node_tail_fwd(hipStream_t s, const float* y1, const float* stats, const PvsNodeW& w, int N,
                      int H, float* u) {
    k_node_tail_fwd<<<ew_grid((long long)N * H), 256, 0, s>>>(y1, stats, w, N, H, u);
    PVS_CHECK_LAUNCH();
    return 0;
}

#define PVS_NODE_DISPATCH_H(H, ...)                                    \
    switch (H) {                                                       \
        case 16: { constexpr int HH = 16; __VA_ARGS__; break; }        \
        case 32: { constexpr int HH = 32; __VA_ARGS__; break; }        \
        case 64: { constexpr int HH = 64; __VA_ARGS__; break; }        \
        default: pvs_set_error("node kernels: hidden size %d unsupported (16,32,64)", H); return -1; \
    }

int pvs_node_out_fwd(hipStream_t s, int H, const float* o, const float* h, const PvsNodeW& w,
                     uint32_t flags, int att_act, int N, float* h_out, float* natt_out) {
    const int blocks = ew_grid((long long)N * H);
    if (H == 128) {
        k_node_out_fwd_wide<2><<<blocks, 256, 0, s>>>(o, h, w, flags, att_act, N, h_out, natt_out);
        PVS_CHECK_LAUNCH();
        return 0;
    }
    PVS_NODE_DISPATCH_H(H, {
        k_node_out_fwd<HH><<<blocks, 256, 0, s>>>(o, h, w, flags, att_act, N, h_out, natt_out);
    });
    PVS_CHECK_LAUNCH();
    return 0;
}

int pvs_node_out_bwd(hipStream_t s, int H, const float* g_hout, const float* o, const float* h,
                     const PvsNodeW& w, uint32_t flags, int att_act, int N, float* g_o, float* g_h,
                     float* gl, float* t1, float* tg) {
    const int blocks = ew_grid((long long)N * H);
    if (H == 128) {
        k_node_out_bwd_wide<2><<<blocks, 256, 0, s>>>(g_hout, o, h, w, flags, att_act, N, g_o, g_h, gl, t1, tg);
        PVS_CHECK_LAUNCH();
        return 0;
    }
    PVS_NODE_DISPATCH_H(H, {
        k_node_out_bwd<HH><<<blocks, 256, 0, s>>>(g_hout, o, h, w, flags, att_act, N, g_o, g_h, gl,
                                                  t1, tg);
    });
    PVS_CHECK_LAUNCH();
    return 0;
}

int pvs_node_tail_bwd1(hipStream_t s, const float* g_u, const float* y1, const float* stats,
                       const PvsNodeW& w, int N, int H, float* g_yn) {
    k_node_tail_bwd1<<<ew_grid((long long)N * H), 256, 0, s>>>(g_u, y1, stats, w, N, H, g_yn);
    PVS_CHECK_LAUNCH();
    return 0;
}

int pvs_graphnorm_bwd_coefs(hipStream_t s, const float* S1, const float* S2, const float* stats,
                            const PvsNodeW& w, int N, int H, float* g_w, float* g_b, float* g_ms,
                            float* coefs) {
    k_gn_bwd_coefs<<<(H + 63) / 64, 64, 0, s>>>(S1, S2, stats, w, N, H, g_w, g_b, g_ms, coefs);
    PVS_CHECK_LAUNCH();
    return 0;
}

int pvs_node_tail_bwd2(hipStream_t s, const float* g_yn, const float* y1, const float* stats,
                       const PvsNodeW& w, const float* coefs, int N, int H, float* g_y1) {
    k_node_tail_bwd2<<<ew_grid((long long)N * H), 256, 0, s>>>(g_yn, y1, stats, w, coefs, N, H, g_y1);
    PVS_CHECK_LAUNCH();
    return 0;
}

int pvs_prep_edge_bwd(hipStream_t s, const float* g_x_out, const float* inv_deg, const float* Magg,
                      const float* gM, int N, int H, float* gxagg, float* softD, float* zero_gPQ,
                      float* zero_gx_row) {
    if (!gxagg && !softD && !zero_gPQ && !zero_gx_row) return 0;
    const long long threads = (long long)N * (H / 4);
    k_prep_edge_bwd<<<(int)((threads + 255) / 256), 256, 0, s>>>(g_x_out, inv_deg, Magg, gM, N, H, gxagg, softD,
                                                                 zero_gPQ, zero_gx_row);
    PVS_CHECK_LAUNCH();
    return 0;
}

int pvs_copy_small(hipStream_t s, const float* src, float* dst, int n) {
    k_copy_small<<<(n + 255) / 256, 256, 0, s>>>(src, dst, n);
    PVS_CHECK_LAUNCH();
    return 0;
}

int pvs_sum_vec(hipStream_t s, const float* v, int n, float* out) {
    k_sum_vec<<<1, 64, 0, s>>>(v, n, out);
    PVS_CHECK_LAUNCH();
    return 0;
}


// ---- clip + Adam for every parameter tensor in one launch (include/pvs_egnn.h) ----
namespace {
// STEP_DEV: the step count is a float on the device (torch's capturable Adam keeps it there, so that a captured step can be
// replayed): every workgroup forms the two bias corrections from it - in double, as the host form does in Python.
// step_size = lr / (1 - beta1^step), bc2_sqrt = sqrt(1 - beta2^step), omb1 = 1 - beta1, omb2 = 1 - beta2: formed in DOUBLE
// and rounded once, as torch's Adam forms them (Python floats handed to lerp_ / addcmul_ / addcdiv_ as scalars: adam.py
// `_single_tensor_adam`). Until round 6 the kernel took 1 - beta in fp32 - 1.f - 0.999f is 1.3e-5 below float(0.001) - and
// exp_avg_sq ran 1.3e-5 (relative) below torch's, the updates 6e-6 (tools/fuzz_adam.py found it with gradients of 30).
template <bool STEP_DEV>
__global__ void __launch_bounds__(256)
k_adam_clip(const PvsAdamEntry* __restrict__ table, float step_size, float beta2, float omb1, float omb2, float eps, float wd,
            float bc2_sqrt, float clip, const float* __restrict__ step_dev, double lr_d, double beta1_d, double beta2_d) {
    if constexpr (STEP_DEV) {
        __shared__ float bc[2];
        if (threadIdx.x == 0) {      // (from the betas as the caller holds them - doubles in Python - like `1 - beta ** step` there)
            // (a counter below 1 - a state restored by hand, a step skipped - would make both corrections 0 and the
            // update a division by zero: treated as the first step)
            const double t = fmax((double)step_dev[0], 1.0);
            bc[0] = (float)(lr_d / (1.0 - pow(beta1_d, t)));
            bc[1] = (float)sqrt(1.0 - pow(beta2_d, t));
        }
        __syncthreads();
        step_size = bc[0];
        bc2_sqrt = bc[1];
    }
    const PvsAdamEntry e = table[blockIdx.y];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < e.numel;
         i += (long long)gridDim.x * blockDim.x) {
        float g = e.grad[i];
        if (clip > 0.f) {
            g = (g != g) ? g : fminf(fmaxf(g, -clip), clip);   // torch.clamp propagates NaN (fminf/fmaxf drop it)
            e.grad[i] = g;                       // clip_grad_value_ is in place
        }
        const float p = e.param[i];
        if (wd != 0.f) g = __fadd_rn(g, __fmul_rn(wd, p));                        // grad.add(param, alpha=wd)
        float m = e.exp_avg[i];
        m = __fadd_rn(m, __fmul_rn(omb1, __fsub_rn(g, m)));                      // lerp_(grad, 1 - beta1)
        float v = __fmul_rn(e.exp_avg_sq[i], beta2);
        v = __fadd_rn(v, __fmul_rn(__fmul_rn(omb2, g), g));                      // addcmul_(g, g, value = 1 - beta2)
        const float denom = __fadd_rn(__fdiv_rn(sqrtf(v), bc2_sqrt), eps);
        e.exp_avg[i] = m;
        e.exp_avg_sq[i] = v;
        e.param[i] = __fadd_rn(p, __fmul_rn(-step_size, __fdiv_rn(m, denom)));   // addcdiv_(m, denom, value=-step_size)
    }
}
}  // namespace

extern "C" int pvs_adam_clip_step(const PvsAdamEntry* table, int32_t n, double lr, double beta1, double beta2,
                                  float eps, float wd, double bc1, double bc2, float clip, pvs_stream_t stream) {
    PVS_REQUIRE(table && n >= 0, "pvs_adam_clip_step: bad arguments");
    PVS_REQUIRE(bc1 > 0.0 && bc2 > 0.0, "pvs_adam_clip_step: bias corrections must be positive (step >= 1)");
    if (n == 0) return 0;
    k_adam_clip<false><<<dim3(8, n), 256, 0, (hipStream_t)stream>>>(table, (float)(lr / bc1), (float)beta2, (float)(1.0 - beta1),
                                                                    (float)(1.0 - beta2), eps, wd, (float)sqrt(bc2), clip,
                                                                    nullptr, 0.0, 0.0, 0.0);
    PVS_CHECK_LAUNCH();
    return 0;
}

extern "C" int pvs_adam_clip_step_dev(const PvsAdamEntry* table, int32_t n, double lr, double beta1, double beta2,
                                      float eps, float wd, const float* step, float clip, pvs_stream_t stream) {
    PVS_REQUIRE(table && step && n >= 0, "pvs_adam_clip_step_dev: bad arguments");
    if (n == 0) return 0;
    k_adam_clip<true><<<dim3(8, n), 256, 0, (hipStream_t)stream>>>(table, 0.f, (float)beta2, (float)(1.0 - beta1),
                                                                   (float)(1.0 - beta2), eps, wd, 1.f, clip, step, lr, beta1, beta2);
    PVS_CHECK_LAUNCH();
    return 0;
}
