// Generic (any flag set, H in {16,32,64}) edge kernels: one wavefront walks one destination-row
// segment of the CSR; lane = (edge slot, channel); the per-edge MLPs are GEMVs with the weights in
// LDS and the edge's activation vector broadcast through a per-wave LDS scratch; the per-row sums
// (scatter-sum / scatter-mean of the reference) are lane-local accumulators, so there are no
// atomics and the result is bitwise reproducible.
//
// Reference semantics: EGNNLayer.coord2radial / edge_model / coord_model / node_model's
// aggregation, /root/reference/point_vs/models/geometric/egnn_satorras.py:123-187.
#include "edge_kernels.h"
#include "profile.h"

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;

// out = bias + sum_k Wt[k*H + c] * vec[k]   (vec: this edge slot's activation vector in LDS)
template <int H>
__device__ __forceinline__ float gemv_lds(const float* __restrict__ Wt, const float* __restrict__ vec,
                                          int c, float bias) {
    float acc = bias;
#pragma unroll
    for (int k = 0; k < H; k += 4) {
        float4 v = *reinterpret_cast<const float4*>(vec + k);
        acc = fmaf(Wt[(k + 0) * H + c], v.x, acc);
        acc = fmaf(Wt[(k + 1) * H + c], v.y, acc);
        acc = fmaf(Wt[(k + 2) * H + c], v.z, acc);
        acc = fmaf(Wt[(k + 3) * H + c], v.w, acc);
    }
    return acc;
}

template <int H>
__device__ __forceinline__ void load_wt(float* dst, const float* __restrict__ W, bool transpose) {
    // dst[k*H + c] = transpose ? W[c*H + k] : W[k*H + c]
    for (int i = threadIdx.x; i < H * H; i += kThreads) {
        int k = i / H, c = i % H;
        dst[i] = transpose ? W[c * H + k] : W[i];
    }
}

template <int H>
__global__ void __launch_bounds__(kThreads)
k_edge_fwd_v0(PvsGraph g, PvsEdgeW w, uint32_t flags, int att_act, PvsEdgeFwdIO io) {
    constexpr int EPW = 64 / H;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* W2t = smem;                    // [k][c]
    float* Wc1t = W2t + H * H;            // [k][c]
    float* WaT = Wc1t + H * H;            // [t][c]
    float* scratch = WaT + PVS_MAX_EDGE_ATTR * H;  // [waves][2][64]

    const bool upd = flags & PVS_UPDATE_COORDS;
    const bool eatt = flags & PVS_EDGE_ATTENTION;
    const bool soft = flags & PVS_SOFTMAX_ATT;
    const bool eres = (flags & PVS_EDGE_RESIDUAL) && io.m_prev != nullptr;

    load_wt<H>(W2t, w.w2, true);
    if (upd) load_wt<H>(Wc1t, w.wc1, true);
    for (int i = threadIdx.x; i < w.n_attr * H; i += kThreads) {
        int t = i / H, c = i % H;
        WaT[i] = w.w1[c * w.ld1 + w.off_rho + 1 + t];
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c = lane % H, sub = lane / H;
    float* vecA = scratch + wv * 128 + sub * H;
    float* vecB = scratch + wv * 128 + 64 + sub * H;

    const float b2c = w.b2[c];
    const float wrho = w.w1[c * w.ld1 + w.off_rho];
    const float bc1c = upd ? w.bc1[c] : 0.f;
    const float wc2c = upd ? w.wc2[c] : 0.f;
    const float wac = eatt ? w.wa[c] : 0.f;
    const float bac = eatt ? w.ba[0] : 0.f;
    float gate = 1.f;
    if (eres && (flags & (PVS_REZERO | PVS_GATED_RESIDUAL))) {
        gate = w.edge_gate[0];
        if (flags & PVS_GATED_RESIDUAL) gate = fmaxf(gate, 0.f);
    }

    const int total_waves = gridDim.x * kWaves;
    for (int i = blockIdx.x * kWaves + wv; i < g.n_nodes; i += total_waves) {
        const int e0 = g.rowptr[i], e1 = g.rowptr[i + 1];
        const float Pi = io.PQ[(size_t)i * 2 * H + c];
        const float xi0 = io.x[3 * i], xi1 = io.x[3 * i + 1], xi2 = io.x[3 * i + 2];
        float macc = 0.f, xa0 = 0.f, xa1 = 0.f, xa2 = 0.f;
        float smax = -INFINITY, ssum = 0.f;
        for (int eb = e0; eb < e1; eb += EPW) {
            const int e = eb + sub;
            const bool valid = e < e1;
            const int ee = valid ? e : e1 - 1;
            const int j = g.col[ee];
            const int t = w.n_attr ? (int)g.etype[ee] : 0;
            const float d0 = xi0 - io.x[3 * j], d1 = xi1 - io.x[3 * j + 1], d2 = xi2 - io.x[3 * j + 2];
            const float rho = d0 * d0 + d1 * d1 + d2 * d2;
            float z1 = Pi + io.PQ[(size_t)j * 2 * H + H + c] + wrho * rho;
            if (w.n_attr) z1 += WaT[t * H + c];
            vecA[c] = pvs_silu(z1);
            pvs_wave_lds_sync();
            float m = pvs_silu(gemv_lds<H>(W2t, vecA, c, b2c));
            if (eres) {
                const float mp = io.m_prev[(size_t)ee * H + c];
                if (flags & PVS_REZERO) m = mp + gate * m;
                else if (flags & PVS_GATED_RESIDUAL) m = gate * m + (1.f - gate) * mp;
                else m = m + mp;
            }
            if (io.m_out && valid) io.m_out[(size_t)e * H + c] = m;
            if (upd) {
                vecB[c] = m;
                pvs_wave_lds_sync();
                const float q = pvs_silu(gemv_lds<H>(Wc1t, vecB, c, bc1c));
                float s = pvs_group_sum<H>(wc2c * q);
                if (flags & PVS_TANH) s = pvs_tanh(s);
                if (flags & PVS_NORMALIZE) s = s / (sqrtf(rho) + 1e-8f);
                if (valid) { xa0 = fmaf(d0, s, xa0); xa1 = fmaf(d1, s, xa1); xa2 = fmaf(d2, s, xa2); }
            }
            if (eatt) {
                const float logit = pvs_group_sum<H>(wac * m) + bac;
                if (soft) {
                    if (valid) {
                        const float nm = fmaxf(smax, logit);
                        const float sc = __expf(smax - nm);   // exp(-inf)=0 on the first edge
                        const float wgt = __expf(logit - nm);
                        macc = fmaf(wgt, m, macc * sc);
                        ssum = fmaf(ssum, sc, wgt);
                        smax = nm;
                        if (c == 0) io.att_out[e] = logit;
                    }
                } else {
                    const float a = pvs_att_act(att_act, logit);
                    if (valid) {
                        macc = fmaf(a, m, macc);
                        if (c == 0) io.att_out[e] = a;
                    }
                }
            } else if (valid) {
                macc += m;
            }
        }
        // combine the EPW edge slots of the wave
        if (eatt && soft) {
#pragma unroll
            for (int o = H; o < 64; o <<= 1) {
                const float omax = __shfl_xor(smax, o, 64), osum = __shfl_xor(ssum, o, 64);
                const float oacc = __shfl_xor(macc, o, 64);
                const float nm = fmaxf(smax, omax);
                const float sa = (smax == -INFINITY) ? 0.f : __expf(smax - nm);
                const float sb = (omax == -INFINITY) ? 0.f : __expf(omax - nm);
                macc = macc * sa + oacc * sb;
                ssum = ssum * sa + osum * sb;
                smax = nm;
            }
            macc = ssum > 0.f ? macc / ssum : 0.f;
            if (lane == 0) { io.smax[i] = smax; io.ssum[i] = ssum; }
        } else {
#pragma unroll
            for (int o = H; o < 64; o <<= 1) macc += __shfl_xor(macc, o, 64);
        }
        if (sub == 0) io.Magg[(size_t)i * H + c] = macc;
        if (upd) {
#pragma unroll
            for (int o = H; o < 64; o <<= 1) {
                xa0 += __shfl_xor(xa0, o, 64);
                xa1 += __shfl_xor(xa1, o, 64);
                xa2 += __shfl_xor(xa2, o, 64);
            }
            if (lane == 0) {
                const float inv = g.inv_deg[i];
                io.x_out[3 * i] = xi0 + xa0 * inv;
                io.x_out[3 * i + 1] = xi1 + xa1 * inv;
                io.x_out[3 * i + 2] = xi2 + xa2 * inv;
            }
        }
    }
}

__global__ void k_softmax_finalize(PvsGraph g, const float* __restrict__ smax,
                                   const float* __restrict__ ssum, float* __restrict__ att) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= g.n_edges || (g.n_edges_dev && e >= *g.n_edges_dev)) return;
    int i = g.row[e];
    att[e] = __expf(att[e] - smax[i]) / ssum[i];
}

// ------------------------------------------------------------------------------------------------
template <int H>
__global__ void __launch_bounds__(kThreads)
k_edge_bwd_v0(PvsGraph g, PvsEdgeW w, uint32_t flags, int att_act, PvsEdgeBwdIO io) {
    constexpr int EPW = 64 / H;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* W2t = smem;                   // W2t[k*H+c] = W2[c][k]   (z2 = W2 a1)
    float* W2n = W2t + H * H;            // W2n[k*H+c] = W2[k][c]   (g_a1 = W2^T g_z2)
    float* Wc1t = W2n + H * H;
    float* Wc1n = Wc1t + H * H;
    float* WaT = Wc1n + H * H;
    float* scratch = WaT + PVS_MAX_EDGE_ATTR * H;  // [waves][3][64]

    const bool upd = (flags & PVS_UPDATE_COORDS) && io.gxagg != nullptr;
    const bool eatt = flags & PVS_EDGE_ATTENTION;
    const bool soft = flags & PVS_SOFTMAX_ATT;
    const bool eres = (flags & PVS_EDGE_RESIDUAL) && io.m_prev != nullptr;

    load_wt<H>(W2t, w.w2, true);
    load_wt<H>(W2n, w.w2, false);
    if (upd) { load_wt<H>(Wc1t, w.wc1, true); load_wt<H>(Wc1n, w.wc1, false); }
    for (int i = threadIdx.x; i < w.n_attr * H; i += kThreads) {
        int t = i / H, c = i % H;
        WaT[i] = w.w1[c * w.ld1 + w.off_rho + 1 + t];
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c = lane % H, sub = lane / H;
    float* vecA = scratch + wv * 192 + sub * H;        // a1
    float* vecB = scratch + wv * 192 + 64 + sub * H;   // m
    float* vecG = scratch + wv * 192 + 128 + sub * H;  // gradient vector being back-propagated

    const float b2c = w.b2[c];
    const float wrho = w.w1[c * w.ld1 + w.off_rho];
    const float bc1c = upd ? w.bc1[c] : 0.f;
    const float wc2c = upd ? w.wc2[c] : 0.f;
    const float wac = eatt ? w.wa[c] : 0.f;
    const float bac = eatt ? w.ba[0] : 0.f;
    float gate_raw = 0.f, gate = 1.f;
    if (eres && (flags & (PVS_REZERO | PVS_GATED_RESIDUAL))) {
        gate_raw = w.edge_gate[0];
        gate = (flags & PVS_GATED_RESIDUAL) ? fmaxf(gate_raw, 0.f) : gate_raw;
    }

    // lane-local weight-gradient accumulators: row c of each matrix, entry c of each vector
    float gW2r[H], gWc1r[H];
#pragma unroll
    for (int k = 0; k < H; ++k) { gW2r[k] = 0.f; gWc1r[k] = 0.f; }
    float g_b2 = 0.f, g_bc1 = 0.f, g_wc2 = 0.f, g_wa = 0.f, g_wrho = 0.f, g_ba = 0.f, g_gate = 0.f;
    float g_wattr[PVS_MAX_EDGE_ATTR];
#pragma unroll
    for (int t = 0; t < PVS_MAX_EDGE_ATTR; ++t) g_wattr[t] = 0.f;

    const int total_waves = gridDim.x * kWaves;
    for (int i = blockIdx.x * kWaves + wv; i < g.n_nodes; i += total_waves) {
        const int e0 = g.rowptr[i], e1 = g.rowptr[i + 1];
        const float Pi = io.PQ[(size_t)i * 2 * H + c];
        const float xi0 = io.x[3 * i], xi1 = io.x[3 * i + 1], xi2 = io.x[3 * i + 2];
        const float gMi = io.gM[(size_t)i * H + c];
        float gT0 = 0.f, gT1 = 0.f, gT2 = 0.f;
        if (upd) { gT0 = io.gxagg[3 * i]; gT1 = io.gxagg[3 * i + 1]; gT2 = io.gxagg[3 * i + 2]; }
        const float Di = (eatt && soft) ? io.softD[i] : 0.f;
        float gP = 0.f, gx0 = 0.f, gx1 = 0.f, gx2 = 0.f;
        for (int eb = e0; eb < e1; eb += EPW) {
            const int e = eb + sub;
            const bool valid = e < e1;
            const int ee = valid ? e : e1 - 1;
            const float vm = valid ? 1.f : 0.f;   // masks every accumulation of a padded slot
            const int j = g.col[ee];
            const int t = w.n_attr ? (int)g.etype[ee] : 0;
            const float d0 = xi0 - io.x[3 * j], d1 = xi1 - io.x[3 * j + 1], d2 = xi2 - io.x[3 * j + 2];
            const float rho = d0 * d0 + d1 * d1 + d2 * d2;
            // ---- recompute the forward of this edge ----
            float z1 = Pi + io.PQ[(size_t)j * 2 * H + H + c] + wrho * rho;
            if (w.n_attr) z1 += WaT[t * H + c];
            const float s1 = pvs_sigmoid(z1);
            const float a1 = z1 * s1;
            vecA[c] = a1;
            pvs_wave_lds_sync();
            const float z2 = gemv_lds<H>(W2t, vecA, c, b2c);
            const float s2 = pvs_sigmoid(z2);
            const float m_new = z2 * s2;
            float m = m_new, mp = 0.f;
            if (eres) {
                mp = io.m_prev[(size_t)ee * H + c];
                if (flags & PVS_REZERO) m = mp + gate * m_new;
                else if (flags & PVS_GATED_RESIDUAL) m = gate * m_new + (1.f - gate) * mp;
                else m = m_new + mp;
            }
            vecB[c] = m;
            pvs_wave_lds_sync();
            // ---- gradient wrt m ----
            float gm = (io.g_m_out && valid) ? io.g_m_out[(size_t)ee * H + c] : 0.f;
            float s_coord = 0.f, nrm = 1.f;
            if (upd) {
                const float zc = gemv_lds<H>(Wc1t, vecB, c, bc1c);
                const float sc = pvs_sigmoid(zc);
                const float q = zc * sc;
                float s = pvs_group_sum<H>(wc2c * q);
                float dact = 1.f;
                if (flags & PVS_TANH) { s = pvs_tanh(s); dact = 1.f - s * s; }
                if (flags & PVS_NORMALIZE) nrm = 1.f / (sqrtf(rho) + 1e-8f);
                s_coord = s;
                // trans = d * nrm * s ; g_s = (d*nrm) . gT
                const float g_s = (d0 * gT0 + d1 * gT1 + d2 * gT2) * nrm * dact * vm;
                const float g_zc = g_s * wc2c * pvs_silu_grad(zc, sc);
                g_wc2 = fmaf(g_s, q, g_wc2);
                g_bc1 += g_zc;
                vecG[c] = g_zc;
                pvs_wave_lds_sync();
                gm += gemv_lds<H>(Wc1n, vecG, c, 0.f);
#pragma unroll
                for (int k = 0; k < H; k += 4) {
                    float4 v = *reinterpret_cast<const float4*>(vecB + k);
                    gWc1r[k] = fmaf(g_zc, v.x, gWc1r[k]);
                    gWc1r[k + 1] = fmaf(g_zc, v.y, gWc1r[k + 1]);
                    gWc1r[k + 2] = fmaf(g_zc, v.z, gWc1r[k + 2]);
                    gWc1r[k + 3] = fmaf(g_zc, v.w, gWc1r[k + 3]);
                }
            }
            if (eatt) {
                const float logit = pvs_group_sum<H>(wac * m) + bac;
                const float a = io.att[ee];
                const float dot = pvs_group_sum<H>(m * gMi);
                float g_l;
                if (soft) g_l = a * (dot - Di);
                else g_l = pvs_att_act_grad(att_act, logit, a) * dot;
                g_l *= vm;
                gm += (a * gMi) * vm + g_l * wac;
                g_wa = fmaf(g_l, m, g_wa);
                g_ba += g_l;
            } else {
                gm += gMi * vm;
            }
            // ---- edge residual ----
            float gm_new = gm;
            if (eres) {
                if (flags & PVS_REZERO) {
                    gm_new = gate * gm;
                    g_gate = fmaf(gm, m_new, g_gate);
                    if (valid) io.g_m_prev[(size_t)e * H + c] = gm;
                } else if (flags & PVS_GATED_RESIDUAL) {
                    gm_new = gate * gm;
                    if (gate_raw > 0.f) g_gate = fmaf(gm, m_new - mp, g_gate);
                    if (valid) io.g_m_prev[(size_t)e * H + c] = (1.f - gate) * gm;
                } else {
                    if (valid) io.g_m_prev[(size_t)e * H + c] = gm;
                }
            }
            // ---- second edge-MLP layer ----
            const float g_z2 = gm_new * pvs_silu_grad(z2, s2);
            g_b2 += g_z2;
            pvs_wave_lds_sync();   // everyone is done reading vecG (coord path) before reuse
            vecG[c] = g_z2;
            pvs_wave_lds_sync();
            const float g_a1 = gemv_lds<H>(W2n, vecG, c, 0.f);
#pragma unroll
            for (int k = 0; k < H; k += 4) {
                float4 v = *reinterpret_cast<const float4*>(vecA + k);
                gW2r[k] = fmaf(g_z2, v.x, gW2r[k]);
                gW2r[k + 1] = fmaf(g_z2, v.y, gW2r[k + 1]);
                gW2r[k + 2] = fmaf(g_z2, v.z, gW2r[k + 2]);
                gW2r[k + 3] = fmaf(g_z2, v.w, gW2r[k + 3]);
            }
            // ---- first edge-MLP layer (its matrix part lives at node level: P and Q) ----
            const float g_z1 = g_a1 * pvs_silu_grad(z1, s1);   // already 0 for padded slots
            if (valid) io.gz1[(size_t)e * H + c] = g_z1;
            gP += g_z1;
            g_wrho = fmaf(g_z1, rho, g_wrho);
#pragma unroll
            for (int tt = 0; tt < PVS_MAX_EDGE_ATTR; ++tt)
                if (tt == t && w.n_attr) g_wattr[tt] += g_z1;
            const float g_rho = pvs_group_sum<H>(wrho * g_z1);
            // ---- coordinates: trans = d*nrm*s (nrm detached), rho = |d|^2 ----
            const float k1 = s_coord * nrm * vm;
            const float gd0 = fmaf(k1, gT0, 2.f * d0 * g_rho);
            const float gd1 = fmaf(k1, gT1, 2.f * d1 * g_rho);
            const float gd2 = fmaf(k1, gT2, 2.f * d2 * g_rho);
            if (valid && c == 0)
                *reinterpret_cast<float4*>(io.gd + (size_t)e * 4) =
                    make_float4(gd0, gd1, gd2, pvs_pack_rho_type(rho, t));
            gx0 += gd0; gx1 += gd1; gx2 += gd2;
        }
#pragma unroll
        for (int o = H; o < 64; o <<= 1) {
            gP += __shfl_xor(gP, o, 64);
            gx0 += __shfl_xor(gx0, o, 64);
            gx1 += __shfl_xor(gx1, o, 64);
            gx2 += __shfl_xor(gx2, o, 64);
        }
        if (sub == 0) io.gPQ[(size_t)i * 2 * H + c] = gP;
        if (lane == 0) {
            io.gx_row[3 * i] = gx0;
            io.gx_row[3 * i + 1] = gx1;
            io.gx_row[3 * i + 2] = gx2;
        }
    }

    // ---- block reduction of the weight-gradient accumulators, in a fixed order ----
    const PvsSlabLayout L = pvs_slab_layout(H);
    __syncthreads();
    float* slab = smem;   // the weights are no longer needed
    for (int i = threadIdx.x; i < L.total; i += kThreads) slab[i] = 0.f;
    __syncthreads();
    for (int turn = 0; turn < kWaves * EPW; ++turn) {
        if (wv * EPW + sub == turn) {
#pragma unroll
            for (int k = 0; k < H; ++k) {
                slab[L.w2 + c * H + k] += gW2r[k];
                slab[L.wc1 + c * H + k] += gWc1r[k];
            }
            slab[L.b2 + c] += g_b2;
            slab[L.bc1 + c] += g_bc1;
            slab[L.wc2 + c] += g_wc2;
            slab[L.wa + c] += g_wa;
            slab[L.wrho + c] += g_wrho;
#pragma unroll
            for (int tt = 0; tt < PVS_MAX_EDGE_ATTR; ++tt) slab[L.wattr + tt * H + c] += g_wattr[tt];
            if (c == 0) slab[L.ba] += g_ba;
        }
        __syncthreads();
    }
    // gate gradient: sum over channels too (fixed order: lane-group reduce then turns)
    {
        float gsum = pvs_group_sum<H>(g_gate);
        for (int turn = 0; turn < kWaves * EPW; ++turn) {
            if (wv * EPW + sub == turn && c == 0) slab[L.gate] += gsum;
            __syncthreads();
        }
    }
    float* dst = io.slabs + (size_t)blockIdx.x * L.total;
    for (int i = threadIdx.x; i < L.total; i += kThreads) dst[i] = slab[i];
}

// Column-side gather: per node n,  g_Q[n] = sum_{e: col=n} gz1[e],  gx_col = sum gd[e]  over the CSC
// list;  g_x[n] = g_x_out[n] + gx_row[n] - gx_col.  With WSUMS it also accumulates, over the same
// rows (every edge is in exactly one CSC list), the per-edge-scalar weight gradients
// g_wrho = sum gz1[e]*rho_e (rho in gd4[e].w) and g_wattr[t] = sum_{type_e=t} gz1[e] into per-block
// slabs [4H] - the MFMA edge backward leaves those to this memory-bound kernel, whose VALU is idle.
// Lane = (edge slot, 16-byte quad): one wave-instruction fetches 64/(H/4) whole rows; UN independent
// batches in flight. Fixed order everywhere: bitwise reproducible.
#ifndef PVS_NG_MINBLOCKS
#define PVS_NG_MINBLOCKS 5      // workgroups per CU the register budget must allow: 20 waves per CU with UN = 6
#endif
template <int H, bool WSUMS>
__global__ void __launch_bounds__(kThreads, PVS_NG_MINBLOCKS)
k_node_gather(PvsGraph g, const float* __restrict__ gz1, const float* __restrict__ gd4,
              const float* __restrict__ gx_row, const float* __restrict__ g_x_out,
              float* __restrict__ gPQ, float* __restrict__ g_x, float* __restrict__ slabs,
              int n_lo, int n_hi) {
    constexpr int QPR = H / 4;
    constexpr int EPW = 64 / QPR;
#ifndef PVS_NG_UN
#define PVS_NG_UN 6      // (8 at four waves per SIMD before round 4: the rate follows the waves in flight - five waves of six
#endif                   // batches each: cfg3 gather -2.4 %, cfg2 -1.5 %; 8 waves of 4 spill and lose 5-8 %: profiles/r04_ab_gather_rows_in_flight.txt)
    constexpr int UN = PVS_NG_UN;
    __shared__ float red[kWaves][4 * H];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int quad = lane % QPR, sub = lane / QPR;
    const int total_waves = gridDim.x * kWaves;
    float4 wrho = make_float4(0.f, 0.f, 0.f, 0.f), wat0 = wrho, wat1 = wrho, wat2 = wrho;
    auto add4 = [](float4& a, const float4& v) { a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; };
    auto fma4 = [](float4& a, const float4& v, float s) {
        a.x = fmaf(v.x, s, a.x); a.y = fmaf(v.y, s, a.y); a.z = fmaf(v.z, s, a.z); a.w = fmaf(v.w, s, a.w);
    };
    // Column -> wave map. The 16-byte per-edge records (gd4) are read at scattered positions: a read
    // fetches a whole 128-byte line = the records of 8 CONSECUTIVE sorted edges, i.e. 8 neighbours of
    // one row with ascending column ids (TCC request counters: these lines cost as many bytes as the
    // gradient rows themselves when every column pulls its own copy through the fabric). So each XCD
    // (blocks b, b + 8, ... share one, MI355X_MICROARCH.md) takes a CONTIGUOUS range of columns and
    // its resident waves sweep it as a window of adjacent columns: the 8 columns that need a line are
    // then in flight on the same L2 at about the same time. Speed only - any map is correct.
    int n_first, n_stride, n_stop;
    if ((gridDim.x & 7) == 0) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int per_xcd = (n_hi - n_lo + 7) / 8;
        n_first = n_lo + xcd * per_xcd + slot * kWaves + wv;
        n_stride = (gridDim.x >> 3) * kWaves;
        n_stop = min(n_hi, n_lo + (xcd + 1) * per_xcd);
    } else {
        n_first = n_lo + blockIdx.x * kWaves + wv;
        n_stride = total_waves;
        n_stop = n_hi;
    }
    for (int n = n_first; n < n_stop; n += n_stride) {
        float4 accc = make_float4(0.f, 0.f, 0.f, 0.f);
        float axc = 0.f;     // lanes with quad < 3 own coordinate component `quad`
        const int p0 = g.colptr[n], p1 = g.colptr[n + 1];
        for (int pb = p0; pb < p1; pb += EPW * UN) {
            int idx[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int p = pb + u * EPW + sub;
                idx[u] = p < p1 ? g.cedge[p] : -1;
            }
            float4 v[UN], d[UN];
            int ty[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const bool ok = idx[u] >= 0;
#ifdef PVS_ABL_NG_RESIDENT      // timing-only: every row from a 4 MB window (32k edges) that stays in the caches
                if (ok) idx[u] &= 32767;
#endif
                v[u] = ok ? *reinterpret_cast<const float4*>(gz1 + (size_t)idx[u] * H + 4 * quad)
                          : make_float4(0.f, 0.f, 0.f, 0.f);
                d[u] = ok ? *reinterpret_cast<const float4*>(gd4 + (size_t)idx[u] * 4)
                          : make_float4(0.f, 0.f, 0.f, 0.f);
                ty[u] = -1;
                if (WSUMS) {
                    int tt;
                    d[u].w = pvs_unpack_rho(d[u].w, &tt);
                    ty[u] = (ok && g.etype) ? tt : -1;
                }
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                add4(accc, v[u]);
                axc += quad == 0 ? d[u].x : (quad == 1 ? d[u].y : (quad == 2 ? d[u].z : 0.f));
                if (WSUMS) {
                    fma4(wrho, v[u], d[u].w);
                    fma4(wat0, v[u], ty[u] == 0 ? 1.f : 0.f);
                    fma4(wat1, v[u], ty[u] == 1 ? 1.f : 0.f);
                    fma4(wat2, v[u], ty[u] == 2 ? 1.f : 0.f);
                }
            }
        }
#pragma unroll
        for (int o = QPR; o < 64; o <<= 1) {
            accc.x += __shfl_xor(accc.x, o, 64); accc.y += __shfl_xor(accc.y, o, 64);
            accc.z += __shfl_xor(accc.z, o, 64); accc.w += __shfl_xor(accc.w, o, 64);
            axc += __shfl_xor(axc, o, 64);
        }
        if (sub == 0) {
            *reinterpret_cast<float4*>(gPQ + (size_t)n * 2 * H + H + 4 * quad) = accc;
            if (g_x && quad < 3) {
                const float base = g_x_out ? g_x_out[3 * n + quad] : 0.f;
                g_x[3 * n + quad] = base + gx_row[3 * n + quad] - axc;
            }
        }
    }
    if (WSUMS) {   // per-block partial of [g_wrho | g_wattr0 | g_wattr1 | g_wattr2], fixed order
        auto reduce_subs = [&](float4& a) {
#pragma unroll
            for (int o = QPR; o < 64; o <<= 1) {
                a.x += __shfl_xor(a.x, o, 64); a.y += __shfl_xor(a.y, o, 64);
                a.z += __shfl_xor(a.z, o, 64); a.w += __shfl_xor(a.w, o, 64);
            }
        };
        reduce_subs(wrho); reduce_subs(wat0); reduce_subs(wat1); reduce_subs(wat2);
        if (sub == 0) {
            *reinterpret_cast<float4*>(&red[wv][0 * H + 4 * quad]) = wrho;
            *reinterpret_cast<float4*>(&red[wv][1 * H + 4 * quad]) = wat0;
            *reinterpret_cast<float4*>(&red[wv][2 * H + 4 * quad]) = wat1;
            *reinterpret_cast<float4*>(&red[wv][3 * H + 4 * quad]) = wat2;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 4 * H; i += kThreads) {
            float t = 0.f;
#pragma unroll
            for (int ww = 0; ww < kWaves; ++ww) t += red[ww][i];
            slabs[(size_t)blockIdx.x * 4 * H + i] = t;
        }
    }
}

template <typename K>
int set_lds(K kernel, size_t lds) {
    if (lds > 48 * 1024)
        PVS_CHECK_HIP(hipFuncSetAttribute((const void*)kernel,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return 0;
}

}  // namespace

int pvs_edge_v0_supported(int H) { return H == 16 || H == 32 || H == 64; }

int pvs_edge_v0_blocks(int N) {
    int b = (N + kWaves - 1) / kWaves;
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;
    return b;
}

#define PVS_DISPATCH_H(H, ...)                                         \
    switch (H) {                                                       \
        case 16: { constexpr int HH = 16; __VA_ARGS__; break; }               \
        case 32: { constexpr int HH = 32; __VA_ARGS__; break; }               \
        case 64: { constexpr int HH = 64; __VA_ARGS__; break; }               \
        default: pvs_set_error("edge kernels: hidden size %d unsupported (16,32,64)", H); return -1; \
    }

int pvs_launch_edge_fwd_v0(hipStream_t s, int H, const PvsGraph& g, const PvsEdgeW& w, uint32_t flags,
                           int att_act, const PvsEdgeFwdIO& io) {
    PVS_REQUIRE(w.n_attr <= PVS_MAX_EDGE_ATTR, "edge_attr classes %d > %d", w.n_attr,
                PVS_MAX_EDGE_ATTR);
    const int blocks = pvs_edge_v0_blocks(g.n_nodes);
    PvsProfScope prof(s, PVS_PROF_EDGE_FWD);
    PVS_DISPATCH_H(H, {
        size_t lds = (size_t)(2 * HH * HH + PVS_MAX_EDGE_ATTR * HH + kWaves * 128) * sizeof(float);
        if (set_lds(k_edge_fwd_v0<HH>, lds)) return -2;
        k_edge_fwd_v0<HH><<<blocks, kThreads, lds, s>>>(g, w, flags, att_act, io);
    });
    PVS_CHECK_LAUNCH();
    if ((flags & PVS_EDGE_ATTENTION) && (flags & PVS_SOFTMAX_ATT))
        return pvs_launch_softmax_finalize(s, g, io.smax, io.ssum, io.att_out);
    return 0;
}

int pvs_launch_softmax_finalize(hipStream_t s, const PvsGraph& g, const float* smax, const float* ssum,
                                float* att) {
    if (g.n_edges <= 0) return 0;
    k_softmax_finalize<<<(g.n_edges + 255) / 256, 256, 0, s>>>(g, smax, ssum, att);
    PVS_CHECK_LAUNCH();
    return 0;
}

int pvs_launch_edge_bwd_v0(hipStream_t s, int H, const PvsGraph& g, const PvsEdgeW& w, uint32_t flags,
                           int att_act, const PvsEdgeBwdIO& io, int* n_slabs) {
    PVS_REQUIRE(w.n_attr <= PVS_MAX_EDGE_ATTR, "edge_attr classes %d > %d", w.n_attr,
                PVS_MAX_EDGE_ATTR);
    int blocks = pvs_edge_v0_blocks(g.n_nodes);
    if (blocks > 512) blocks = 512;
    *n_slabs = blocks;
    PvsProfScope prof(s, PVS_PROF_EDGE_BWD);
    PVS_DISPATCH_H(H, {
        const PvsSlabLayout L = pvs_slab_layout(HH);
        size_t words = (size_t)(4 * HH * HH + PVS_MAX_EDGE_ATTR * HH + kWaves * 192);
        if (words < (size_t)L.total) words = L.total;
        size_t lds = words * sizeof(float);
        if (set_lds(k_edge_bwd_v0<HH>, lds)) return -2;
        k_edge_bwd_v0<HH><<<blocks, kThreads, lds, s>>>(g, w, flags, att_act, io);
    });
    PVS_CHECK_LAUNCH();
    return 0;
}

int pvs_node_gather_blocks(int N) {
    int b = pvs_edge_v0_blocks(N);
#ifndef PVS_NG_BLOCKS
#define PVS_NG_BLOCKS 1280   // 512 -> 1024: cfg3 gather -18 %, cfg2 -3 %; 1280 = five resident workgroups per CU (PVS_NG_MINBLOCKS)
#endif
    return b > PVS_NG_BLOCKS ? PVS_NG_BLOCKS : b;
}

// Column gather over the node range [n_lo, n_hi) (whole batch or one segment of whole graphs).
int pvs_launch_node_gather(hipStream_t s, int H, const PvsGraph& g, bool wsums, const float* gz1,
                           const float* gd4, const float* gx_row, const float* g_x_out, float* gPQ,
                           float* g_x, float* slabs, int n_lo, int n_hi, int* n_slabs) {
    *n_slabs = 0;
    if (n_hi <= n_lo) return 0;
    const int blocks = wsums ? pvs_node_gather_blocks(n_hi - n_lo) : pvs_edge_v0_blocks(n_hi - n_lo);
    *n_slabs = wsums ? blocks : 0;
    PvsProfScope prof(s, PVS_PROF_COL_GATHER);
    if (H == 128) {      // (the wide layer: only the column gather of this file is built at 128 channels)
        if (wsums)
            k_node_gather<128, true><<<blocks, kThreads, 0, s>>>(g, gz1, gd4, gx_row, g_x_out, gPQ, g_x, slabs, n_lo, n_hi);
        else
            k_node_gather<128, false><<<blocks, kThreads, 0, s>>>(g, gz1, gd4, gx_row, g_x_out, gPQ, g_x, slabs, n_lo, n_hi);
        PVS_CHECK_LAUNCH();
        return 0;
    }
    PVS_DISPATCH_H(H, {
        if (wsums)
            k_node_gather<HH, true><<<blocks, kThreads, 0, s>>>(g, gz1, gd4, gx_row, g_x_out, gPQ, g_x, slabs,
                                                                n_lo, n_hi);
        else
            k_node_gather<HH, false><<<blocks, kThreads, 0, s>>>(g, gz1, gd4, gx_row, g_x_out, gPQ, g_x, slabs,
                                                                 n_lo, n_hi);
    });
    PVS_CHECK_LAUNCH();
    return 0;
}
