// MFMA edge kernels for H = 32*HB (HB = 1, 2): the fast path of EGNNLayer's per-edge work.
//
// Tile = 32 consecutive CSR-sorted edges per wavefront. Activations live in the "X layout" of
// v_mfma_f32_32x32x2_f32 accumulators: lane l = (edge slot j = l&31, half hh = l>>5), register t
// of channel block b holds channel 32b + (t&3) + 8(t>>2) + 4hh. An accumulator in that layout is
// directly the B operand of the next product over channels (k pairs {ch(t,0), ch(t,1)}), so the
// edge-MLP chain  z1 -> SiLU -> W2 -> SiLU -> Wc1 -> SiLU  needs no lane movement; the weights
// are staged once per workgroup in LDS in A-operand order (one ds_read_b32 per MFMA).
// The first edge-MLP layer is algebraically split per node (P_i + Q_j + w_rho*rho + W_a[type]),
// so the per-edge MFMA work is the HxH products only.
// Per-row sums (the reference's scatter-sum / scatter-mean, egnn_satorras.py:332-347): each wave
// owns a row-aligned, edge-balanced chunk of the CSR; a tile's weighted messages go through a
// per-wave LDS tile and are re-read channel-per-lane, where segment boundaries are wave-uniform
// scalars: no atomics, fixed summation order, bitwise reproducible.
//
// Layout maps validated lane-by-lane in tools/mfma_layout_check.py.
#include "edge_kernels.h"
#include "profile.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256;
constexpr int kWaves = 4;
constexpr int kTile = 32;

__device__ __forceinline__ int xch(int t, int hh) { return (t & 3) + 8 * (t >> 2) + 4 * hh; }

// Stage W[H][H] (row-major, W[out][in]) for  Z = W V  (transpose=false)  or  Z = W^T V  (true)
// in A-operand order: dst[((bo*HB + bi)*16 + t)*64 + l] = Wx[32bo + (l&31)][32bi + ch(t, l>>5)].
template <int HB>
__device__ __forceinline__ void stage_weights(float* dst, const float* __restrict__ W, bool transpose) {
    constexpr int H = 32 * HB;
    for (int i = threadIdx.x; i < H * H; i += kThreads) {
        const int l = i & 63, t = (i >> 6) & 15, bb = i >> 10;
        const int bi = bb % HB, bo = bb / HB;
        const int o = 32 * bo + (l & 31), k = 32 * bi + xch(t, l >> 5);
        dst[i] = transpose ? W[k * H + o] : W[o * H + k];
    }
}

// acc[bo] += sum over (bi,t) of A-staged weights x v[bi][t]   (v in X layout)
template <int HB>
__device__ __forceinline__ void mfma_chain(const float* __restrict__ Ws, int lane,
                                           const float (&v)[HB][16], f32x16 (&acc)[HB]) {
#pragma unroll
    for (int bo = 0; bo < HB; ++bo)
#pragma unroll
        for (int bi = 0; bi < HB; ++bi)
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const float a = Ws[((bo * HB + bi) * 16 + t) * 64 + lane];
                acc[bo] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, v[bi][t], acc[bo], 0, 0, 0);
            }
}

// value of a per-channel table at this lane's X-layout channels: out[b][4g+q] = tab[32b+8g+4hh+q]
template <int HB>
__device__ __forceinline__ void load_tab(const float* __restrict__ tab, int hh, float (&out)[HB][16]) {
#pragma unroll
    for (int b = 0; b < HB; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 v = *reinterpret_cast<const float4*>(tab + 32 * b + 8 * g + 4 * hh);
            out[b][4 * g] = v.x; out[b][4 * g + 1] = v.y; out[b][4 * g + 2] = v.z; out[b][4 * g + 3] = v.w;
        }
}

template <int HB>
__device__ __forceinline__ float dot_tab(const float* __restrict__ tab, int hh, const float (&v)[HB][16]) {
    float s = 0.f;
#pragma unroll
    for (int b = 0; b < HB; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 w = *reinterpret_cast<const float4*>(tab + 32 * b + 8 * g + 4 * hh);
            s = fmaf(w.x, v[b][4 * g], s); s = fmaf(w.y, v[b][4 * g + 1], s);
            s = fmaf(w.z, v[b][4 * g + 2], s); s = fmaf(w.w, v[b][4 * g + 3], s);
        }
    return s + __shfl_xor(s, 32, 64);   // other half holds the other 16 channels of each block
}

__device__ __forceinline__ int chunk_begin(const PvsGraph& g, int k, int n_chunks) {
    if (k <= 0) return 0;
    if (k >= n_chunks) return g.n_edges;
    const long long t = (long long)k * g.n_edges / n_chunks;
    return g.rowptr[g.row[t]];   // start of the row that contains edge t: chunks are row-aligned
}

template <int HB>
__global__ void __launch_bounds__(kThreads)
k_edge_fwd_mfma(PvsGraph g, PvsEdgeW w, uint32_t flags, int att_act, PvsEdgeFwdIO io, int n_chunks) {
    constexpr int H = 32 * HB;
    constexpr int TS = H + 4;   // tile row stride (floats): conflict-free b128 writes / b32 reads
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* W2s = smem;
    float* Wc1s = W2s + H * H;
    float* b2t = Wc1s + H * H;
    float* bc1t = b2t + H;
    float* wc2t = bc1t + H;
    float* wat = wc2t + H;
    float* wrhot = wat + H;
    float* attrt = wrhot + H;                            // [PVS_MAX_EDGE_ATTR][H]
    float* wave_base = attrt + PVS_MAX_EDGE_ATTR * H;    // per wave: tile[32][TS], tx[32][4], rowbuf[32]
    constexpr int kWaveFloats = kTile * TS + kTile * 4 + kTile;

    const bool upd = flags & PVS_UPDATE_COORDS;
    const bool eatt = flags & PVS_EDGE_ATTENTION;
    const bool eres = (flags & PVS_EDGE_RESIDUAL) && io.m_prev != nullptr;

    stage_weights<HB>(W2s, w.w2, false);
    if (upd) stage_weights<HB>(Wc1s, w.wc1, false);
    for (int c = threadIdx.x; c < H; c += kThreads) {
        b2t[c] = w.b2[c];
        bc1t[c] = upd ? w.bc1[c] : 0.f;
        wc2t[c] = upd ? w.wc2[c] : 0.f;
        wat[c] = eatt ? w.wa[c] : 0.f;
        wrhot[c] = w.w1[c * w.ld1 + w.off_rho];
        for (int t = 0; t < PVS_MAX_EDGE_ATTR; ++t)
            attrt[t * H + c] = t < w.n_attr ? w.w1[c * w.ld1 + w.off_rho + 1 + t] : 0.f;
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int j = lane & 31, hh = lane >> 5;
    float* tile = wave_base + wv * kWaveFloats;
    float* tx = tile + kTile * TS;
    int* rowbuf = reinterpret_cast<int*>(tx + kTile * 4);
    const float bac = eatt ? w.ba[0] : 0.f;
    float gate = 1.f;
    if (eres && (flags & (PVS_REZERO | PVS_GATED_RESIDUAL))) {
        gate = w.edge_gate[0];
        if (flags & PVS_GATED_RESIDUAL) gate = fmaxf(gate, 0.f);
    }

    const int total_waves = gridDim.x * kWaves;
    for (int chunk = blockIdx.x * kWaves + wv; chunk < n_chunks; chunk += total_waves) {
        const int e_begin = chunk_begin(g, chunk, n_chunks);
        const int e_end = chunk_begin(g, chunk + 1, n_chunks);
        int cur_row = -1;
        float acc[HB], accx = 0.f;   // Y-phase: lane = channel 32b + j (and coordinate j < 3), parity hh
#pragma unroll
        for (int b = 0; b < HB; ++b) acc[b] = 0.f;

        auto flush = [&](int row_id) {
            if (row_id < 0) return;
#pragma unroll
            for (int b = 0; b < HB; ++b) {
                const float tot = acc[b] + __shfl_xor(acc[b], 32, 64);
                if (hh == 0) io.Magg[(size_t)row_id * H + 32 * b + j] = tot;
                acc[b] = 0.f;
            }
            if (upd) {
                const float totx = accx + __shfl_xor(accx, 32, 64);
                if (hh == 0 && j < 3)
                    io.x_out[3 * row_id + j] = io.x[3 * row_id + j] + totx * g.inv_deg[row_id];
                accx = 0.f;
            }
        };

        for (int e0 = e_begin; e0 < e_end; e0 += kTile) {
            const int e = e0 + j;
            const bool valid = e < e_end;
            const int ee = valid ? e : e_end - 1;
            const int i = g.row[ee], jn = g.col[ee];
            const int ty = w.n_attr ? (int)g.etype[ee] : 0;
            const int prev_row = (ee == e_begin) ? -1 : g.row[ee - 1];
            const unsigned long long ball = __ballot(valid && hh == 0 && i != prev_row);
            const unsigned bmask = (unsigned)ball;
            const float d0 = io.x[3 * i] - io.x[3 * jn], d1 = io.x[3 * i + 1] - io.x[3 * jn + 1];
            const float d2 = io.x[3 * i + 2] - io.x[3 * jn + 2];
            const float rho = d0 * d0 + d1 * d1 + d2 * d2;

            // ---- first layer: z1 = P_i + Q_j + w_rho*rho + W_a[type]; a1 = SiLU(z1) ----
            float a1[HB][16];
            {
                const float* Pp = io.PQ + (size_t)i * 2 * H + 4 * hh;
                const float* Qp = io.PQ + (size_t)jn * 2 * H + H + 4 * hh;
                const float* At = attrt + ty * H + 4 * hh;
                const float* Rt = wrhot + 4 * hh;
#pragma unroll
                for (int b = 0; b < HB; ++b)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const int off = 32 * b + 8 * gq;
                        const float4 p = *reinterpret_cast<const float4*>(Pp + off);
                        const float4 q = *reinterpret_cast<const float4*>(Qp + off);
                        const float4 a = *reinterpret_cast<const float4*>(At + off);
                        const float4 r = *reinterpret_cast<const float4*>(Rt + off);
                        a1[b][4 * gq] = pvs_silu(p.x + q.x + fmaf(r.x, rho, a.x));
                        a1[b][4 * gq + 1] = pvs_silu(p.y + q.y + fmaf(r.y, rho, a.y));
                        a1[b][4 * gq + 2] = pvs_silu(p.z + q.z + fmaf(r.z, rho, a.z));
                        a1[b][4 * gq + 3] = pvs_silu(p.w + q.w + fmaf(r.w, rho, a.w));
                    }
            }
            // ---- second layer on the matrix cores: m = SiLU(W2 a1 + b2) ----
            float m[HB][16];
            {
                f32x16 acc2[HB];
                float bias[HB][16];
                load_tab<HB>(b2t, hh, bias);
#pragma unroll
                for (int b = 0; b < HB; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc2[b][r] = bias[b][r];
                mfma_chain<HB>(W2s, lane, a1, acc2);
#pragma unroll
                for (int b = 0; b < HB; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) m[b][r] = pvs_silu(acc2[b][r]);
            }
            if (eres) {
#pragma unroll
                for (int b = 0; b < HB; ++b)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const float4 mp = *reinterpret_cast<const float4*>(
                            io.m_prev + (size_t)ee * H + 32 * b + 8 * gq + 4 * hh);
                        const float mpv[4] = {mp.x, mp.y, mp.z, mp.w};
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            float& mv = m[b][4 * gq + q];
                            if (flags & PVS_REZERO) mv = mpv[q] + gate * mv;
                            else if (flags & PVS_GATED_RESIDUAL) mv = gate * mv + (1.f - gate) * mpv[q];
                            else mv = mv + mpv[q];
                        }
                    }
            }
            if (io.m_out && valid) {
#pragma unroll
                for (int b = 0; b < HB; ++b)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq)
                        *reinterpret_cast<float4*>(io.m_out + (size_t)e * H + 32 * b + 8 * gq + 4 * hh) =
                            make_float4(m[b][4 * gq], m[b][4 * gq + 1], m[b][4 * gq + 2], m[b][4 * gq + 3]);
            }
            // ---- coordinate branch: s = wc2 . SiLU(Wc1 m + bc1) ----
            float s = 0.f;
            if (upd) {
                f32x16 accc[HB];
                float bias[HB][16];
                load_tab<HB>(bc1t, hh, bias);
#pragma unroll
                for (int b = 0; b < HB; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) accc[b][r] = bias[b][r];
                mfma_chain<HB>(Wc1s, lane, m, accc);
                float q[HB][16];
#pragma unroll
                for (int b = 0; b < HB; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) q[b][r] = pvs_silu(accc[b][r]);
                s = dot_tab<HB>(wc2t, hh, q);
                if (flags & PVS_TANH) s = pvs_tanh(s);
                if (flags & PVS_NORMALIZE) s = s / (sqrtf(rho) + 1e-8f);
            }
            // ---- attention gate ----
            float a = 1.f;
            if (eatt) {
                a = pvs_att_act(att_act, dot_tab<HB>(wat, hh, m) + bac);
                if (valid && hh == 0) io.att_out[e] = a;
            }
            // ---- hand the weighted messages to the channel-per-lane reduction ----
            const float wgt = valid ? a : 0.f;
#pragma unroll
            for (int b = 0; b < HB; ++b)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
                    *reinterpret_cast<float4*>(tile + j * TS + 32 * b + 8 * gq + 4 * hh) =
                        make_float4(wgt * m[b][4 * gq], wgt * m[b][4 * gq + 1], wgt * m[b][4 * gq + 2],
                                    wgt * m[b][4 * gq + 3]);
            if (hh == 0) {
                const float sv = valid ? s : 0.f;
                *reinterpret_cast<float4*>(tx + j * 4) = make_float4(d0 * sv, d1 * sv, d2 * sv, 0.f);
                rowbuf[j] = i;
            }
            pvs_wave_lds_sync();
            if (bmask == 0u) {   // the whole tile continues the current row
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int el = 2 * k + hh;
#pragma unroll
                    for (int b = 0; b < HB; ++b) acc[b] += tile[el * TS + 32 * b + j];
                    if (upd) accx += tx[el * 4 + (j & 3)];
                }
            } else {
                for (int el = 0; el < kTile; ++el) {
                    if ((bmask >> el) & 1u) {
                        flush(cur_row);
                        cur_row = __builtin_amdgcn_readfirstlane(rowbuf[el]);
                    }
                    if ((el & 1) == hh) {
#pragma unroll
                        for (int b = 0; b < HB; ++b) acc[b] += tile[el * TS + 32 * b + j];
                        if (upd) accx += tx[el * 4 + (j & 3)];
                    }
                }
            }
            pvs_wave_lds_sync();
        }
        flush(cur_row);
    }
}


// ---- X-layout helpers ----
template <int HB>
__device__ __forceinline__ void load_x(const float* __restrict__ base, int hh, float (&out)[HB][16]) {
    // base points at channel 0 of one row of a row-major [rows][H] array
#pragma unroll
    for (int b = 0; b < HB; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 v = *reinterpret_cast<const float4*>(base + 32 * b + 8 * g + 4 * hh);
            out[b][4 * g] = v.x; out[b][4 * g + 1] = v.y; out[b][4 * g + 2] = v.z; out[b][4 * g + 3] = v.w;
        }
}

template <int HB>
__device__ __forceinline__ void store_x(float* __restrict__ base, int hh, const float (&v)[HB][16]) {
#pragma unroll
    for (int b = 0; b < HB; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(base + 32 * b + 8 * g + 4 * hh) =
                make_float4(v[b][4 * g], v[b][4 * g + 1], v[b][4 * g + 2], v[b][4 * g + 3]);
}

// Backward of the per-edge work on the matrix cores. Per 32-edge tile: recompute the forward
// (2 products), back-propagate (2 transposed products) and accumulate the two HxH weight
// gradients as products over the EDGE index (operands re-read edge-major from per-wave LDS tiles),
// 96 MFMAs per 32x32 block in total. Vector gradients ride on the same LDS tiles with the channel
// on the lane (one accumulator register each).
template <int HB, bool ERES, bool EATT>
__global__ void __launch_bounds__(kThreads, 2)
k_edge_bwd_mfma(PvsGraph g, PvsEdgeW w, uint32_t flags, int att_act, PvsEdgeBwdIO io, int n_chunks) {
    constexpr int H = 32 * HB;
    constexpr int TS = H + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* W2s = smem;               // z2 = W2 a1
    float* W2ts = W2s + H * H;       // g_a1 = W2^T g_z2
    float* Wc1s = W2ts + H * H;      // zc = Wc1 m
    float* Wc1ts = Wc1s + H * H;     // g_m += Wc1^T g_zc
    float* b2t = Wc1ts + H * H;
    float* bc1t = b2t + H;
    float* wc2t = bc1t + H;
    float* wat = wc2t + H;
    float* wrhot = wat + H;
    float* attrt = wrhot + H;                            // [PVS_MAX_EDGE_ATTR][H]
    float* wave_base = attrt + PVS_MAX_EDGE_ATTR * H;
    // per wave: T0 (a1), T1 (m, then g_z1), T2 (g_zc, then g_z2), tx[32][4], sc[32][8], rowbuf[32]
    constexpr int kWaveFloats = 3 * kTile * TS + kTile * 4 + kTile * 8 + kTile;

    const bool upd = (flags & PVS_UPDATE_COORDS) && io.gxagg != nullptr;
    constexpr bool eatt = EATT;
    constexpr bool eres = ERES;

    stage_weights<HB>(W2s, w.w2, false);
    stage_weights<HB>(W2ts, w.w2, true);
    if (upd) {
        stage_weights<HB>(Wc1s, w.wc1, false);
        stage_weights<HB>(Wc1ts, w.wc1, true);
    }
    for (int c = threadIdx.x; c < H; c += kThreads) {
        b2t[c] = w.b2[c];
        bc1t[c] = upd ? w.bc1[c] : 0.f;
        wc2t[c] = upd ? w.wc2[c] : 0.f;
        wat[c] = eatt ? w.wa[c] : 0.f;
        wrhot[c] = w.w1[c * w.ld1 + w.off_rho];
        for (int t = 0; t < PVS_MAX_EDGE_ATTR; ++t)
            attrt[t * H + c] = t < w.n_attr ? w.w1[c * w.ld1 + w.off_rho + 1 + t] : 0.f;
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int j = lane & 31, hh = lane >> 5;
    float* T0 = wave_base + wv * kWaveFloats;
    float* T1 = T0 + kTile * TS;
    float* T2 = T1 + kTile * TS;
    float* tx = T2 + kTile * TS;
    float* sc = tx + kTile * 4;      // per edge: [0]=rho, [1]=g_logit, [2..2+A)=one-hot(type)
    int* rowbuf = reinterpret_cast<int*>(sc + kTile * 8);
    const float bac = eatt ? w.ba[0] : 0.f;
    float gate_raw = 0.f, gate = 1.f;
    if (eres && (flags & (PVS_REZERO | PVS_GATED_RESIDUAL))) {
        gate_raw = w.edge_gate[0];
        gate = (flags & PVS_GATED_RESIDUAL) ? fmaxf(gate_raw, 0.f) : gate_raw;
    }

    // ---- accumulators that live for the whole kernel ----
    f32x16 gW2[HB][HB], gWc1[HB][HB];       // D layout: [c = 32bo + ch(r,hh)][k = 32bi + j]
#pragma unroll
    for (int bo = 0; bo < HB; ++bo)
#pragma unroll
        for (int bi = 0; bi < HB; ++bi)
#pragma unroll
            for (int r = 0; r < 16; ++r) { gW2[bo][bi][r] = 0.f; gWc1[bo][bi][r] = 0.f; }
    float g_wc2x[HB][16];                    // X layout (channel in the register, edges on lanes)
#pragma unroll
    for (int b = 0; b < HB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) g_wc2x[b][r] = 0.f;
    // channel-on-lane accumulators (lane = channel 32b + j, parity hh)
    float g_b2[HB], g_bc1[HB], g_wa[HB], g_wrho[HB], g_wattr[HB][6];
#pragma unroll
    for (int b = 0; b < HB; ++b) {
        g_b2[b] = g_bc1[b] = g_wa[b] = g_wrho[b] = 0.f;
#pragma unroll
        for (int t = 0; t < 6; ++t) g_wattr[b][t] = 0.f;
    }
    float g_ba = 0.f, g_gate = 0.f;

    const int total_waves = gridDim.x * kWaves;
    for (int chunk = blockIdx.x * kWaves + wv; chunk < n_chunks; chunk += total_waves) {
        const int e_begin = chunk_begin(g, chunk, n_chunks);
        const int e_end = chunk_begin(g, chunk + 1, n_chunks);
        int cur_row = -1;
        float gP[HB], gxr = 0.f;
#pragma unroll
        for (int b = 0; b < HB; ++b) gP[b] = 0.f;

        auto flush = [&](int row_id) {
            if (row_id < 0) return;
#pragma unroll
            for (int b = 0; b < HB; ++b) {
                const float tot = gP[b] + __shfl_xor(gP[b], 32, 64);
                if (hh == 0) io.gPQ[(size_t)row_id * 2 * H + 32 * b + j] = tot;
                gP[b] = 0.f;
            }
            const float totx = gxr + __shfl_xor(gxr, 32, 64);
            if (hh == 0 && j < 3) io.gx_row[3 * row_id + j] = totx;
            gxr = 0.f;
        };

        for (int e0 = e_begin; e0 < e_end; e0 += kTile) {
            const int e = e0 + j;
            const bool valid = e < e_end;
            const float vm = valid ? 1.f : 0.f;
            const int ee = valid ? e : e_end - 1;
            const int i = g.row[ee], jn = g.col[ee];
            const int ty = w.n_attr ? (int)g.etype[ee] : 0;
            const int prev_row = (ee == e_begin) ? -1 : g.row[ee - 1];
            const unsigned bmask = (unsigned)__ballot(valid && hh == 0 && i != prev_row);
            const float d0 = io.x[3 * i] - io.x[3 * jn], d1 = io.x[3 * i + 1] - io.x[3 * jn + 1];
            const float d2 = io.x[3 * i + 2] - io.x[3 * jn + 2];
            const float rho = d0 * d0 + d1 * d1 + d2 * d2;

            // ---- recompute: z1, a1 ----
            {
                const float* Pp = io.PQ + (size_t)i * 2 * H + 4 * hh;
                const float* Qp = io.PQ + (size_t)jn * 2 * H + H + 4 * hh;
                const float* At = attrt + ty * H + 4 * hh;
                const float* Rt = wrhot + 4 * hh;
                float a1[HB][16];
#pragma unroll
                for (int b = 0; b < HB; ++b)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const int off = 32 * b + 8 * gq;
                        const float4 p = *reinterpret_cast<const float4*>(Pp + off);
                        const float4 q = *reinterpret_cast<const float4*>(Qp + off);
                        const float4 a = *reinterpret_cast<const float4*>(At + off);
                        const float4 r = *reinterpret_cast<const float4*>(Rt + off);
                        const float zz[4] = {p.x + q.x + fmaf(r.x, rho, a.x), p.y + q.y + fmaf(r.y, rho, a.y),
                                             p.z + q.z + fmaf(r.z, rho, a.z), p.w + q.w + fmaf(r.w, rho, a.w)};
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) {
                            a1[b][4 * gq + q4] = pvs_silu(zz[q4]);
                        }
                    }
                // a1 edge-major in T0 for the W2 weight gradient (zero rows for padded slots)
#pragma unroll
                for (int b = 0; b < HB; ++b)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq)
                        *reinterpret_cast<float4*>(T0 + j * TS + 32 * b + 8 * gq + 4 * hh) =
                            make_float4(a1[b][4 * gq], a1[b][4 * gq + 1], a1[b][4 * gq + 2], a1[b][4 * gq + 3]);
                // ---- z2 = W2 a1 + b2 ----
                f32x16 acc2[HB];
                float bias[HB][16];
                load_tab<HB>(b2t, hh, bias);
#pragma unroll
                for (int b = 0; b < HB; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc2[b][r] = bias[b][r];
                mfma_chain<HB>(W2s, lane, a1, acc2);
                float dz2[HB][16], m[HB][16];     // SiLU'(z2) and the message
                float m_new[ERES ? HB : 1][16], mp[ERES ? HB : 1][16];
#pragma unroll
                for (int b = 0; b < HB; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float z2 = acc2[b][r];
                        const float sg = pvs_sigmoid(z2);
                        dz2[b][r] = pvs_silu_grad(z2, sg);
                        m[b][r] = z2 * sg;
                        if constexpr (ERES) m_new[b][r] = m[b][r];
                    }
                if constexpr (ERES) {
                    load_x<HB>(io.m_prev + (size_t)ee * H, hh, mp);
#pragma unroll
                    for (int b = 0; b < HB; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            if (flags & PVS_REZERO) m[b][r] = mp[b][r] + gate * m_new[b][r];
                            else if (flags & PVS_GATED_RESIDUAL) m[b][r] = gate * m_new[b][r] + (1.f - gate) * mp[b][r];
                            else m[b][r] = m_new[b][r] + mp[b][r];
                        }
                }
                // m edge-major in T1 (operand of the Wc1 weight gradient and of g_wa)
#pragma unroll
                for (int b = 0; b < HB; ++b)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq)
                        *reinterpret_cast<float4*>(T1 + j * TS + 32 * b + 8 * gq + 4 * hh) =
                            make_float4(m[b][4 * gq], m[b][4 * gq + 1], m[b][4 * gq + 2], m[b][4 * gq + 3]);

                // ---- gradient wrt m: external + attention + coordinate branch ----
                f32x16 gm[HB];
                {
                    float init[HB][16];
                    if (io.g_m_out) load_x<HB>(io.g_m_out + (size_t)ee * H, hh, init);
#pragma unroll
                    for (int b = 0; b < HB; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r) gm[b][r] = io.g_m_out ? init[b][r] * vm : 0.f;
                }
                float gMi[HB][16];
                load_x<HB>(io.gM + (size_t)i * H, hh, gMi);
                float g_l = 0.f, aval = 1.f;
                if constexpr (EATT) {
                    const float logit = dot_tab<HB>(wat, hh, m) + bac;
                    aval = io.att[ee];
                    float dot = 0.f;
#pragma unroll
                    for (int b = 0; b < HB; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r) dot = fmaf(m[b][r], gMi[b][r], dot);
                    dot += __shfl_xor(dot, 32, 64);
                    g_l = pvs_att_act_grad(att_act, logit, aval) * dot * vm;
                    if (hh == 0) g_ba += g_l;
                    float wax[HB][16];
                    load_tab<HB>(wat, hh, wax);
#pragma unroll
                    for (int b = 0; b < HB; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r) gm[b][r] += (aval * vm) * gMi[b][r] + g_l * wax[b][r];
                } else {
#pragma unroll
                    for (int b = 0; b < HB; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r) gm[b][r] += vm * gMi[b][r];
                }
                float s_coord = 0.f, nrm = 1.f;
                float gT0 = 0.f, gT1 = 0.f, gT2 = 0.f;
                if (upd) {
                    gT0 = io.gxagg[3 * i]; gT1 = io.gxagg[3 * i + 1]; gT2 = io.gxagg[3 * i + 2];
                    f32x16 accc[HB];
                    float bias2[HB][16];
                    load_tab<HB>(bc1t, hh, bias2);
#pragma unroll
                    for (int b = 0; b < HB; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r) accc[b][r] = bias2[b][r];
                    mfma_chain<HB>(Wc1s, lane, m, accc);
                    float q[HB][16], dq[HB][16];
#pragma unroll
                    for (int b = 0; b < HB; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float zc = accc[b][r];
                            const float sg = pvs_sigmoid(zc);
                            q[b][r] = zc * sg;
                            dq[b][r] = pvs_silu_grad(zc, sg);
                        }
                    float s = dot_tab<HB>(wc2t, hh, q);
                    float dact = 1.f;
                    if (flags & PVS_TANH) { s = pvs_tanh(s); dact = 1.f - s * s; }
                    if (flags & PVS_NORMALIZE) nrm = 1.f / (sqrtf(rho) + 1e-8f);
                    s_coord = s;
                    const float g_s = (d0 * gT0 + d1 * gT1 + d2 * gT2) * nrm * dact * vm;
                    float wc2x[HB][16], g_zc[HB][16];
                    load_tab<HB>(wc2t, hh, wc2x);
#pragma unroll
                    for (int b = 0; b < HB; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            g_zc[b][r] = g_s * wc2x[b][r] * dq[b][r];
                            g_wc2x[b][r] = fmaf(g_s, q[b][r], g_wc2x[b][r]);
                        }
#pragma unroll
                    for (int b = 0; b < HB; ++b)
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq)
                            *reinterpret_cast<float4*>(T2 + j * TS + 32 * b + 8 * gq + 4 * hh) =
                                make_float4(g_zc[b][4 * gq], g_zc[b][4 * gq + 1], g_zc[b][4 * gq + 2],
                                            g_zc[b][4 * gq + 3]);
                    mfma_chain<HB>(Wc1ts, lane, g_zc, gm);     // g_m += Wc1^T g_zc
                }
                if (hh == 0) {
                    float4 s0 = make_float4(rho, g_l, 0.f, 0.f), s1v = make_float4(0.f, 0.f, 0.f, 0.f);
                    float oh[6];
#pragma unroll
                    for (int t = 0; t < 6; ++t) oh[t] = (valid && w.n_attr && ty == t) ? 1.f : 0.f;
                    s0.z = oh[0]; s0.w = oh[1];
                    s1v = make_float4(oh[2], oh[3], oh[4], oh[5]);
                    *reinterpret_cast<float4*>(sc + j * 8) = s0;
                    *reinterpret_cast<float4*>(sc + j * 8 + 4) = s1v;
                    rowbuf[j] = i;
                }
                pvs_wave_lds_sync();
                // ---- Wc1 weight gradient + g_bc1 + g_wa from the edge-major tiles ----
                if (upd || eatt) {
#pragma unroll
                    for (int sI = 0; sI < 16; ++sI) {
                        const int el = 2 * sI + hh;
                        float av[HB], bv[HB];
#pragma unroll
                        for (int b = 0; b < HB; ++b) {
                            av[b] = upd ? T2[el * TS + 32 * b + j] : 0.f;
                            bv[b] = T1[el * TS + 32 * b + j];
                        }
                        const float gl_e = sc[el * 8 + 1];
#pragma unroll
                        for (int b = 0; b < HB; ++b) {
                            g_bc1[b] += av[b];
                            g_wa[b] = fmaf(gl_e, bv[b], g_wa[b]);
                        }
                        if (upd) {
#pragma unroll
                            for (int bo = 0; bo < HB; ++bo)
#pragma unroll
                                for (int bi = 0; bi < HB; ++bi)
                                    gWc1[bo][bi] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                        av[bo], bv[bi], gWc1[bo][bi], 0, 0, 0);
                        }
                    }
                }
                // ---- edge residual ----
                float g_z2[HB][16];
#pragma unroll
                for (int b = 0; b < HB; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float gmv = gm[b][r];
                        float gnew = gmv;
                        if constexpr (ERES) {
                            if (flags & PVS_REZERO) {
                                gnew = gate * gmv;
                                g_gate = fmaf(gmv, m_new[b][r], g_gate);
                                mp[b][r] = gmv;
                            } else if (flags & PVS_GATED_RESIDUAL) {
                                gnew = gate * gmv;
                                if (gate_raw > 0.f) g_gate = fmaf(gmv, m_new[b][r] - mp[b][r], g_gate);
                                mp[b][r] = (1.f - gate) * gmv;
                            } else {
                                mp[b][r] = gmv;
                            }
                        }
                        g_z2[b][r] = gnew * dz2[b][r];
                    }
                if constexpr (ERES) {
                    if (valid) store_x<HB>(io.g_m_prev + (size_t)e * H, hh, mp);
                }
                pvs_wave_lds_sync();      // all reads of T2 (g_zc) done before it is reused
#pragma unroll
                for (int b = 0; b < HB; ++b)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq)
                        *reinterpret_cast<float4*>(T2 + j * TS + 32 * b + 8 * gq + 4 * hh) =
                            make_float4(g_z2[b][4 * gq], g_z2[b][4 * gq + 1], g_z2[b][4 * gq + 2],
                                        g_z2[b][4 * gq + 3]);
                // ---- g_a1 = W2^T g_z2 ; g_z1 = g_a1 * SiLU'(z1) ----
                f32x16 ga1[HB];
#pragma unroll
                for (int b = 0; b < HB; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) ga1[b][r] = 0.f;
                mfma_chain<HB>(W2ts, lane, g_z2, ga1);
                // SiLU'(z1): z1 is re-gathered here (L2-hot) instead of living in 16 registers
                // across the whole tile
                float g_z1[HB][16];
#pragma unroll
                for (int b = 0; b < HB; ++b)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const int off = 32 * b + 8 * gq;
                        const float4 p = *reinterpret_cast<const float4*>(Pp + off);
                        const float4 q = *reinterpret_cast<const float4*>(Qp + off);
                        const float4 a = *reinterpret_cast<const float4*>(At + off);
                        const float4 r = *reinterpret_cast<const float4*>(Rt + off);
                        const float zz[4] = {p.x + q.x + fmaf(r.x, rho, a.x), p.y + q.y + fmaf(r.y, rho, a.y),
                                             p.z + q.z + fmaf(r.z, rho, a.z), p.w + q.w + fmaf(r.w, rho, a.w)};
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4)
                            g_z1[b][4 * gq + q4] =
                                ga1[b][4 * gq + q4] * pvs_silu_grad(zz[q4], pvs_sigmoid(zz[q4]));
                    }
                if (valid) store_x<HB>(io.gz1 + (size_t)e * H, hh, g_z1);
                const float g_rho = dot_tab<HB>(wrhot, hh, g_z1);
                const float k1 = s_coord * nrm * vm;
                const float gd0 = fmaf(k1, gT0, 2.f * d0 * g_rho);
                const float gd1 = fmaf(k1, gT1, 2.f * d1 * g_rho);
                const float gd2 = fmaf(k1, gT2, 2.f * d2 * g_rho);
                if (hh == 0) {
                    *reinterpret_cast<float4*>(tx + j * 4) = make_float4(gd0, gd1, gd2, 0.f);
                    if (valid) {
                        io.gd[(size_t)e * 3] = gd0; io.gd[(size_t)e * 3 + 1] = gd1; io.gd[(size_t)e * 3 + 2] = gd2;
                    }
                }
                pvs_wave_lds_sync();      // T2 = g_z2 visible; T1 (m) no longer needed
                // ---- W2 weight gradient + g_b2 ----
#pragma unroll
                for (int sI = 0; sI < 16; ++sI) {
                    const int el = 2 * sI + hh;
                    float av[HB], bv[HB];
#pragma unroll
                    for (int b = 0; b < HB; ++b) {
                        av[b] = T2[el * TS + 32 * b + j];
                        bv[b] = T0[el * TS + 32 * b + j];
                        g_b2[b] += av[b];
                    }
#pragma unroll
                    for (int bo = 0; bo < HB; ++bo)
#pragma unroll
                        for (int bi = 0; bi < HB; ++bi)
                            gW2[bo][bi] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[bo], bv[bi], gW2[bo][bi], 0, 0, 0);
                }
                // ---- g_z1 edge-major in T1: row sums (g_P), g_wrho, g_wattr, row-side g_x ----
#pragma unroll
                for (int b = 0; b < HB; ++b)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq)
                        *reinterpret_cast<float4*>(T1 + j * TS + 32 * b + 8 * gq + 4 * hh) =
                            make_float4(g_z1[b][4 * gq], g_z1[b][4 * gq + 1], g_z1[b][4 * gq + 2],
                                        g_z1[b][4 * gq + 3]);
                pvs_wave_lds_sync();
                for (int el = 0; el < kTile; ++el) {
                    if (bmask && ((bmask >> el) & 1u)) {
                        flush(cur_row);
                        cur_row = __builtin_amdgcn_readfirstlane(rowbuf[el]);
                    }
                    if ((el & 1) == hh) {
                        const float4 s0 = *reinterpret_cast<const float4*>(sc + el * 8);
                        const float4 s1v = *reinterpret_cast<const float4*>(sc + el * 8 + 4);
                        const float ohv[6] = {s0.z, s0.w, s1v.x, s1v.y, s1v.z, s1v.w};
#pragma unroll
                        for (int b = 0; b < HB; ++b) {
                            const float v = T1[el * TS + 32 * b + j];
                            gP[b] += v;
                            g_wrho[b] = fmaf(v, s0.x, g_wrho[b]);
#pragma unroll
                            for (int t = 0; t < 6; ++t) g_wattr[b][t] = fmaf(v, ohv[t], g_wattr[b][t]);
                        }
                        gxr += tx[el * 4 + (j & 3)];
                    }
                }
                pvs_wave_lds_sync();
            }
        }
        flush(cur_row);
    }

    // ---- block reduction into one slab, fixed order ----
    const PvsSlabLayout L = pvs_slab_layout(H);
    __syncthreads();
    float* slab = smem;
    for (int i = threadIdx.x; i < L.total; i += kThreads) slab[i] = 0.f;
    __syncthreads();
    // X-layout g_wc2: sum over the 32 edge lanes of each half
#pragma unroll
    for (int b = 0; b < HB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = g_wc2x[b][r];
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) v += __shfl_xor(v, o, 64);
            g_wc2x[b][r] = v;
        }
    g_ba += __shfl_xor(g_ba, 32, 64);          // only hh == 0 lanes accumulated
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) g_ba += __shfl_xor(g_ba, o, 64);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) g_gate += __shfl_xor(g_gate, o, 64);
#pragma unroll
    for (int b = 0; b < HB; ++b) {
        g_b2[b] += __shfl_xor(g_b2[b], 32, 64);
        g_bc1[b] += __shfl_xor(g_bc1[b], 32, 64);
        g_wa[b] += __shfl_xor(g_wa[b], 32, 64);
        g_wrho[b] += __shfl_xor(g_wrho[b], 32, 64);
#pragma unroll
        for (int t = 0; t < 6; ++t) g_wattr[b][t] += __shfl_xor(g_wattr[b][t], 32, 64);
    }
    for (int turn = 0; turn < kWaves; ++turn) {
        if (wv == turn) {
#pragma unroll
            for (int bo = 0; bo < HB; ++bo)
#pragma unroll
                for (int bi = 0; bi < HB; ++bi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int c = 32 * bo + xch(r, hh), k = 32 * bi + j;
                        slab[L.w2 + c * H + k] += gW2[bo][bi][r];
                        slab[L.wc1 + c * H + k] += gWc1[bo][bi][r];
                    }
            if (hh == 0) {
#pragma unroll
                for (int b = 0; b < HB; ++b) {
                    const int c = 32 * b + j;
                    slab[L.b2 + c] += g_b2[b];
                    slab[L.bc1 + c] += g_bc1[b];
                    slab[L.wa + c] += g_wa[b];
                    slab[L.wrho + c] += g_wrho[b];
#pragma unroll
                    for (int t = 0; t < 6; ++t) slab[L.wattr + t * H + c] += g_wattr[b][t];
                }
            }
            if (j == 0) {
#pragma unroll
                for (int b = 0; b < HB; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) slab[L.wc2 + 32 * b + xch(r, hh)] += g_wc2x[b][r];
            }
            if (lane == 0) { slab[L.ba] += g_ba; slab[L.gate] += g_gate; }
        }
        __syncthreads();
    }
    float* dst = io.slabs + (size_t)blockIdx.x * L.total;
    for (int i = threadIdx.x; i < L.total; i += kThreads) dst[i] = slab[i];
}

template <typename K>
int set_lds(K kernel, size_t lds) {
    if (lds > 48 * 1024)
        PVS_CHECK_HIP(hipFuncSetAttribute((const void*)kernel,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return 0;
}

void pick_grid(int E, int* blocks, int* n_chunks) {
    // one chunk = a few thousand edges; every wave gets the same number of chunks
    long long b = ((long long)E + 4095) / 4096;
    if (b < 1) b = 1;
    if (b > 1024) b = 1024;
    const long long waves = b * kWaves;
    long long per_wave = ((long long)E + waves * 4096 - 1) / (waves * 4096);
    if (per_wave < 1) per_wave = 1;
    *blocks = (int)b;
    *n_chunks = (int)(waves * per_wave);
}

}  // namespace

int pvs_edge_mfma_supported(int H, uint32_t flags) {
    if (H != 32 && H != 64) return 0;
    if ((flags & PVS_EDGE_ATTENTION) && (flags & PVS_SOFTMAX_ATT)) return 0;   // generic path
    return 1;
}

int pvs_launch_edge_fwd_mfma(hipStream_t s, int H, const PvsGraph& g, const PvsEdgeW& w, uint32_t flags,
                             int att_act, const PvsEdgeFwdIO& io) {
    PVS_REQUIRE(w.n_attr <= PVS_MAX_EDGE_ATTR, "edge_attr classes %d > %d", w.n_attr,
                PVS_MAX_EDGE_ATTR);
    // rows without edges are never flushed: M = 0, x_out = x
    PVS_CHECK_HIP(hipMemsetAsync(io.Magg, 0, sizeof(float) * (size_t)g.n_nodes * H, s));
    if (flags & PVS_UPDATE_COORDS)
        PVS_CHECK_HIP(hipMemcpyAsync(io.x_out, io.x, sizeof(float) * 3 * (size_t)g.n_nodes,
                                     hipMemcpyDeviceToDevice, s));
    if (g.n_edges == 0) return 0;
    int blocks, n_chunks;
    pick_grid(g.n_edges, &blocks, &n_chunks);
    PvsProfScope prof(s, PVS_PROF_EDGE_FWD);
    const int HB = H / 32;
    const size_t words = (size_t)2 * H * H + (5 + PVS_MAX_EDGE_ATTR) * H +
                         (size_t)kWaves * (kTile * (H + 4) + kTile * 4 + kTile);
    const size_t lds = words * sizeof(float);
    if (HB == 1) {
        if (set_lds(k_edge_fwd_mfma<1>, lds)) return -2;
        k_edge_fwd_mfma<1><<<blocks, kThreads, lds, s>>>(g, w, flags, att_act, io, n_chunks);
    } else {
        if (set_lds(k_edge_fwd_mfma<2>, lds)) return -2;
        k_edge_fwd_mfma<2><<<blocks, kThreads, lds, s>>>(g, w, flags, att_act, io, n_chunks);
    }
    PVS_CHECK_LAUNCH();
    return 0;
}

int pvs_launch_edge_bwd_mfma(hipStream_t s, int H, const PvsGraph& g, const PvsEdgeW& w, uint32_t flags,
                             int att_act, const PvsEdgeBwdIO& io, int* n_slabs) {
    PVS_REQUIRE(w.n_attr <= 6, "MFMA edge backward supports up to 6 edge classes (got %d)", w.n_attr);
    PVS_REQUIRE(H == 32, "MFMA edge backward is built for H = 32 (got %d)", H);
    // rows without edges are never flushed
    PVS_CHECK_HIP(hipMemsetAsync(io.gPQ, 0, sizeof(float) * 2 * (size_t)g.n_nodes * H, s));
    PVS_CHECK_HIP(hipMemsetAsync(io.gx_row, 0, sizeof(float) * 3 * (size_t)g.n_nodes, s));
    int blocks, n_chunks;
    pick_grid(g.n_edges, &blocks, &n_chunks);
    if (blocks > 512) {   // the slab buffer holds 512 partials
        blocks = 512;
        const long long waves = (long long)blocks * kWaves;
        long long per_wave = ((long long)g.n_edges + waves * 4096 - 1) / (waves * 4096);
        n_chunks = (int)(waves * (per_wave < 1 ? 1 : per_wave));
    }
    *n_slabs = blocks;
    PvsProfScope prof(s, PVS_PROF_EDGE_BWD);
    const PvsSlabLayout L = pvs_slab_layout(H);
    size_t words = (size_t)4 * H * H + (5 + PVS_MAX_EDGE_ATTR) * H +
                   (size_t)kWaves * (3 * kTile * (H + 4) + kTile * 4 + kTile * 8 + kTile);
    if (words < (size_t)L.total) words = L.total;
    const size_t lds = words * sizeof(float);
    const bool eres = (flags & PVS_EDGE_RESIDUAL) && io.m_prev != nullptr;
    const bool eatt = flags & PVS_EDGE_ATTENTION;
#define PVS_BWD_LAUNCH(ER, EA)                                                                    \
    do {                                                                                          \
        if (set_lds(k_edge_bwd_mfma<1, ER, EA>, lds)) return -2;                                  \
        k_edge_bwd_mfma<1, ER, EA><<<blocks, kThreads, lds, s>>>(g, w, flags, att_act, io, n_chunks); \
    } while (0)
    if (eres && eatt) PVS_BWD_LAUNCH(true, true);
    else if (eres) PVS_BWD_LAUNCH(true, false);
    else if (eatt) PVS_BWD_LAUNCH(false, true);
    else PVS_BWD_LAUNCH(false, false);
#undef PVS_BWD_LAUNCH
    PVS_CHECK_LAUNCH();
    return 0;
}

int pvs_edge_bwd_mfma_supported(int H, uint32_t flags, int n_attr) {
    if (H != 32 || n_attr > 6) return 0;
    if ((flags & PVS_EDGE_ATTENTION) && (flags & PVS_SOFTMAX_ATT)) return 0;
    return 1;
}
