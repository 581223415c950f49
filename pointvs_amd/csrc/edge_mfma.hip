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
