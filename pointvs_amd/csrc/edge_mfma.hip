// MFMA edge backward kernels for H = 32*HB (HB = 1, 2); the forward kernel and the description of the
// X layout live in edge_mfma_fwd.hip, the shared device helpers in edge_mfma_common.h.
#include "edge_mfma_common.h"

namespace {

// Backward of the per-edge work on the matrix cores. Per 32-edge tile: recompute the forward
// (2 products), back-propagate (2 transposed products) and accumulate the two HxH weight
// gradients as products over the EDGE index (operands re-read edge-major from per-wave LDS tiles),
// 96 MFMAs per 32x32 block in total. Vector gradients ride on the same LDS tiles with the channel
// on the lane (one accumulator register each).
template <int HB, bool ERES, bool EATT>
__global__ void __launch_bounds__(kThreads, HB == 1 ? 2 : 1)
k_edge_bwd_mfma(PvsGraph g, PvsEdgeW w, uint32_t flags, int att_act, PvsEdgeBwdIO io, int n_chunks,
                int e_lo, int e_hi) {
    constexpr int H = 32 * HB;
    constexpr int TS = H + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // W2 / Wc1 natural with padded rows (one copy serves W and W^T products); exact fp32 MFMAs throughout: the
    // arithmetic cross-check family of the tests. The split-product forms this kernel carried in rounds 1-2 are gone.
    constexpr int NT = kThreads;
    constexpr int NW = NT / 64;
    constexpr int kWeightWords = 2 * H * (H + 1);
    float* W2n = smem;
    float* Wc1n = W2n + H * (H + 1);
    float* b2t = smem + kWeightWords;
    float* bc1t = b2t + H;
    float* wc2t = bc1t + H;
    float* wat = wc2t + H;
    float* wrhot = wat + H;
    float* attrt = wrhot + H;                            // [PVS_MAX_EDGE_ATTR][H]
    float* wave_base = attrt + PVS_MAX_EDGE_ATTR * H;
    // per wave: T0 (a1), T1 (m, then g_z1), T2 (g_zc, then g_z2), tx[32][4] (gd), gl[32], rowbuf[32]
    constexpr int kWaveFloats = 3 * kTile * TS + kTile * 4 + 2 * kTile;

    const bool upd = (flags & PVS_UPDATE_COORDS) && io.gxagg != nullptr;
    constexpr bool eatt = EATT;
    constexpr bool eres = ERES;

    stage_weights_nat<HB>(W2n, w.w2);
    if (upd) stage_weights_nat<HB>(Wc1n, w.wc1);
    for (int c = threadIdx.x; c < H; c += NT) {
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
    float* glb = tx + kTile * 4;
    int* rowbuf = reinterpret_cast<int*>(glb + kTile);
    const float bac = eatt ? w.ba[0] : 0.f;
    float gate_raw = 0.f, gate = 1.f;
    if (eres && (flags & (PVS_REZERO | PVS_GATED_RESIDUAL))) {
        gate_raw = w.edge_gate[0];
        gate = (flags & PVS_GATED_RESIDUAL) ? fmaxf(gate_raw, 0.f) : gate_raw;
    }
    // edge residual without per-element branches: m = res_a * m_new + res_b * m_prev
    // (plain: 1, 1; rezero: g, 1; gated: relu(g), 1 - relu(g))
    const float res_a = (flags & (PVS_REZERO | PVS_GATED_RESIDUAL)) ? gate : 1.f;
    const float res_b = (flags & PVS_GATED_RESIDUAL) ? 1.f - gate : 1.f;

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
    float g_b2[HB], g_bc1[HB], g_wa[HB];
#pragma unroll
    for (int b = 0; b < HB; ++b) g_b2[b] = g_bc1[b] = g_wa[b] = 0.f;
    float g_ba = 0.f, g_gate = 0.f;

    const int total_waves = gridDim.x * NW;
    for (int chunk = pvs_xcd_block(blockIdx.x, gridDim.x) * NW + wv; chunk < n_chunks; chunk += total_waves) {
        const int e_begin = chunk_begin(g, chunk, n_chunks, e_lo, e_hi);
        const int e_end = chunk_begin(g, chunk + 1, n_chunks, e_lo, e_hi);
        int cur_row = -1;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), accx = acc;   // open row: lane = (row slot, quad)
        constexpr int QPR = H / 4;
        const int quad = lane % QPR, rsub = lane / QPR;
        auto flush = [&](int row_id) {
            if (row_id >= 0) {
                const float4 tot = sum_row_slots<HB>(acc);
                if (rsub == 0) *reinterpret_cast<float4*>(io.gPQ + (size_t)row_id * 2 * H + 4 * quad) = tot;
                const float4 tx4 = sum_row_slots<HB>(accx);
                if (lane == 0) {
                    io.gx_row[3 * row_id] = tx4.x;
                    io.gx_row[3 * row_id + 1] = tx4.y;
                    io.gx_row[3 * row_id + 2] = tx4.z;
                }
            }
            acc = make_float4(0.f, 0.f, 0.f, 0.f);
            accx = acc;
        };
        // tile t+1's indices are loaded at the top of tile t; its node rows are gathered at its own
        // start (register prefetch of the rows measured slower: occupancy)
        TileIdx I = load_tile_idx(g, w.n_attr | ((flags & kAblNoGather) ? 0x100 : 0), e_begin, e_begin, e_end, j);
        TileGather<HB> G;
        for (int e0 = e_begin; e0 < e_end; e0 += kTile) {
            const int e_next = (e0 + kTile < e_end) ? e0 + kTile : e0;
            const TileIdx In = load_tile_idx(g, w.n_attr | ((flags & kAblNoGather) ? 0x100 : 0), e_next, e_begin, e_end, j);
            const int e = I.e, ee = I.ee, i = I.i, ty = I.ty;
            const bool valid = I.valid;
            const float vm = valid ? 1.f : 0.f;
            gather_tile<HB>(io.PQ, io.x, I, hh, G);
            const unsigned bmask = (unsigned)__ballot(valid && hh == 0 && i != I.prev_row);
            const float d0 = G.d0, d1 = G.d1, d2 = G.d2;
            const float rho = d0 * d0 + d1 * d1 + d2 * d2;

            // ---- recompute: z1, a1 ----
            {
                float a1[HB][16];
                assemble_z1<HB>(G, attrt, wrhot, ty, hh, rho, a1);
#pragma unroll
                for (int b = 0; b < HB; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) a1[b][r] = pvs_silu(a1[b][r]);
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
                mfma_chain_nat<HB, false>(W2n, lane, a1, acc2, flags & kAblNoMfma);
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
                            m[b][r] = fmaf(res_a, m_new[b][r], res_b * mp[b][r]);
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
                    g_l = (flags & PVS_SOFTMAX_ATT) ? aval * (dot - io.softD[i]) * vm      // softD = M_i . g_M_i
                                             : pvs_att_act_grad(att_act, logit, aval) * dot * vm;
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
                    mfma_chain_nat<HB, false>(Wc1n, lane, m, accc, flags & kAblNoMfma);
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
                    mfma_chain_nat<HB, true>(Wc1n, lane, g_zc, gm, flags & kAblNoMfma);      // g_m += Wc1^T g_zc
                }
                if (hh == 0) { glb[j] = g_l; rowbuf[j] = i; }
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
                        const float gl_e = glb[el];
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
                // SiLU'(z1): this tile's rows are re-gathered (L2-hot) under the W2^T product
                // instead of holding z1 in 16 registers across the whole tile
                gather_tile<HB>(io.PQ, io.x, I, hh, G);
                f32x16 ga1[HB];
#pragma unroll
                for (int b = 0; b < HB; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) ga1[b][r] = 0.f;
                mfma_chain_nat<HB, true>(W2n, lane, g_z2, ga1, flags & kAblNoMfma);
                float g_z1[HB][16];
                assemble_z1<HB>(G, attrt, wrhot, ty, hh, rho, g_z1);
#pragma unroll
                for (int b = 0; b < HB; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        g_z1[b][r] = ga1[b][r] * pvs_silu_grad(g_z1[b][r], pvs_sigmoid(g_z1[b][r]));
                const float g_rho = dot_tab<HB>(wrhot, hh, g_z1);
                const float k1 = s_coord * nrm * vm;
                const float gd0 = fmaf(k1, gT0, 2.f * d0 * g_rho);
                const float gd1 = fmaf(k1, gT1, 2.f * d1 * g_rho);
                const float gd2 = fmaf(k1, gT2, 2.f * d2 * g_rho);
                // per edge: grad wrt (x_row - x_col) and rho, 16 B, for the node gather kernel
                if (hh == 0) {
                    *reinterpret_cast<float4*>(tx + j * 4) = make_float4(gd0, gd1, gd2, 0.f);
                    if (valid)
                        pvs_store_nt(io.gd + (size_t)e * 4, make_float4(gd0, gd1, gd2, pvs_pack_rho_type(rho, ty)));
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
                // ---- g_z1 edge-major in T1, then whole 128-B rows to HBM (8 lanes per row) ----
#pragma unroll
                for (int b = 0; b < HB; ++b)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq)
                        *reinterpret_cast<float4*>(T1 + j * TS + 32 * b + 8 * gq + 4 * hh) =
                            make_float4(g_z1[b][4 * gq], g_z1[b][4 * gq + 1], g_z1[b][4 * gq + 2],
                                        g_z1[b][4 * gq + 3]);
                pvs_wave_lds_sync();
                // whole 128-B rows of g_z1 to HBM + the row-side sums g_P / g_x from the same reads
                if (!(flags & kAblNoReduce))
                    reduce_rows_tile<HB>(T1, tx, rowbuf, bmask, lane, acc, accx, cur_row, flush,
                                         [&](int rl, int q, const float4& v) {
                                             if (e0 + rl < e_end)   // streamed once: non-temporal
                                                 pvs_store_nt(io.gz1 + (size_t)(e0 + rl) * H + 4 * q, v);
                                         });
                I = In;
                pvs_wave_lds_sync();
            }
        }
        flush(cur_row);
    }

    // ---- block reduction into one slab, fixed order ----
    const PvsSlabLayout L = pvs_slab_layout(H);
    __syncthreads();
    float* slab = smem;
    for (int i = threadIdx.x; i < L.total; i += NT) slab[i] = 0.f;
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
    }
    for (int turn = 0; turn < NW; ++turn) {
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
    for (int i = threadIdx.x; i < L.total; i += NT) dst[i] = slab[i];
}


// ====================================================================================================
// H = 32*HB, HB >= 2: a TEAM of HB waves shares one 32-edge tile. Wave cb owns channel block cb: all
// elementwise work, its 32 output channels of every product and row block cb of the weight gradients
// (the register footprint of the H = 32 kernel instead of HB^2 accumulator blocks per wave). The
// other blocks of a product's input are read back in X layout from the edge-major LDS tiles the
// weight-gradient products need anyway; per-edge scalars (dots over all channels) are summed through
// a small LDS array. Teams of a workgroup run in lockstep (uniform trip count) because the only
// barrier is the workgroup barrier.
template <int HB>
__device__ __forceinline__ void xread_block(const float* __restrict__ T, int j, int hh, int bi,
                                            float (&v)[16]) {
    constexpr int TS = 32 * HB + 4;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 q = *reinterpret_cast<const float4*>(T + j * TS + 32 * bi + 8 * g + 4 * hh);
        v[4 * g] = q.x; v[4 * g + 1] = q.y; v[4 * g + 2] = q.z; v[4 * g + 3] = q.w;
    }
}

template <int HB>
__device__ __forceinline__ void xwrite_block(float* __restrict__ T, int j, int hh, int cb,
                                             const float (&v)[16]) {
    constexpr int TS = 32 * HB + 4;
#pragma unroll
    for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(T + j * TS + 32 * cb + 8 * g + 4 * hh) =
            make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
}

// acc (output block cb) += sum over input blocks bi of W(cb,bi) v_bi ; v_cb from registers, the
// other blocks from the tile T. Wn natural padded [H][H+1]; TRANSPOSE: W^T.
// GLOBAL (H = 128: two fp32 weight matrices do not fit in 160 KB of LDS beside the team's tiles): Wn points at
// [W | W^T], two dense [H][H] copies in GLOBAL memory (pvs_stage_weight_pair), so that the 32 lanes of a half always
// read 32 consecutive floats of one row - the product that LDS serves from one padded copy needs the other
// orientation here to stay coalesced. An L1 / L2 hit per MFMA; the fp32 MFMA's 64 cycles cover it.
template <int HB, bool TRANSPOSE, bool GLOBAL = false>
__device__ __forceinline__ void chain_team(const float* __restrict__ Wn, int lane, int cb,
                                           const float (&own)[16], const float* __restrict__ T,
                                           f32x16& acc) {
    constexpr int H = 32 * HB, LD = GLOBAL ? H : H + 1;
    const int j = lane & 31, hh = lane >> 5;
    if constexpr (GLOBAL) { if (!TRANSPOSE) Wn += H * H; }      // W v reads the W^T copy by rows
    // (HB = 4: one input block at a time - fully unrolled, the 64 operand loads of a product are all hoisted and
    // the wave's 512 registers overflow by 250)
#pragma unroll(HB > 2 ? 1 : HB)
    for (int bi = 0; bi < HB; ++bi) {
        float v[16];
        if (bi == cb) {
#pragma unroll
            for (int t = 0; t < 16; ++t) v[t] = own[t];
        } else {
            xread_block<HB>(T, j, hh, bi, v);
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int kc = 32 * bi + (t & 3) + 8 * (t >> 2) + 4 * hh;
            const float a = (TRANSPOSE || GLOBAL) ? Wn[kc * LD + 32 * cb + j] : Wn[(32 * cb + j) * LD + kc];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, v[t], acc, 0, 0, 0);
        }
    }
}

__device__ __forceinline__ float dot16_tab(const float* __restrict__ tab, int hh, const float (&v)[16]) {
    float s = 0.f;   // tab points at this wave's channel block
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 w4 = *reinterpret_cast<const float4*>(tab + 8 * g + 4 * hh);
        s = fmaf(w4.x, v[4 * g], s); s = fmaf(w4.y, v[4 * g + 1], s);
        s = fmaf(w4.z, v[4 * g + 2], s); s = fmaf(w4.w, v[4 * g + 3], s);
    }
    return s + __shfl_xor(s, 32, 64);
}

__device__ __forceinline__ void load16_tab(const float* __restrict__ tab, int hh, float (&out)[16]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 v = *reinterpret_cast<const float4*>(tab + 8 * g + 4 * hh);
        out[4 * g] = v.x; out[4 * g + 1] = v.y; out[4 * g + 2] = v.z; out[4 * g + 3] = v.w;
    }
}

template <int HB, bool ERES, bool EATT, int NT = 64 * HB>
__global__ void __launch_bounds__(NT, 1)
k_edge_bwd_team(PvsGraph g, PvsEdgeW w, uint32_t flags, int att_act, PvsEdgeBwdIO io, int n_chunks,
                int e_lo, int e_hi, const float* __restrict__ wc1_pair) {
    constexpr int H = 32 * HB, TS = H + 4, NW = NT / 64, TEAMS = NW / HB;
    // H = 128: W2 in LDS, Wc1 (both orientations) from global memory (wc1_pair, chain_team<.., GLOBAL>)
    constexpr bool WC1G = HB > 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // natural padded weights [H][H+1] serving W and W^T (exact fp32 MFMAs throughout)
    constexpr int kWeightFloats = (WC1G ? 1 : 2) * H * (H + 1);
    float* W2n = smem;
    float* Wc1n = W2n + H * (H + 1);
    float* b2t = smem + kWeightFloats;
    float* bc1t = b2t + H;
    float* wc2t = bc1t + H;
    float* wat = wc2t + H;
    float* wrhot = wat + H;
    float* attrt = wrhot + H;                              // [PVS_MAX_EDGE_ATTR][H]
    int* lens = reinterpret_cast<int*>(attrt + PVS_MAX_EDGE_ATTR * H);   // [TEAMS] (+pad to 16)
    float* team_base = reinterpret_cast<float*>(lens + 16);
    // per team: T0 (a1), T1 (m, then g_z1), T2 (g_zc, then g_z2), tx[32][4], glb[32], rowbuf[32],
    //           pdA[HB][32], pdB[HB][32]
    constexpr int kTeamFloats = 3 * kTile * TS + kTile * 4 + 2 * kTile + 2 * HB * kTile;

    const bool upd = (flags & PVS_UPDATE_COORDS) && io.gxagg != nullptr;

    stage_weights_nat<HB>(W2n, w.w2);
    if (upd && !WC1G) stage_weights_nat<HB>(Wc1n, w.wc1);
    for (int c = threadIdx.x; c < H; c += NT) {
        b2t[c] = w.b2[c];
        bc1t[c] = upd ? w.bc1[c] : 0.f;
        wc2t[c] = upd ? w.wc2[c] : 0.f;
        wat[c] = EATT ? w.wa[c] : 0.f;
        wrhot[c] = w.w1[c * w.ld1 + w.off_rho];
        for (int t = 0; t < PVS_MAX_EDGE_ATTR; ++t)
            attrt[t * H + c] = t < w.n_attr ? w.w1[c * w.ld1 + w.off_rho + 1 + t] : 0.f;
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int j = lane & 31, hh = lane >> 5;
    const int team = wv / HB, cb = wv % HB;
    const int co = 32 * cb;                       // first channel of this wave's block
    float* T0 = team_base + team * kTeamFloats;
    float* T1 = T0 + kTile * TS;
    float* T2 = T1 + kTile * TS;
    float* tx = T2 + kTile * TS;
    float* glb = tx + kTile * 4;
    int* rowbuf = reinterpret_cast<int*>(glb + kTile);
    float* pdA = reinterpret_cast<float*>(rowbuf + kTile);
    float* pdB = pdA + HB * kTile;
    const float bac = EATT ? w.ba[0] : 0.f;
    float gate_raw = 0.f, gate = 1.f;
    if (ERES && (flags & (PVS_REZERO | PVS_GATED_RESIDUAL))) {
        gate_raw = w.edge_gate[0];
        gate = (flags & PVS_GATED_RESIDUAL) ? fmaxf(gate_raw, 0.f) : gate_raw;
    }
    // edge residual without per-element branches: m = res_a * m_new + res_b * m_prev
    // (plain: 1, 1; rezero: g, 1; gated: relu(g), 1 - relu(g))
    const float res_a = (flags & (PVS_REZERO | PVS_GATED_RESIDUAL)) ? gate : 1.f;
    const float res_b = (flags & PVS_GATED_RESIDUAL) ? 1.f - gate : 1.f;
    auto sum_pd = [&](const float* pd) {
        float s = 0.f;
#pragma unroll
        for (int b = 0; b < HB; ++b) s += pd[b * kTile + j];
        return s;
    };

    // kernel-lifetime accumulators: row block cb of the two weight gradients, own-channel vectors
    f32x16 gW2[HB], gWc1[HB];
#pragma unroll
    for (int bi = 0; bi < HB; ++bi)
#pragma unroll
        for (int r = 0; r < 16; ++r) { gW2[bi][r] = 0.f; gWc1[bi][r] = 0.f; }
    float g_wc2x[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) g_wc2x[r] = 0.f;
    float g_b2 = 0.f, g_bc1 = 0.f, g_wa = 0.f, g_ba = 0.f, g_gate = 0.f;

    for (int cbase = pvs_xcd_block(blockIdx.x, gridDim.x) * TEAMS; cbase < n_chunks; cbase += gridDim.x * TEAMS) {
        const int chunk = cbase + team;
        const int e_begin = chunk < n_chunks ? chunk_begin(g, chunk, n_chunks, e_lo, e_hi) : e_hi;
        const int e_end = chunk < n_chunks ? chunk_begin(g, chunk + 1, n_chunks, e_lo, e_hi) : e_hi;
        __syncthreads();
        if (lane == 0 && cb == 0) lens[team] = e_end - e_begin;
        __syncthreads();
        int max_len = 0;
#pragma unroll
        for (int t = 0; t < TEAMS; ++t) max_len = max(max_len, lens[t]);
        const int n_iter = (max_len + kTile - 1) / kTile;

        int cur_row = -1;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), accx = acc;
        const int quad = lane % 8, rsub = lane / 8;
        auto flush = [&](int row_id) {
            if (row_id >= 0) {
                const float4 tot = sum_row_slots<1>(acc);
                if (rsub == 0) *reinterpret_cast<float4*>(io.gPQ + (size_t)row_id * 2 * H + co + 4 * quad) = tot;
                const float4 tx4 = sum_row_slots<1>(accx);
                if (lane == 0 && cb == 0) {
                    io.gx_row[3 * row_id] = tx4.x;
                    io.gx_row[3 * row_id + 1] = tx4.y;
                    io.gx_row[3 * row_id + 2] = tx4.z;
                }
            }
            acc = make_float4(0.f, 0.f, 0.f, 0.f);
            accx = acc;
        };

        // software pipeline over the tiles of the chunk: the indices of tile it+2 and the gathered rows
        // of tile it+1 are in flight while tile it is processed (one wave per SIMD: nothing else hides
        // the two dependent gather latencies)
        struct Idx { int ee, i, jn, ty, prev; };
        auto idx_of = [&](int itx) {
            Idx o;
            const int e = e_begin + itx * kTile + j;
            int ee = e < e_end ? e : e_end - 1;
            ee = min(max(ee, 0), g.n_edges - 1);
            o.ee = ee;
            o.i = g.row[ee];
            o.jn = g.col[ee];
            o.ty = w.n_attr ? (int)g.etype[ee] : 0;
            o.prev = (ee == e_begin || ee == 0) ? -1 : g.row[ee - 1];
            return o;
        };
        float pp[16], qq[16], xi[3], xj[3];
        auto rows_of = [&](const Idx& t) {
            load16_tab(io.PQ + (size_t)t.i * 2 * H + co, hh, pp);
            load16_tab(io.PQ + (size_t)t.jn * 2 * H + H + co, hh, qq);
#pragma unroll
            for (int c = 0; c < 3; ++c) { xi[c] = io.x[3 * t.i + c]; xj[c] = io.x[3 * t.jn + c]; }
        };
        Idx cur = idx_of(0), nxt = cur;
        if (n_iter > 0) {
            rows_of(cur);
            nxt = idx_of(1);
        }

        for (int it = 0; it < n_iter; ++it) {
            const int e0 = e_begin + it * kTile;
            const int e = e0 + j;
            const bool valid = e < e_end;
            const float vm = valid ? 1.f : 0.f;
            const int ee = cur.ee, i = cur.i, jn = cur.jn, ty = cur.ty;
            const unsigned bmask = (unsigned)__ballot(valid && hh == 0 && i != cur.prev);
            const float d0 = xi[0] - xj[0], d1 = xi[1] - xj[1], d2 = xi[2] - xj[2];
            const float rho = d0 * d0 + d1 * d1 + d2 * d2;

            // ---- recompute: a1 (own block) -> T0 ----
            float a1[16], sd1[16];     // a1 = SiLU(z1), sd1 = SiLU'(z1) (kept in registers)
            {
                float aa[16], rr[16];
                load16_tab(attrt + ty * H + co, hh, aa);
                load16_tab(wrhot + co, hh, rr);
#pragma unroll
                for (int r = 0; r < 16; ++r) a1[r] = pp[r] + qq[r] + fmaf(rr[r], rho, aa[r]);
            }
            // rows of this tile that are needed later (issued before the prefetch: loads return in order)
            float gMi[16];
            load16_tab(io.gM + (size_t)i * H + co, hh, gMi);
            rows_of(nxt);
            const Idx nn = idx_of(it + 2);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float sg = pvs_sigmoid(a1[r]);
                sd1[r] = pvs_silu_grad(a1[r], sg);
                a1[r] *= sg;
            }
            xwrite_block<HB>(T0, j, hh, cb, a1);
            if (cb == 0 && hh == 0) rowbuf[j] = i;
            __syncthreads();                                                     // (1) T0 complete
            // ---- z2 = W2 a1 + b2 (own output block) ----
            float dz2[16], m[16];
            float m_new[ERES ? 16 : 1], mp[ERES ? 16 : 1];
            {
                f32x16 acc2;
                float bias[16];
                load16_tab(b2t + co, hh, bias);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[r] = bias[r];
                chain_team<HB, false>(W2n, lane, cb, a1, T0, acc2);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float z2 = acc2[r];
                    const float sg = pvs_sigmoid(z2);
                    dz2[r] = pvs_silu_grad(z2, sg);
                    m[r] = z2 * sg;
                    if constexpr (ERES) m_new[r] = m[r];
                }
            }
            if constexpr (ERES) {
                load16_tab(io.m_prev + (size_t)ee * H + co, hh, mp);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    m[r] = fmaf(res_a, m_new[r], res_b * mp[r]);
                }
            }
            xwrite_block<HB>(T1, j, hh, cb, m);
            // ---- gradient wrt m (own block): external + attention + coordinate branch ----
            f32x16 gm;
            {
                float init[16];
                if (io.g_m_out) load16_tab(io.g_m_out + (size_t)ee * H + co, hh, init);
#pragma unroll
                for (int r = 0; r < 16; ++r) gm[r] = io.g_m_out ? init[r] * vm : 0.f;
            }
            if constexpr (EATT) {
                float pl = dot16_tab(wat + co, hh, m);
                float pdot = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) pdot = fmaf(m[r], gMi[r], pdot);
                pdot += __shfl_xor(pdot, 32, 64);
                if (hh == 0) { pdA[cb * kTile + j] = pl; pdB[cb * kTile + j] = pdot; }
            }
            __syncthreads();                                                     // (2) T1, pd complete
            float g_l = 0.f, aval = 1.f;
            if constexpr (EATT) {
                const float logit = sum_pd(pdA) + bac;
                const float dot = sum_pd(pdB);
                aval = io.att[ee];
                g_l = (flags & PVS_SOFTMAX_ATT) ? aval * (dot - io.softD[i]) * vm      // softD = M_i . g_M_i
                                             : pvs_att_act_grad(att_act, logit, aval) * dot * vm;
                if (hh == 0 && cb == 0) { g_ba += g_l; glb[j] = g_l; }
                float wax[16];
                load16_tab(wat + co, hh, wax);
#pragma unroll
                for (int r = 0; r < 16; ++r) gm[r] += (aval * vm) * gMi[r] + g_l * wax[r];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) gm[r] += vm * gMi[r];
            }
            float s_coord = 0.f, nrm = 1.f, gT0 = 0.f, gT1 = 0.f, gT2 = 0.f;
            if (upd) {
                gT0 = io.gxagg[3 * i]; gT1 = io.gxagg[3 * i + 1]; gT2 = io.gxagg[3 * i + 2];
                f32x16 accc;
                float bias2[16];
                load16_tab(bc1t + co, hh, bias2);
#pragma unroll
                for (int r = 0; r < 16; ++r) accc[r] = bias2[r];
                if constexpr (WC1G) chain_team<HB, false, true>(wc1_pair, lane, cb, m, T1, accc);
                else chain_team<HB, false>(Wc1n, lane, cb, m, T1, accc);
                float q[16], dq[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float zc = accc[r];
                    const float sg = pvs_sigmoid(zc);
                    q[r] = zc * sg;
                    dq[r] = pvs_silu_grad(zc, sg);
                }
                const float ps = dot16_tab(wc2t + co, hh, q);
                __syncthreads();                                                 // (3a) pdA reads of (2) done
                if (hh == 0) pdA[cb * kTile + j] = ps;
                __syncthreads();                                                 // (3) pdA complete
                float s = sum_pd(pdA);
                float dact = 1.f;
                if (flags & PVS_TANH) { s = pvs_tanh(s); dact = 1.f - s * s; }
                if (flags & PVS_NORMALIZE) nrm = 1.f / (sqrtf(rho) + 1e-8f);
                s_coord = s;
                const float g_s = (d0 * gT0 + d1 * gT1 + d2 * gT2) * nrm * dact * vm;
                float wc2x[16], g_zc[16];
                load16_tab(wc2t + co, hh, wc2x);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    g_zc[r] = g_s * wc2x[r] * dq[r];
                    g_wc2x[r] = fmaf(g_s, q[r], g_wc2x[r]);
                }
                xwrite_block<HB>(T2, j, hh, cb, g_zc);
                __syncthreads();                                                 // (4) T2 = g_zc complete
                if constexpr (WC1G) chain_team<HB, true, true>(wc1_pair, lane, cb, g_zc, T2, gm);
                else chain_team<HB, true>(Wc1n, lane, cb, g_zc, T2, gm);
            } else if (EATT) {
                __syncthreads();                                                 // glb visible
            }
            // ---- Wc1 weight gradient (row block cb) + g_bc1 + g_wa ----
            if (upd || EATT) {
#pragma unroll
                for (int sI = 0; sI < 16; ++sI) {
                    const int el = 2 * sI + hh;
                    const float av = upd ? T2[el * TS + co + j] : 0.f;
                    float bv[HB];
#pragma unroll
                    for (int bi = 0; bi < HB; ++bi) bv[bi] = T1[el * TS + 32 * bi + j];
                    g_bc1 += av;
                    if constexpr (EATT) g_wa = fmaf(glb[el], T1[el * TS + co + j], g_wa);
                    if (upd) {
#pragma unroll
                        for (int bi = 0; bi < HB; ++bi)
                            gWc1[bi] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[bi], gWc1[bi], 0, 0, 0);
                    }
                }
            }
            // ---- edge residual, g_z2 ----
            float g_z2[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float gmv = gm[r];
                float gnew = gmv;
                if constexpr (ERES) {
                    if (flags & PVS_REZERO) {
                        gnew = gate * gmv;
                        g_gate = fmaf(gmv, m_new[r], g_gate);
                        mp[r] = gmv;
                    } else if (flags & PVS_GATED_RESIDUAL) {
                        gnew = gate * gmv;
                        if (gate_raw > 0.f) g_gate = fmaf(gmv, m_new[r] - mp[r], g_gate);
                        mp[r] = (1.f - gate) * gmv;
                    } else {
                        mp[r] = gmv;
                    }
                }
                g_z2[r] = gnew * dz2[r];
            }
            if constexpr (ERES) {
                if (valid) {
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq)
                        *reinterpret_cast<float4*>(io.g_m_prev + (size_t)e * H + co + 8 * gq + 4 * hh) =
                            make_float4(mp[4 * gq], mp[4 * gq + 1], mp[4 * gq + 2], mp[4 * gq + 3]);
                }
            }
            __syncthreads();                                                     // (5) T2 / T1 reads done
            xwrite_block<HB>(T2, j, hh, cb, g_z2);
            __syncthreads();                                                     // (6) T2 = g_z2 complete
            // ---- g_a1 = W2^T g_z2 (own block); g_z1 = g_a1 * SiLU'(z1) ----
            f32x16 ga1;
#pragma unroll
            for (int r = 0; r < 16; ++r) ga1[r] = 0.f;
            chain_team<HB, true>(W2n, lane, cb, g_z2, T2, ga1);
            float g_z1[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) g_z1[r] = ga1[r] * sd1[r];
            const float prho = dot16_tab(wrhot + co, hh, g_z1);
            if (hh == 0) pdA[cb * kTile + j] = prho;
            // ---- W2 weight gradient (row block cb) + g_b2 ----
#pragma unroll
            for (int sI = 0; sI < 16; ++sI) {
                const int el = 2 * sI + hh;
                const float av = T2[el * TS + co + j];
                g_b2 += av;
#pragma unroll
                for (int bi = 0; bi < HB; ++bi)
                    gW2[bi] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, T0[el * TS + 32 * bi + j], gW2[bi], 0, 0, 0);
            }
            __syncthreads();                                                     // (7) pdA complete; T1 free
            const float g_rho = sum_pd(pdA);
            const float k1 = s_coord * nrm * vm;
            const float gd0 = fmaf(k1, gT0, 2.f * d0 * g_rho);
            const float gd1 = fmaf(k1, gT1, 2.f * d1 * g_rho);
            const float gd2 = fmaf(k1, gT2, 2.f * d2 * g_rho);
            if (hh == 0 && cb == 0) {
                *reinterpret_cast<float4*>(tx + j * 4) = make_float4(gd0, gd1, gd2, 0.f);
                if (valid)
                    *reinterpret_cast<float4*>(io.gd + (size_t)e * 4) =
                        make_float4(gd0, gd1, gd2, pvs_pack_rho_type(rho, ty));
            }
            xwrite_block<HB>(T1, j, hh, cb, g_z1);
            __syncthreads();                                                     // (8) T1 = g_z1, tx complete
            // own 128-byte half-rows of g_z1 to HBM + row-side sums of the own channel block
            {
                float4 v[4], dx[4];
                int seg[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int rl = k * 8 + rsub;
                    v[k] = *reinterpret_cast<const float4*>(T1 + rl * TS + co + 4 * quad);
                    dx[k] = *reinterpret_cast<const float4*>(tx + rl * 4);
                    if (e0 + rl < e_end) pvs_store_nt(io.gz1 + (size_t)(e0 + rl) * H + co + 4 * quad, v[k]);
                    const unsigned upto = rl == 31 ? 0xffffffffu : ((2u << rl) - 1u);
                    seg[k] = __popc(bmask & upto);
                }
                unsigned bm = bmask;
                for (int sgi = 0;; ++sgi) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float mk = seg[k] == sgi ? 1.f : 0.f;
                        acc.x = fmaf(mk, v[k].x, acc.x); acc.y = fmaf(mk, v[k].y, acc.y);
                        acc.z = fmaf(mk, v[k].z, acc.z); acc.w = fmaf(mk, v[k].w, acc.w);
                        accx.x = fmaf(mk, dx[k].x, accx.x); accx.y = fmaf(mk, dx[k].y, accx.y);
                        accx.z = fmaf(mk, dx[k].z, accx.z);
                    }
                    if (bm == 0u) break;
                    flush(cur_row);
                    const int pos = __builtin_ctz(bm);
                    bm &= bm - 1u;
                    cur_row = __builtin_amdgcn_readfirstlane(rowbuf[pos]);
                }
            }
            __syncthreads();                                                     // (9) tile buffers free
            cur = nxt;
            nxt = nn;
        }
        flush(cur_row);
    }

    // ---- block reduction into one slab, fixed order ----
    const PvsSlabLayout L = pvs_slab_layout(H);
    __syncthreads();
    float* slab = smem;
    for (int i = threadIdx.x; i < L.total; i += NT) slab[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float v = g_wc2x[r];
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) v += __shfl_xor(v, o, 64);
        g_wc2x[r] = v;
    }
    g_ba += __shfl_xor(g_ba, 32, 64);
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) g_ba += __shfl_xor(g_ba, o, 64);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) g_gate += __shfl_xor(g_gate, o, 64);
    g_b2 += __shfl_xor(g_b2, 32, 64);
    g_bc1 += __shfl_xor(g_bc1, 32, 64);
    g_wa += __shfl_xor(g_wa, 32, 64);
    for (int turn = 0; turn < NW; ++turn) {
        if (wv == turn) {
#pragma unroll
            for (int bi = 0; bi < HB; ++bi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int c = co + xch(r, hh), k = 32 * bi + j;
                    slab[L.w2 + c * H + k] += gW2[bi][r];
                    slab[L.wc1 + c * H + k] += gWc1[bi][r];
                }
            if (hh == 0) {
                slab[L.b2 + co + j] += g_b2;
                slab[L.bc1 + co + j] += g_bc1;
                slab[L.wa + co + j] += g_wa;
            }
            if (j == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) slab[L.wc2 + co + xch(r, hh)] += g_wc2x[r];
            }
            if (lane == 0) { slab[L.ba] += g_ba; slab[L.gate] += g_gate; }
        }
        __syncthreads();
    }
    float* dst = io.slabs + (size_t)blockIdx.x * L.total;
    for (int i = threadIdx.x; i < L.total; i += NT) dst[i] = slab[i];
}


// dst[0][c][k] = W[c][k], dst[1][k][c] = W[c][k]  (W and W^T, dense [H][H] each)
__global__ void k_stage_weight_pair(const float* __restrict__ W, int H, float* __restrict__ dst) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < H * H; i += gridDim.x * blockDim.x) {
        const int c = i / H, k = i - c * H;
        const float v = W[i];
        dst[i] = v;
        dst[H * H + k * H + c] = v;
    }
}

}  // namespace

int pvs_edge_bwd_mfma_max_blocks(int H) { (void)H; return 512; }

// Edge backward over the CSR edge range [e_lo, e_hi) (row-aligned: the whole batch or one segment
// of whole graphs). gPQ's row part and gx_row must have been zeroed by the caller (rows without
// edges are never written). Writes *n_slabs per-block weight-gradient partials at io.slabs.
int pvs_launch_edge_bwd_mfma(hipStream_t s, int H, const PvsGraph& g, const PvsEdgeW& w, uint32_t flags,
                             int att_act, const PvsEdgeBwdIO& io, int e_lo, int e_hi, int* n_slabs) {
    PVS_REQUIRE(w.n_attr <= 3, "MFMA edge backward supports up to 3 edge classes (got %d)", w.n_attr);
    PVS_REQUIRE(H == 32 || H == 64 || H == 128, "MFMA edge backward is built for H = 32, 64, 128 (got %d)", H);
    *n_slabs = 0;
    if (e_hi <= e_lo) return 0;
    if (H == 128) {
        // The wide layer (64 < hidden <= 128, padded to 128). Default: the four-wave team on three-term fp16 products
        // (edge_bwd_wide.hip). PVS_EGNN_BF16X3=0: the same team layout on exact fp32 MFMAs - the cross-check family -
        // with W2 in LDS and Wc1 in both orientations from global memory (two fp32 128x128 matrices do not fit beside
        // the team's 51 KB of tiles). One team per 256-thread block, one block per CU.
        {
            const char* bfw = getenv("PVS_EGNN_BF16X3");
            if (!(bfw && bfw[0] == '0'))
                return pvs_launch_edge_bwd_wide(s, H, g, w, flags, att_act, io, e_lo, e_hi, n_slabs);
        }
        PVS_REQUIRE(io.wpair, "H = 128 edge backward needs the weight-pair scratch");
        const bool upd = (flags & PVS_UPDATE_COORDS) && io.gxagg != nullptr;
        if (upd) {
            k_stage_weight_pair<<<64, 256, 0, s>>>(w.wc1, H, io.wpair);
            PVS_CHECK_LAUNCH();
        }
        const int E = e_hi - e_lo;
        long long b = ((long long)E + 511) / 512;
        if (b < 1) b = 1;
        if (b > 256) b = 256;
        long long per_team = ((long long)E + b * 4096 - 1) / (b * 4096);
        if (per_team < 1) per_team = 1;
        const int blocks = (int)b, n_chunks = (int)(b * per_team);
        *n_slabs = blocks;
        PvsProfScope prof(s, PVS_PROF_EDGE_BWD);
        const PvsSlabLayout L = pvs_slab_layout(H);
        size_t tw = (size_t)H * (H + 1) + (5 + PVS_MAX_EDGE_ATTR) * H + 16 +
                    (size_t)(3 * kTile * (H + 4) + kTile * 4 + 2 * kTile + 2 * 4 * kTile);
        if (tw < (size_t)L.total) tw = L.total;
        const size_t tlds = tw * sizeof(float);
        const bool eres = (flags & PVS_EDGE_RESIDUAL) && io.m_prev != nullptr;
        const bool eatt = flags & PVS_EDGE_ATTENTION;
#define PVS_WIDE_LAUNCH(ER, EA)                                                                    \
    do {                                                                                          \
        if (set_lds(k_edge_bwd_team<4, ER, EA>, tlds)) return -2;                          \
        k_edge_bwd_team<4, ER, EA><<<blocks, 256, tlds, s>>>(g, w, flags, att_act, io, n_chunks, e_lo, e_hi, io.wpair); \
    } while (0)
        if (eres && eatt) PVS_WIDE_LAUNCH(true, true);
        else if (eres) PVS_WIDE_LAUNCH(true, false);
        else if (eatt) PVS_WIDE_LAUNCH(false, true);
        else PVS_WIDE_LAUNCH(false, false);
#undef PVS_WIDE_LAUNCH
        PVS_CHECK_LAUNCH();
        return 0;
    }
    // Default: H = 32 as three-term fp16 products (edge_bwd_f16.hip), H = 64 as one wave per 16-edge tile with
    // six-term bf16 products (edge_bwd_h64.hip). PVS_EGNN_BF16X3=0: every product as an exact fp32 MFMA
    // (v_mfma_f32_32x32x2_f32) - the kernels below, kept as the arithmetic cross-check family of the tests
    // (tests/test_gpu_properties.py); the round-1/2 split kernels they replaced are gone from the library.
    const char* bf = getenv("PVS_EGNN_BF16X3");
    const char* bf64 = getenv("PVS_EGNN_BF16X3_H64");       // (=0: the fp32 family for H = 64 only)
    if (!(bf && bf[0] == '0') && (H == 32 || !(bf64 && bf64[0] == '0'))) {
        if (H == 32) return pvs_launch_edge_bwd_f16(s, H, g, w, flags, att_act, io, e_lo, e_hi, n_slabs);
        return pvs_launch_edge_bwd_h64(s, g, w, flags, att_act, io, e_lo, e_hi, n_slabs);
    }
    const int nt = kThreads, nw = nt / 64;
    int blocks, n_chunks;
    {
        const int E = e_hi - e_lo;
        const int max_blocks = pvs_edge_bwd_mfma_max_blocks(H);       // H = 32: 2 x 256 threads per CU
        const long long per = pvs_edges_per_wave();
        long long b = ((long long)E + (long long)nw * per - 1) / ((long long)nw * per);   // fill the chip first
        if (b < 1) b = 1;
        if (b > max_blocks) b = max_blocks;
        const long long waves = b * nw;
        long long per_wave = ((long long)E + waves * 4096 - 1) / (waves * 4096);
        if (per_wave < 1) per_wave = 1;
        blocks = (int)b;
        n_chunks = (int)(waves * per_wave);
    }
    *n_slabs = blocks;
    PvsProfScope prof(s, PVS_PROF_EDGE_BWD);
    const PvsSlabLayout L = pvs_slab_layout(H);
    size_t words = (size_t)2 * H * (H + 1) + (5 + PVS_MAX_EDGE_ATTR) * H +
                   (size_t)nw * (3 * kTile * (H + 4) + kTile * 4 + 2 * kTile);
    if (words < (size_t)L.total) words = L.total;
    const size_t lds = words * sizeof(float);
    const bool eres = (flags & PVS_EDGE_RESIDUAL) && io.m_prev != nullptr;
    const bool eatt = flags & PVS_EDGE_ATTENTION;
    if (H == 32) {
#define PVS_BWD_LAUNCH(ER, EA)                                                                      \
    do {                                                                                           \
        if (set_lds(k_edge_bwd_mfma<1, ER, EA>, lds)) return -2;                            \
        k_edge_bwd_mfma<1, ER, EA><<<blocks, nt, lds, s>>>(g, w, flags, att_act, io, n_chunks, e_lo, e_hi); \
    } while (0)
        if (eres && eatt) PVS_BWD_LAUNCH(true, true);
        else if (eres) PVS_BWD_LAUNCH(true, false);
        else if (eatt) PVS_BWD_LAUNCH(false, true);
        else PVS_BWD_LAUNCH(false, false);
#undef PVS_BWD_LAUNCH
    } else {
        // team kernel: one team of 2 waves per 128-thread block, two blocks per CU: one wave per SIMD
        // with the whole register file
        constexpr int kTeams = 1;
        const int E = e_hi - e_lo;
        long long b = ((long long)E + 511) / 512;
        if (b < 1) b = 1;
        if (b > 512) b = 512;
        const long long teams = b * kTeams;
        long long per_team = ((long long)E + teams * 4096 - 1) / (teams * 4096);
        if (per_team < 1) per_team = 1;
        blocks = (int)b;
        n_chunks = (int)(teams * per_team);
        *n_slabs = blocks;
        size_t tw = (size_t)2 * H * (H + 1) + (5 + PVS_MAX_EDGE_ATTR) * H + 16 +
                    (size_t)kTeams * (3 * kTile * (H + 4) + kTile * 4 + 2 * kTile + 2 * 2 * kTile);
        if (tw < (size_t)L.total) tw = L.total;
        const size_t tlds = tw * sizeof(float);
#define PVS_TEAM_LAUNCH(ER, EA)                                                                    \
    do {                                                                                          \
        if (set_lds(k_edge_bwd_team<2, ER, EA>, tlds)) return -2;                          \
        k_edge_bwd_team<2, ER, EA><<<blocks, 128, tlds, s>>>(g, w, flags, att_act, io, n_chunks, e_lo, e_hi, nullptr); \
    } while (0)
        if (eres && eatt) PVS_TEAM_LAUNCH(true, true);
        else if (eres) PVS_TEAM_LAUNCH(true, false);
        else if (eatt) PVS_TEAM_LAUNCH(false, true);
        else PVS_TEAM_LAUNCH(false, false);
#undef PVS_TEAM_LAUNCH
    }
    PVS_CHECK_LAUNCH();
    return 0;
}

int pvs_edge_bwd_mfma_supported(int H, uint32_t flags, int n_attr) {
    if ((H != 32 && H != 64 && H != 128) || n_attr > 3) return 0;
    (void)flags;
    return 1;
}
