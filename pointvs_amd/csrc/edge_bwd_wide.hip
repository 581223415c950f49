// Edge backward for H = 128 (hidden sizes 65..128, zero-padded by the caller): the H = 32 f16x2 kernel's arithmetic
// (edge_bwd_f16.hip: every product as three fp16 terms with per-tile power-of-two scales, weight-gradient operands
// through transposing LDS reads) spread over a TEAM of four waves per 32-edge tile. Wave cb owns channel block cb of
// every tensor - its elementwise work, its 32 output channels of each chain product, row block cb of both weight
// gradients - so a wave carries the H = 32 kernel's per-tensor footprint (16 values per lane) plus 128 accumulator
// registers, alone on its SIMD. What the team shares goes through LDS:
//   * every tensor that feeds a product is split by its owner and written into ONE row-major fp16 image pair
//     [32 edges][128 channels] (hi, lo); the other waves read their B operands from it with two ds_read_b64 per k-step,
//     and both operands of the weight gradients come back through ds_read_b64_tr_b16 - the owner splits once;
//   * ONE scale per tensor and tile (the maximum over the team: four words in LDS), so the three terms of every block
//     pair of a product sum in one accumulator;
//   * per-edge dots over all 128 channels (attention logit, m . g_M, the coordinate scalar, g_rho) as four partial sums.
// W2 (64 KB as hi + lo images) lives in LDS; coord_mlp.0's weight comes from GLOBAL memory, pre-arranged per launch
// in A-operand order for both orientations (k_stage_wide_weights: one 16-byte load per lane and fragment, coalesced
// 1 KB per wave; an L2 hit) - two split 128x128 matrices do not fit beside 64 KB of images.
// Replaces the four-wave fp32 team kernel (k_edge_bwd_team<4>, 384 fp32 MFMAs of 64 cycles per tile and wave) with
// 152 fp16 MFMAs of 32. Eleven workgroup barriers per tile; one team per 256-thread block, one block per CU.
//
// Reference semantics: autograd of EGNNLayer.edge_model / coord_model / node_model's aggregation,
// /root/reference/point_vs/models/geometric/egnn_satorras.py:123-206 (SURVEY.md §8a "Backward spec").
#include "edge_mfma_common.h"

namespace {

constexpr int kWH = 128;                          // (the staging kernel of the global weight copies: H = 128 only)
template <int HB>
struct WideDims {
    static constexpr int H = 32 * HB;
    static constexpr int kPart = 32 * H;          // one fp16 part image of a [32 edges][H channels] tensor (shorts)
    static constexpr int kImg = 2 * kPart;        // hi + lo
    static constexpr int kTS = H + 4;             // g_z1 tile row stride (floats)
};
// global A-operand copies of Wc1: [orientation 2][part 2][bo 4][bi 4][ks 2][64 lanes][4 words] + a header
constexpr int kWFragWords = 64 * 4;
constexpr int kWCopyWords = 2 * 2 * 4 * 4 * 2 * kWFragWords;      // 32,768 words = 128 KB
constexpr int kWHeaderWords = 4;                  // [0]: 1 / scale of Wc1 (fp32)

// One workgroup: scale from max |W|, then both orientations as f16x2 A operands (layout of stage_weights_f16x2).
__global__ void __launch_bounds__(1024) k_stage_wide_weights(const float* __restrict__ W, unsigned* __restrict__ dst) {
    __shared__ unsigned wmax;
    if (threadIdx.x == 0) wmax = 0u;
    __syncthreads();
    pvs_block_absmax(W, kWH * kWH, &wmax);
    __syncthreads();
    float inv;
    const float s = pvs_f16_scale(wmax, &inv);
    if (threadIdx.x == 0) reinterpret_cast<float*>(dst)[0] = inv;
    unsigned* body = dst + kWHeaderWords;
    // one item = (orientation, bo, bi, ks, lane, q): two fp16 pairs (hi word, lo word)
    for (int i = threadIdx.x; i < 2 * 4 * 4 * 2 * 64 * 4; i += blockDim.x) {
        const int q = i & 3, l = (i >> 2) & 63, ks = (i >> 8) & 1, bi = (i >> 9) & 3, bo = (i >> 11) & 3, o = i >> 13;
        const int row = 32 * bo + (l & 31), hh = l >> 5;
        const int k0 = 32 * bi + xch(8 * ks + 2 * q, hh), k1 = 32 * bi + xch(8 * ks + 2 * q + 1, hh);
        const float x0 = o ? W[k0 * kWH + row] : W[row * kWH + k0];
        const float x1 = o ? W[k1 * kWH + row] : W[row * kWH + k1];
        unsigned h, lo;
        pvs_f16_split2(x0, x1, s, h, lo);
        const int frag = (((o * 2 + 0) * 4 + bo) * 4 + bi) * 2 + ks;            // part 0 (hi)
        const int frag_lo = (((o * 2 + 1) * 4 + bo) * 4 + bi) * 2 + ks;         // part 1 (lo)
        body[frag * kWFragWords + l * 4 + q] = h;
        body[frag_lo * kWFragWords + l * 4 + q] = lo;
    }
}

__device__ __forceinline__ f16x8 wide_gfrag(const unsigned* __restrict__ body, int o, int part, int bo, int bi, int ks,
                                            int lane) {
    const int frag = (((o * 2 + part) * 4 + bo) * 4 + bi) * 2 + ks;
    return __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(body + frag * kWFragWords + lane * 4));
}

template <int HB, bool TRANSPOSE>
__device__ __forceinline__ f16x8 wide_frag(const unsigned short* __restrict__ part, int lane, int bo, int bi, int s) {
    return __builtin_bit_cast(f16x8, img_fragment_bits<HB, TRANSPOSE>(part, lane, bo, bi, s));
}

// own block of a tensor (X layout) -> the team's image pair
template <int HB>
__device__ __forceinline__ void wide_write_image(unsigned short* __restrict__ img, int j, int hh, int cb, const F16Parts& b) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        unsigned short* part = img + p * WideDims<HB>::kPart;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const uint4 u = __builtin_bit_cast(uint4, p ? b.lo[s] : b.hi[s]);
            *reinterpret_cast<uint2*>(part + img_off<HB>(j, 32 * cb + 16 * s + 4 * hh)) = make_uint2(u.x, u.y);
            *reinterpret_cast<uint2*>(part + img_off<HB>(j, 32 * cb + 16 * s + 8 + 4 * hh)) = make_uint2(u.z, u.w);
        }
    }
}

// acc (output block cb) += sum over the input blocks of (W s_w)(cb, bi) (v s_v)_bi, three terms each; the own block's
// parts from registers, the others from the team's image. LDSW: W as image pair in LDS (TRANSPOSE through the
// transposing read); else W from the global A-operand copies (orientation = TRANSPOSE).
template <int HB, bool TRANSPOSE, bool LDSW>
__device__ __forceinline__ void wide_chain(const unsigned short* __restrict__ wimg, const unsigned* __restrict__ wglob,
                                           const unsigned short* __restrict__ vimg, int lane, int cb,
                                           const F16Parts& own, f32x16& acc) {
    // One input block at a time, the loop NOT unrolled: unrolled, the 32 fragment loads of a product are all hoisted
    // and the wave's 512 registers overflow by 300. The GLOBAL weight fragments (an L2 hit of 500+ cycles) of the next
    // block are fetched before this block's MFMAs (16 loop-carried registers); fetching the LDS fragments ahead as
    // well costs 60 spilled registers and 15 % (2.32 against 2.01 ms per launch at k = 128, 4 graphs): not kept.
    f16x8 nah[2], nal[2];
    if constexpr (!LDSW) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            nah[s] = wide_gfrag(wglob, TRANSPOSE ? 1 : 0, 0, cb, 0, s, lane);
            nal[s] = wide_gfrag(wglob, TRANSPOSE ? 1 : 0, 1, cb, 0, s, lane);
        }
    }
#pragma unroll 1
    for (int bi = 0; bi < HB; ++bi) {
        f16x8 ah[2], al[2];
        if constexpr (!LDSW) {
            const int bn = bi + 1 < HB ? bi + 1 : bi;       // (the last block re-fetches itself: no branch)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                ah[s] = nah[s]; al[s] = nal[s];
                nah[s] = wide_gfrag(wglob, TRANSPOSE ? 1 : 0, 0, cb, bn, s, lane);
                nal[s] = wide_gfrag(wglob, TRANSPOSE ? 1 : 0, 1, cb, bn, s, lane);
            }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f16x8 bh, bl;
            if (bi == cb) { bh = own.hi[s]; bl = own.lo[s]; }
            else {
                bh = wide_frag<HB, false>(vimg, lane, 0, bi, s);
                bl = wide_frag<HB, false>(vimg + WideDims<HB>::kPart, lane, 0, bi, s);
            }
            if constexpr (LDSW) {
                ah[s] = wide_frag<HB, TRANSPOSE>(wimg, lane, cb, bi, s);
                al[s] = wide_frag<HB, TRANSPOSE>(wimg + WideDims<HB>::H * WideDims<HB>::H, lane, cb, bi, s);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bh, acc, 0, 0, 0);
        }
    }
}

// gW[bi] (rows: own channel block cb of G, columns: channel block bi of Act) += inv_w * sum over the tile's edges of
// G'[e][c] Act'[e][k]; gB[.][col] += inv_b * sum_e G'[e][.] (ones-column product). Both operands read transposed.
template <int HB>
__device__ __forceinline__ void wide_wgrad(const unsigned short* __restrict__ g_img,
                                           const unsigned short* __restrict__ act_img,
                                           const unsigned* __restrict__ ones, int lane, int cb, float inv_w, float inv_b,
                                           f32x16 (&gW)[HB], f32x16& gB) {
    const f16x8 one = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(ones + lane * 4));
    f16x8 gh[2], gl[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        gh[s] = wide_frag<HB, true>(g_img, lane, cb, 0, s);
        gl[s] = wide_frag<HB, true>(g_img + WideDims<HB>::kPart, lane, cb, 0, s);
    }
    {
        f32x16 tb;
#pragma unroll
        for (int r = 0; r < 16; ++r) tb[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            tb = __builtin_amdgcn_mfma_f32_32x32x16_f16(gl[s], one, tb, 0, 0, 0);
            tb = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh[s], one, tb, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) gB[r] = fmaf(tb[r], inv_b, gB[r]);
    }
#pragma unroll
    for (int bi = 0; bi < HB; ++bi) {
        __builtin_amdgcn_sched_barrier(0);       // (one block at a time: see wide_chain)
        f32x16 t;
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const f16x8 ah = wide_frag<HB, true>(act_img, lane, bi, 0, s);
            const f16x8 al = wide_frag<HB, true>(act_img + WideDims<HB>::kPart, lane, bi, 0, s);
            t = __builtin_amdgcn_mfma_f32_32x32x16_f16(gl[s], ah, t, 0, 0, 0);
            t = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh[s], al, t, 0, 0, 0);
            t = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh[s], ah, t, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) gW[bi][r] = fmaf(t[r], inv_w, gW[bi][r]);
    }
}

template <int HB>
struct WideCfg {
    using D = WideDims<HB>;
    static constexpr int kThreads = 64 * HB;              // one team per workgroup
    static constexpr bool kWc1Global = HB > 2;            // H = 128: coord_mlp.0's weight from global memory
    // W images (W2, and Wc1 where it fits), tables, two ones columns, weight maxima, per-tile team words, three tensor
    // images, SiLU'(z1) per wave
    static constexpr int kWBytes = (kWc1Global ? 1 : 2) * 2 * D::H * D::H * 2;
    static constexpr int kTabBytes = (5 + PVS_MAX_EDGE_ATTR) * D::H * 4;
    static constexpr int kOnesBytes = 2 * 64 * 16;
    static constexpr int kTeamBytes = 16 + 4 * 4 * 4 + 4 * HB * kTile * 4;       // wmax | tmax[4 tensors][4] | pdA..pdD
    static constexpr int kImgBytes = 3 * D::kImg * 2;
    static constexpr int kD1Bytes = HB * 16 * 64 * 4;
    static constexpr int kBytes = kWBytes + kTabBytes + kOnesBytes + kTeamBytes + kImgBytes + kD1Bytes;
    static_assert(kTile * D::kTS * 4 + kTile * 16 + kTile * 4 <= 2 * D::kImg * 2, "g_z1 tile + tx + rowbuf must fit the m + gradient images");
    static_assert(kBytes <= 160 * 1024, "LDS");
};

// ERK: edge residual kind - 0 none, 1 the plain sum m + m_prev, 2 rezero / gated (edge_bwd_f16.hip)
template <int HB, int ERK, bool EATT>
__global__ void __launch_bounds__(WideCfg<HB>::kThreads, 1)
k_edge_bwd_wide(PvsGraph g, PvsEdgeW w, uint32_t flags, int att_act, PvsEdgeBwdIO io, int n_chunks, int e_lo, int e_hi,
                const unsigned* __restrict__ wc1_glob) {
    using Cfg = WideCfg<HB>;
    using D = WideDims<HB>;
    constexpr int H = D::H, NT = Cfg::kThreads, kWImg = D::kImg, kWTS = D::kTS;
    constexpr bool WC1G = Cfg::kWc1Global;
    constexpr bool ERES = ERK != 0;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* base = reinterpret_cast<char*>(smem);
    unsigned short* W2i = reinterpret_cast<unsigned short*>(base);                  // hi, lo: [H][H] fp16 each
    unsigned short* Wc1i = W2i + 2 * H * H;                                         // (only where it fits: !WC1G)
    float* b2t = reinterpret_cast<float*>(base + Cfg::kWBytes);
    float* bc1t = b2t + H;
    float* wc2t = bc1t + H;
    float* wat = wc2t + H;
    float* wrhot = wat + H;
    float* attrt = wrhot + H;                                                       // [PVS_MAX_EDGE_ATTR][H]
    unsigned* ones0 = reinterpret_cast<unsigned*>(base + Cfg::kWBytes + Cfg::kTabBytes);   // column 0 (g_bc1)
    unsigned* ones1 = ones0 + 64 * 4;                                               // column 1 (g_b2)
    unsigned* wmax = ones1 + 64 * 4;                                                // [0]: max |W2|
    unsigned* tmax = wmax + 4;                                                      // [4 tensors][4 waves]
    float* pdA = reinterpret_cast<float*>(tmax + 16);                               // [HB][32]
    float* pdB = pdA + HB * kTile;                                                   // (one array per dot: no
    float* pdC = pdB + HB * kTile;                                                   //  barrier between a read of one
    float* pdD = pdC + HB * kTile;                                                   //  and the next dot's writes)
    unsigned short* A1I = reinterpret_cast<unsigned short*>(pdD + HB * kTile);
    unsigned short* MI = A1I + kWImg;
    unsigned short* GI = MI + kWImg;
    float* d1all = reinterpret_cast<float*>(GI + kWImg);
    // once the m and gradient images are dead (after the W2 weight gradient) their slots hold the g_z1 tile
    float* T1 = reinterpret_cast<float*>(MI);
    float* tx = T1 + kTile * kWTS;
    int* rowbuf = reinterpret_cast<int*>(tx + kTile * 4);

    const bool upd = (flags & PVS_UPDATE_COORDS) && io.gxagg != nullptr;
    const unsigned* wc1b = wc1_glob + kWHeaderWords;

    if (threadIdx.x < 4) wmax[threadIdx.x] = 0u;
    __syncthreads();
    pvs_block_absmax(w.w2, H * H, wmax);
    if (!WC1G && upd) pvs_block_absmax(w.wc1, H * H, wmax + 1);
    __syncthreads();
    float inv_sw2, inv_swc1 = 1.f;
    const float sw2 = pvs_f16_scale(wmax[0], &inv_sw2);
    stage_weights_img_f16<HB>(W2i, w.w2, sw2);
    if constexpr (WC1G) {
        if (upd) inv_swc1 = reinterpret_cast<const float*>(wc1_glob)[0];
    } else if (upd) {
        const float swc1 = pvs_f16_scale(wmax[1], &inv_swc1);
        stage_weights_img_f16<HB>(Wc1i, w.wc1, swc1);
    }
    for (int c = threadIdx.x; c < H; c += NT) {
        b2t[c] = w.b2[c];
        bc1t[c] = upd ? w.bc1[c] : 0.f;
        wc2t[c] = upd ? w.wc2[c] : 0.f;
        wat[c] = EATT ? w.wa[c] : 0.f;
        wrhot[c] = w.w1[c * w.ld1 + w.off_rho];
        for (int t = 0; t < PVS_MAX_EDGE_ATTR; ++t)
            attrt[t * H + c] = t < w.n_attr ? w.w1[c * w.ld1 + w.off_rho + 1 + t] : 0.f;
    }
    for (int i = threadIdx.x; i < 64 * 4; i += NT) {
        const int col = (i >> 2) & 31;
        ones0[i] = col == 0 ? 0x3c003c00u : 0u;      // fp16 1.0 pairs
        ones1[i] = col == 1 ? 0x3c003c00u : 0u;
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, cb = threadIdx.x >> 6;
    const int j = lane & 31, hh = lane >> 5;
    const int co = 32 * cb;                                    // first channel of this wave's block
    float* d1b = d1all + cb * (16 * 64);                       // SiLU'(z1), X layout, lane-private

    const float bac = EATT ? w.ba[0] : 0.f;
    float gate_raw = 0.f, gate = 1.f;
    if (ERK == 2 && (flags & (PVS_REZERO | PVS_GATED_RESIDUAL))) {
        gate_raw = w.edge_gate[0];
        gate = (flags & PVS_GATED_RESIDUAL) ? fmaxf(gate_raw, 0.f) : gate_raw;
    }
    const float res_a = (flags & (PVS_REZERO | PVS_GATED_RESIDUAL)) ? gate : 1.f;
    const float res_b = (flags & PVS_GATED_RESIDUAL) ? 1.f - gate : 1.f;

    // ---- accumulators that live for the whole kernel: row block cb of the two weight gradients ----
    f32x16 gW2[HB], gWc1[HB];                  // D layout: [c = co + ch(r,hh)][k = 32 bi + j]
    f32x16 gB;                                 // column 0: g_bc1, column 1: g_b2 (rows = own channels)
    float g_wc2x[16], g_wax[16];               // X layout (own channels, edges on lanes)
#pragma unroll
    for (int bi = 0; bi < HB; ++bi)
#pragma unroll
        for (int r = 0; r < 16; ++r) { gW2[bi][r] = 0.f; gWc1[bi][r] = 0.f; }
#pragma unroll
    for (int r = 0; r < 16; ++r) { gB[r] = 0.f; g_wc2x[r] = 0.f; g_wax[r] = 0.f; }
    float g_ba = 0.f, g_gate = 0.f;

    // team-wide maximum of a tensor whose own block this wave holds: slot `ts` of tmax; ONE barrier inside
    auto team_scale = [&](const float (&v)[16], int ts, float* inv) {
        const unsigned mine = pvs_wave_max_u32(__float_as_uint(pvs_absmax16(v)));
        if (lane == 0) tmax[ts * 4 + cb] = mine;
        __syncthreads();
        unsigned m4 = tmax[ts * 4];
#pragma unroll
        for (int b = 1; b < HB; ++b) m4 = max(m4, tmax[ts * 4 + b]);
        return pvs_f16_scale(__builtin_amdgcn_readfirstlane(m4), inv);
    };
    auto sum_pd = [&](const float* pd) {
        float s = 0.f;
#pragma unroll
        for (int b = 0; b < HB; ++b) s += pd[b * kTile + j];
        return s;
    };
    auto dot16 = [&](const float* tab, const float (&v)[16]) {       // tab points at this wave's channel block
        float s = 0.f;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const float4 w4 = *reinterpret_cast<const float4*>(tab + 8 * gq + 4 * hh);
            s = fmaf(w4.x, v[4 * gq], s); s = fmaf(w4.y, v[4 * gq + 1], s);
            s = fmaf(w4.z, v[4 * gq + 2], s); s = fmaf(w4.w, v[4 * gq + 3], s);
        }
        return pvs_xor32_sum(s);
    };
    auto tab16 = [&](const float* tab, float (&out)[16]) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const float4 v = *reinterpret_cast<const float4*>(tab + 8 * gq + 4 * hh);
            out[4 * gq] = v.x; out[4 * gq + 1] = v.y; out[4 * gq + 2] = v.z; out[4 * gq + 3] = v.w;
        }
    };

    for (int chunk = pvs_xcd_block(blockIdx.x, gridDim.x); chunk < n_chunks; chunk += gridDim.x) {
        const int e_begin = __builtin_amdgcn_readfirstlane(chunk_begin(g, chunk, n_chunks, e_lo, e_hi));
        const int e_end = __builtin_amdgcn_readfirstlane(chunk_begin(g, chunk + 1, n_chunks, e_lo, e_hi));
        int cur_row = -1;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), accx = acc;   // open row: lane = (row slot, quad of the own block)
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
        // a tile ends where a graph ends (one scale per operand and tile: edge_bwd_f16.hip)
        int gk = 0, gb = e_hi;
        if (g.graph_eptr) {
            int lo = 0, hi = g.n_graphs;
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (__builtin_amdgcn_readfirstlane(g.graph_eptr[mid]) <= e_begin) lo = mid; else hi = mid;
            }
            gk = lo + 1;
            gb = __builtin_amdgcn_readfirstlane(g.graph_eptr[gk]);
        }
        auto tile_end = [&](int start, int bound) { return min(min(start + kTile, e_end), bound > start ? bound : e_end); };
        int t_end = tile_end(e_begin, gb);
        for (int e0 = e_begin; e0 < e_end;) {
            const int e_this_end = t_end;
            if (g.graph_eptr)
                while (gk < g.n_graphs && gb <= e_this_end) { ++gk; gb = __builtin_amdgcn_readfirstlane(g.graph_eptr[gk]); }
            const int n_end = e_this_end < e_end ? tile_end(e_this_end, gb) : e_this_end;
            // (prefetching the next tile's indices here costs 7 spilled registers more than it hides: 1.91 against 1.87 ms)
            const TileIdx I = load_tile_idx(g, w.n_attr, e0, e_begin, e_this_end, j);
            const int e = e0 + j, i = I.i, ty = I.ty, ee = I.ee;
            const bool valid = e < e_this_end;
            const float vm = valid ? 1.f : 0.f;
            const unsigned bmask = (unsigned)__ballot(valid && hh == 0 && i != I.prev_row);
            const float d0 = io.x[3 * i] - io.x[3 * I.jn], d1 = io.x[3 * i + 1] - io.x[3 * I.jn + 1],
                        d2 = io.x[3 * i + 2] - io.x[3 * I.jn + 2];
            const float rho = d0 * d0 + d1 * d1 + d2 * d2;
            F16Parts pb;
            float inv_sa1, inv_sm = 1.f;

            // ---- recompute (own block): z1, a1 = SiLU(z1), SiLU'(z1) ----
            float a1[16];
            {
                float pp[16], qq[16], aa[16], rr[16];
                tab16(io.PQ + (size_t)i * 2 * H + co, pp);
                tab16(io.PQ + (size_t)I.jn * 2 * H + H + co, qq);
                tab16(attrt + ty * H + co, aa);
                tab16(wrhot + co, rr);
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    float dd[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int r = 4 * gq + q;
                        const float z = pp[r] + qq[r] + fmaf(rr[r], rho, aa[r]);
                        const float sg = pvs_sigmoid(z);
                        const float av = z * sg;
                        dd[q] = fmaf(av, 1.0f - sg, sg);
                        a1[r] = av;
                    }
                    *reinterpret_cast<float4*>(d1b + (gq * 64 + lane) * 4) = make_float4(dd[0], dd[1], dd[2], dd[3]);
                }
            }
            const float sa1 = team_scale(a1, 0, &inv_sa1);                        // barrier 1
            split_f16x2(a1, sa1, pb);
            wide_write_image<HB>(A1I, j, hh, cb, pb);
            __syncthreads();                                                      // barrier 2: the a1 image is complete
            // ---- z2 = W2 a1 + b2 (own output block); m, SiLU'(z2) ----
            float dz2[16], m[16];
            float m_new[ERK == 2 ? 16 : 1], mp[ERES ? 16 : 1];
            {
                f32x16 acc2;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
                wide_chain<HB, false, true>(W2i, nullptr, A1I, lane, cb, pb, acc2);
                float bias[16];
                tab16(b2t + co, bias);
                const float k2 = inv_sa1 * inv_sw2;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float z2 = fmaf(acc2[r], k2, bias[r]);
                    const float sg = pvs_sigmoid(z2);
                    m[r] = z2 * sg;
                    dz2[r] = fmaf(m[r], 1.0f - sg, sg);
                    if constexpr (ERK == 2) m_new[r] = m[r];
                }
            }
            if constexpr (ERES) {
                tab16(io.m_prev + (size_t)ee * H + co, mp);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if constexpr (ERK == 2) m[r] = fmaf(res_a, m_new[r], res_b * mp[r]);
                    else m[r] += mp[r];
                }
            }
            float gMi[16];
            tab16(io.gM + (size_t)i * H + co, gMi);
            if constexpr (EATT) {                                // partial logit and m . g_M over the own channels
                const float pl = dot16(wat + co, m);
                float pdot = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) pdot = fmaf(m[r], gMi[r], pdot);
                pdot = pvs_xor32_sum(pdot);
                if (hh == 0) { pdA[cb * kTile + j] = pl; pdB[cb * kTile + j] = pdot; }
            }
            const float sm = team_scale(m, 1, &inv_sm);                           // barrier 3 (also: pdA / pdB complete)
            split_f16x2(m, sm, pb);
            wide_write_image<HB>(MI, j, hh, cb, pb);
            float att_v = 1.f, g_l = 0.f;
            if constexpr (EATT) {
                const float logit = sum_pd(pdA) + bac;
                const float dot = sum_pd(pdB);
                const float aval = io.att[ee];
                g_l = (flags & PVS_SOFTMAX_ATT) ? aval * (dot - io.softD[i]) * vm      // softD = M_i . g_M_i
                                                : pvs_att_act_grad(att_act, logit, aval) * dot * vm;
                att_v = aval * vm;
                if (hh == 0 && cb == 0) g_ba += g_l;
#pragma unroll
                for (int r = 0; r < 16; ++r) g_wax[r] = fmaf(g_l, m[r], g_wax[r]);
            }
            __syncthreads();                                                      // barrier 4: the m image is complete
            // ---- gradient wrt m (own block): coordinate branch, then the external / aggregated / attention terms ----
            float gm[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) gm[r] = 0.f;
            float s_coord = 0.f, nrm = 1.f, gT0 = 0.f, gT1 = 0.f, gT2 = 0.f;
            if (upd) {
                gT0 = io.gxagg[3 * i]; gT1 = io.gxagg[3 * i + 1]; gT2 = io.gxagg[3 * i + 2];
                f32x16 accc;
#pragma unroll
                for (int r = 0; r < 16; ++r) accc[r] = 0.f;
                wide_chain<HB, false, !WC1G>(Wc1i, wc1b, MI, lane, cb, pb, accc);       // (Wc1 m) s_m s_wc1
                float bias2[16], wc2x[16], q[16], dq[16];
                tab16(bc1t + co, bias2);
                tab16(wc2t + co, wc2x);
                const float kc = inv_sm * inv_swc1;
                float ps = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float zc = fmaf(accc[r], kc, bias2[r]);
                    const float sg = pvs_sigmoid(zc);
                    q[r] = zc * sg;
                    dq[r] = fmaf(q[r], 1.0f - sg, sg);
                    ps = fmaf(wc2x[r], q[r], ps);
                }
                ps = pvs_xor32_sum(ps);
                if (hh == 0) pdC[cb * kTile + j] = ps;
                __syncthreads();                                                  // barrier 5: pdC complete
                float s = sum_pd(pdC);
                float dact = 1.f;
                if (flags & PVS_TANH) { s = pvs_tanh(s); dact = 1.f - s * s; }
                if (flags & PVS_NORMALIZE) nrm = 1.f / (sqrtf(rho) + 1e-8f);
                s_coord = s;
                const float g_s = (d0 * gT0 + d1 * gT1 + d2 * gT2) * nrm * dact * vm;
                float g_zc[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    g_zc[r] = g_s * wc2x[r] * dq[r];
                    g_wc2x[r] = fmaf(g_s, q[r], g_wc2x[r]);
                }
                float inv_sg;
                const float sg_ = team_scale(g_zc, 2, &inv_sg);                   // barrier 6
                split_f16x2(g_zc, sg_, pb);
                wide_write_image<HB>(GI, j, hh, cb, pb);
                __syncthreads();                                                  // barrier 7: the g_zc image is complete
                f32x16 accg;
#pragma unroll
                for (int r = 0; r < 16; ++r) accg[r] = 0.f;
                wide_chain<HB, true, !WC1G>(Wc1i, wc1b, GI, lane, cb, pb, accg);         // (Wc1^T g_zc) s_g s_wc1
                const float kg = inv_sg * inv_swc1;
#pragma unroll
                for (int r = 0; r < 16; ++r) gm[r] = accg[r] * kg;
                wide_wgrad<HB>(GI, MI, ones0, lane, cb, inv_sg * inv_sm, inv_sg, gWc1, gB);    // gWc1 += g_zc (x) m ; g_bc1
            }
            if (io.g_m_out) {
                float init[16];
                tab16(io.g_m_out + (size_t)ee * H + co, init);
#pragma unroll
                for (int r = 0; r < 16; ++r) gm[r] = fmaf(init[r], vm, gm[r]);
            }
            if constexpr (EATT) {
                float wax[16];
                tab16(wat + co, wax);
#pragma unroll
                for (int r = 0; r < 16; ++r) gm[r] += att_v * gMi[r] + g_l * wax[r];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) gm[r] = fmaf(vm, gMi[r], gm[r]);
            }
            // ---- edge residual; g_z2 = g_m_new * SiLU'(z2) ----
            float g_z2[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float gmv = gm[r];
                float gnew = gmv;
                if constexpr (ERK == 1) mp[r] = gmv;         // (plain sum: m_prev receives g_m as it is)
                if constexpr (ERK == 2) {
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
            // ---- g_a1 = W2^T g_z2 ; gW2 += g_z2 (x) a1 ; g_b2 ; g_z1 = g_a1 * SiLU'(z1) ----
            float inv_sg2;
            const float sg2 = team_scale(g_z2, 3, &inv_sg2);                      // barrier 8 (also: every read of the
            split_f16x2(g_z2, sg2, pb);                                           //   g_zc image by the team is done)
            wide_write_image<HB>(GI, j, hh, cb, pb);
            __syncthreads();                                                      // barrier 9: the g_z2 image is complete
            f32x16 ga1;
#pragma unroll
            for (int r = 0; r < 16; ++r) ga1[r] = 0.f;
            wide_chain<HB, true, true>(W2i, nullptr, GI, lane, cb, pb, ga1);
            wide_wgrad<HB>(GI, A1I, ones1, lane, cb, inv_sg2 * inv_sa1, inv_sg2, gW2, gB);
            float g_z1[16];
            const float k1g = inv_sg2 * inv_sw2;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const float4 dd = *reinterpret_cast<const float4*>(d1b + (gq * 64 + lane) * 4);
                g_z1[4 * gq] = ga1[4 * gq] * (dd.x * k1g);
                g_z1[4 * gq + 1] = ga1[4 * gq + 1] * (dd.y * k1g);
                g_z1[4 * gq + 2] = ga1[4 * gq + 2] * (dd.z * k1g);
                g_z1[4 * gq + 3] = ga1[4 * gq + 3] * (dd.w * k1g);
            }
            const float prho = dot16(wrhot + co, g_z1);
            if (hh == 0) pdD[cb * kTile + j] = prho;
            __syncthreads();               // barrier 10: pdD complete; every read of the m / gradient images is done
            const float g_rho = sum_pd(pdD);
            const float k1 = s_coord * nrm * vm;
            const float gd0 = fmaf(k1, gT0, 2.f * d0 * g_rho);
            const float gd1 = fmaf(k1, gT1, 2.f * d1 * g_rho);
            const float gd2 = fmaf(k1, gT2, 2.f * d2 * g_rho);
            if (hh == 0 && cb == 0) {
                *reinterpret_cast<float4*>(tx + j * 4) = make_float4(gd0, gd1, gd2, 0.f);
                rowbuf[j] = i;
                if (valid)
                    pvs_store_nt(io.gd + (size_t)e * 4, make_float4(gd0, gd1, gd2, pvs_pack_rho_type(rho, ty)));
            }
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
                *reinterpret_cast<float4*>(T1 + j * kWTS + co + 8 * gq + 4 * hh) =
                    make_float4(g_z1[4 * gq], g_z1[4 * gq + 1], g_z1[4 * gq + 2], g_z1[4 * gq + 3]);
            __syncthreads();                                                      // barrier 11: the g_z1 tile, tx, rowbuf
            // own 128-byte quarter-rows of g_z1 to HBM + row-side sums of the own channel block
            {
                float4 v[4], dx[4];
                int seg[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int rl = k * 8 + rsub;
                    v[k] = *reinterpret_cast<const float4*>(T1 + rl * kWTS + co + 4 * quad);
                    dx[k] = *reinterpret_cast<const float4*>(tx + rl * 4);
                    if (e0 + rl < e_this_end) pvs_store_nt(io.gz1 + (size_t)(e0 + rl) * H + co + 4 * quad, v[k]);
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
            __syncthreads();                                                      // barrier 12: the tile buffers are free
            e0 = e_this_end;
            t_end = n_end;
        }
        flush(cur_row);
    }

    // ---- block reduction into one slab, fixed order ----
    const PvsSlabLayout L = pvs_slab_layout(H);
    __syncthreads();
    float* slab = smem;
    for (int i = threadIdx.x; i < L.total; i += NT) slab[i] = 0.f;
    __syncthreads();
    auto lanes32 = [](float v) {
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) v += __shfl_xor(v, o, 64);
        return v;
    };
#pragma unroll
    for (int r = 0; r < 16; ++r) { g_wc2x[r] = lanes32(g_wc2x[r]); g_wax[r] = lanes32(g_wax[r]); }
    g_ba += __shfl_xor(g_ba, 32, 64);          // only hh == 0 lanes of wave 0 accumulated
    g_ba = lanes32(g_ba);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) g_gate += __shfl_xor(g_gate, o, 64);
    for (int turn = 0; turn < HB; ++turn) {
        if (cb == turn) {
#pragma unroll
            for (int bi = 0; bi < HB; ++bi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int c = co + xch(r, hh), k = 32 * bi + j;
                    slab[L.w2 + c * H + k] += gW2[bi][r];
                    slab[L.wc1 + c * H + k] += gWc1[bi][r];
                }
            if (j == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int c = co + xch(r, hh);
                    slab[L.wc2 + c] += g_wc2x[r];
                    if constexpr (EATT) slab[L.wa + c] += g_wax[r];
                }
            }
            if (j <= 1) {      // bias gradients: column 0 of gB is g_bc1, column 1 is g_b2
#pragma unroll
                for (int r = 0; r < 16; ++r) slab[(j == 0 ? L.bc1 : L.b2) + co + xch(r, hh)] += gB[r];
            }
            if (lane == 0) { slab[L.ba] += g_ba; slab[L.gate] += g_gate; }
        }
        __syncthreads();
    }
    float* dst = io.slabs + (size_t)blockIdx.x * L.total;
    for (int i = threadIdx.x; i < L.total; i += NT) dst[i] = slab[i];
}

}  // namespace

size_t pvs_edge_bwd_wide_scratch_floats() { return (size_t)kWHeaderWords + kWCopyWords; }

namespace {
template <int HB>
int launch_wide(hipStream_t s, const PvsGraph& g, const PvsEdgeW& w, uint32_t flags, int att_act, const PvsEdgeBwdIO& io,
                int e_lo, int e_hi, int* n_slabs) {
    using Cfg = WideCfg<HB>;
    constexpr int H = 32 * HB;
    const bool upd = (flags & PVS_UPDATE_COORDS) && io.gxagg != nullptr;
    unsigned* wglob = reinterpret_cast<unsigned*>(io.wpair);
    if (Cfg::kWc1Global && upd) {
        k_stage_wide_weights<<<1, 1024, 0, s>>>(w.wc1, wglob);
        PVS_CHECK_LAUNCH();
    }
    const PvsSlabLayout L = pvs_slab_layout(H);
    size_t lds = Cfg::kBytes;
    if (lds < (size_t)L.total * 4) lds = (size_t)L.total * 4;
    const int per_cu = (int)((160 * 1024) / lds) > 0 ? (int)((160 * 1024) / lds) : 1;     // teams resident per CU
    const int E = e_hi - e_lo;
    long long b = ((long long)E + 511) / 512;          // fill the chip first: >= 16 tiles per team
    if (b < 1) b = 1;
    if (b > 256LL * per_cu) b = 256LL * per_cu;
    long long per_team = ((long long)E + b * 4096 - 1) / (b * 4096);
    if (per_team < 1) per_team = 1;
    const int blocks = (int)b, n_chunks = (int)(b * per_team);
    *n_slabs = blocks;
    PvsProfScope prof(s, PVS_PROF_EDGE_BWD);
    const bool eres = (flags & PVS_EDGE_RESIDUAL) && io.m_prev != nullptr;
    const bool eatt = flags & PVS_EDGE_ATTENTION;
#define PVS_BWD_WIDE_LAUNCH(ER, EA)                                                                            \
    do {                                                                                                      \
        if (set_lds(k_edge_bwd_wide<HB, ER, EA>, lds)) return -2;                                             \
        k_edge_bwd_wide<HB, ER, EA><<<blocks, Cfg::kThreads, lds, s>>>(g, w, flags, att_act, io, n_chunks, e_lo, \
                                                                       e_hi, wglob);                        \
    } while (0)
    const bool gated = flags & (PVS_REZERO | PVS_GATED_RESIDUAL);
    if (eres && gated && eatt) PVS_BWD_WIDE_LAUNCH(2, true);
    else if (eres && gated) PVS_BWD_WIDE_LAUNCH(2, false);
    else if (eres && eatt) PVS_BWD_WIDE_LAUNCH(1, true);
    else if (eres) PVS_BWD_WIDE_LAUNCH(1, false);
    else if (eatt) PVS_BWD_WIDE_LAUNCH(0, true);
    else PVS_BWD_WIDE_LAUNCH(0, false);
#undef PVS_BWD_WIDE_LAUNCH
    PVS_CHECK_LAUNCH();
    return 0;
}
}  // namespace

// Same contract as pvs_launch_edge_bwd_mfma (edge_mfma.hip). H = 128; io.wpair: pvs_edge_bwd_wide_scratch_floats()
// floats of scratch. (The kernel is written for H = 32 HB; its HB = 2 form - two waves per tile, both weight matrices in
// LDS, two teams per CU - is correct, 138 GPU tests, but 50 % slower than the one-wave-per-16-edge-tile kernel of
// edge_bwd_h64.hip at cfg3: 1.51 against 1.01 ms per launch, profiles/r03_ab_h64_team_f16_rejected.txt. Not built.)
int pvs_launch_edge_bwd_wide(hipStream_t s, int H, const PvsGraph& g, const PvsEdgeW& w, uint32_t flags, int att_act,
                             const PvsEdgeBwdIO& io, int e_lo, int e_hi, int* n_slabs) {
    PVS_REQUIRE(w.n_attr <= 3, "MFMA edge backward supports up to 3 edge classes (got %d)", w.n_attr);
    PVS_REQUIRE(H == 128, "the team edge backward is built for H = 128 (got %d)", H);
    PVS_REQUIRE(io.wpair, "H = 128 edge backward needs its weight scratch");
    *n_slabs = 0;
    if (e_hi <= e_lo) return 0;
    return launch_wide<4>(s, g, w, flags, att_act, io, e_lo, e_hi, n_slabs);
}
