// Edge backward for H = 32 * HB (HB = 1, 2) with EVERY product on the bf16 matrix pipe ("bf16x3",
// edge_mfma_common.h), one wave per 32-edge tile, no inter-wave exchange.
//
// The round-1 kernels run the two weight-gradient products  gW2 = sum_e g_z2 (x) a1,
// gWc1 = sum_e g_zc (x) m  either as fp32 MFMAs (H = 32: 2048 of 3584 matrix cycles per tile, never
// co-executing with VALU work) or, at H = 64, in a TEAM of two waves that own 32 channels each and hand
// every activation to the partner through LDS behind a workgroup barrier (four barriers per tile; 40k
// SIMD cycles per tile against ~11k of issue work). Here both products are 6-term bf16 products over the
// EDGE index inside ONE wave. Their operands need the edge index on the k axis, i.e. the transpose of
// the X layout (edge on the lane, channels in the registers) everything else lives in:
//   * activation side (a1, m): the three bf16 parts - computed anyway for the chain products - go to
//     LDS as swizzled row-major [edge][channel] images and come back as B operands through gfx950's
//     transposing read (ds_read_b64_tr_b16);
//   * gradient side (g_zc, g_z2): transposed ON THE MATRIX CORE. A part in X layout is a valid A operand
//     (lane = row = edge, registers = k = channel), so  D = part x I  (I: the 32x32 identity in the X
//     layout's k order, a constant B operand kept in LDS) leaves  D[edge][channel]  in accumulator
//     layout = lane: channel, registers: edges - exactly the A operand of the weight-gradient product.
//     bf16 x 1.0 accumulated in fp32 is exact, so the parts survive bit for bit; 2 MFMAs + 8 v_perm
//     per part and 32-channel block. No LDS round trip, no third image slot;
//   * the bias gradients (column sums of g_zc, g_z2 over the edges) are products of the SAME transposed
//     operands with a ones column: the matrix core does the sum over the edges.
// H = 32: 512-thread workgroups, two waves per SIMD (256 registers), SiLU'(z1) parked in LDS.
// (H = 64 as one wave per SIMD with SiLU'(z1) in registers: see pvs_launch_edge_bwd_bf16.)
//
// Reference semantics: autograd of EGNNLayer.edge_model / coord_model / node_model's aggregation,
// /root/reference/point_vs/models/geometric/egnn_satorras.py:123-206 (SURVEY.md §8a "Backward spec").
#include "edge_mfma_common.h"

namespace {

// acc[bo] += sum_bi W(bo,bi) v[bi]  (TRANSPOSE: W^T), v given as its bf16 parts (B operand, X layout)
template <int HB, bool TRANSPOSE>
__device__ __forceinline__ void chain_parts(const unsigned short* __restrict__ img, int lane,
                                            const Bf16Parts (&b)[HB], f32x16 (&acc)[HB]) {
    constexpr int H = 32 * HB;
#pragma unroll
    for (int bo = 0; bo < HB; ++bo)
#pragma unroll
        for (int bi = 0; bi < HB; ++bi)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 ah = img_fragment<HB, TRANSPOSE>(img, lane, bo, bi, s);
                const bf16x8 am = img_fragment<HB, TRANSPOSE>(img + H * H, lane, bo, bi, s);
                const bf16x8 al = img_fragment<HB, TRANSPOSE>(img + 2 * H * H, lane, bo, bi, s);
                acc[bo] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b[bi].hi[s], acc[bo], 0, 0, 0);
                acc[bo] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b[bi].lo[s], acc[bo], 0, 0, 0);
                acc[bo] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b[bi].mid[s], acc[bo], 0, 0, 0);
                acc[bo] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b[bi].hi[s], acc[bo], 0, 0, 0);
                acc[bo] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b[bi].mid[s], acc[bo], 0, 0, 0);
                acc[bo] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b[bi].hi[s], acc[bo], 0, 0, 0);
            }
}

// one bf16 part of channel block blk of a [32 edges][H channels] tensor, X layout -> row-major image
template <int HB>
__device__ __forceinline__ void write_part_image(unsigned short* __restrict__ part, int j, int hh, int blk,
                                                 const bf16x8 (&p)[2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const uint4 u = __builtin_bit_cast(uint4, p[s]);
        // registers 8s..8s+3 hold channels 16s + 4hh + (0..3), registers 8s+4..8s+7 channels 16s + 8 + 4hh + (0..3)
        *reinterpret_cast<uint2*>(part + img_off<HB>(j, 32 * blk + 16 * s + 4 * hh)) = make_uint2(u.x, u.y);
        *reinterpret_cast<uint2*>(part + img_off<HB>(j, 32 * blk + 16 * s + 8 + 4 * hh)) = make_uint2(u.z, u.w);
    }
}

template <int HB>
__device__ __forceinline__ void write_image(unsigned short* __restrict__ img, int j, int hh, const Bf16Parts (&b)[HB]) {
    constexpr int kPart = 32 * 32 * HB;
#pragma unroll
    for (int blk = 0; blk < HB; ++blk) {
        write_part_image<HB>(img, j, hh, blk, b[blk].hi);
        write_part_image<HB>(img + kPart, j, hh, blk, b[blk].mid);
        write_part_image<HB>(img + 2 * kPart, j, hh, blk, b[blk].lo);
    }
}

// X-layout part (A operand: lane = edge) -> its transpose as the A operand of a product over the edge
// index (lane = channel, 8 edges per k-step in the accumulator's row order), through the matrix core.
__device__ __forceinline__ void transpose_part(const bf16x8 (&p)[2], const bf16x8 (&ident)[2], bf16x8 (&out)[2]) {
    f32x16 t;
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = 0.f;
    t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p[0], ident[0], t, 0, 0, 0);
    t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p[1], ident[1], t, 0, 0, 0);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        uint4 u;
        u.x = pvs_pack_hi16(t[8 * s + 0], t[8 * s + 1]);
        u.y = pvs_pack_hi16(t[8 * s + 2], t[8 * s + 3]);
        u.z = pvs_pack_hi16(t[8 * s + 4], t[8 * s + 5]);
        u.w = pvs_pack_hi16(t[8 * s + 6], t[8 * s + 7]);
        out[s] = __builtin_bit_cast(bf16x8, u);
    }
}

// gW[bo][bi] (D layout [c = 32bo + ch(r,hh)][k = 32bi + j]) += sum over the tile's edges of G[e][c] * Act[e][k]:
// G = the gradient tensor's parts in X layout (transposed here), Act = the activation's image.
// Also gB[bo][.][col] += sum over the tile's edges of G[e][.]: the bias gradient belonging to G, as a product
// of the SAME transposed operand with a B operand that is all ones in column `col` (`ones`, a constant in
// LDS; bf16 1.0 is exact): the matrix core does the sum over the edges that an X-layout register
// accumulator would need 16 VALU adds, 16 registers and a final cross-lane reduction for.
template <int HB>
__device__ __forceinline__ void wgrad_tile(const Bf16Parts (&g)[HB], const unsigned short* __restrict__ act_img,
                                           const unsigned* __restrict__ idt, const unsigned* __restrict__ ones,
                                           int lane, f32x16 (&gW)[HB][HB], f32x16 (&gB)[HB]) {
    constexpr int kPart = 32 * 32 * HB;
    // operands are fetched from LDS where they are used (the register file is the scarce resource), the
    // compiler may keep what fits
    bf16x8 ident[2];
    ident[0] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(idt + (0 * 64 + lane) * 4));
    ident[1] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(idt + (1 * 64 + lane) * 4));
    const bf16x8 one = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(ones + lane * 4));
    const unsigned short* hi = act_img;
    const unsigned short* mid = act_img + kPart;
    const unsigned short* lo = act_img + 2 * kPart;
    auto frag = [&](const unsigned short* part, int bi, int s) { return img_fragment<HB, true>(part, lane, bi, 0, s); };
#pragma unroll
    for (int bo = 0; bo < HB; ++bo) {
        bf16x8 y[2];
        transpose_part(g[bo].lo, ident, y);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int bi = 0; bi < HB; ++bi)
                gW[bo][bi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y[s], frag(hi, bi, s), gW[bo][bi], 0, 0, 0);
            gB[bo] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y[s], one, gB[bo], 0, 0, 0);
        }
        transpose_part(g[bo].mid, ident, y);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int bi = 0; bi < HB; ++bi) {
                gW[bo][bi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y[s], frag(mid, bi, s), gW[bo][bi], 0, 0, 0);
                gW[bo][bi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y[s], frag(hi, bi, s), gW[bo][bi], 0, 0, 0);
            }
            gB[bo] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y[s], one, gB[bo], 0, 0, 0);
        }
        transpose_part(g[bo].hi, ident, y);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int bi = 0; bi < HB; ++bi) {
                gW[bo][bi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y[s], frag(lo, bi, s), gW[bo][bi], 0, 0, 0);
                gW[bo][bi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y[s], frag(mid, bi, s), gW[bo][bi], 0, 0, 0);
                gW[bo][bi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y[s], frag(hi, bi, s), gW[bo][bi], 0, 0, 0);
            }
            gB[bo] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y[s], one, gB[bo], 0, 0, 0);
        }
    }
}

template <int HB>
struct BwdCfg {
    static constexpr int H = 32 * HB;
    static constexpr int kThreadsPerBlock = HB == 1 ? 512 : 256;          // two waves / one wave per SIMD
    static constexpr int kWavesPerBlock = kThreadsPerBlock / 64;
    static constexpr int kImgShorts = 3 * 32 * H;                         // one tensor's three part images
    static constexpr bool kD1InLds = HB == 1;                             // SiLU'(z1): LDS (H=32) or registers (H=64)
    static constexpr int kWaveBytes = 2 * kImgShorts * 2 + (kD1InLds ? HB * 16 * 64 * 4 : 0);
    static constexpr int kTS = H + 4;                                     // g_z1 tile row stride (floats)
    static constexpr int kSharedBytes = 2 * 3 * H * H * 2 + (5 + PVS_MAX_EDGE_ATTR) * H * 4 + 4 * 64 * 16;
    static_assert(kTile * kTS * 4 + kTile * 16 + kTile * 4 <= kImgShorts * 2, "g_z1 tile + tx + rowbuf must fit the m image");
};

template <int HB, bool ERES, bool EATT>
__global__ void __launch_bounds__(BwdCfg<HB>::kThreadsPerBlock, HB == 1 ? 2 : 1)
k_edge_bwd_bf16(PvsGraph g, PvsEdgeW w, uint32_t flags, int att_act, PvsEdgeBwdIO io, int n_chunks, int e_lo, int e_hi) {
    using Cfg = BwdCfg<HB>;
    constexpr int H = Cfg::H, NT = Cfg::kThreadsPerBlock, NW = Cfg::kWavesPerBlock, kImgShorts = Cfg::kImgShorts;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned short* W2i = reinterpret_cast<unsigned short*>(smem);         // 3 parts x [H][H] bf16
    unsigned short* Wc1i = W2i + 3 * H * H;
    float* b2t = smem + 3 * H * H;                                         // (2 images x 3 H^2 shorts = 3 H^2 floats)
    float* bc1t = b2t + H;
    float* wc2t = bc1t + H;
    float* wat = wc2t + H;
    float* wrhot = wat + H;
    float* attrt = wrhot + H;                                  // [PVS_MAX_EDGE_ATTR][H]
    unsigned* idt = reinterpret_cast<unsigned*>(attrt + PVS_MAX_EDGE_ATTR * H);   // [2 k-steps][64 lanes][4]
    unsigned* ones0 = idt + 2 * 64 * 4;             // [64 lanes][4]: bf16 ones for the lanes of column 0 (g_bc1)
    unsigned* ones1 = ones0 + 64 * 4;               // ... of column 1 (g_b2)
    char* wave_base = reinterpret_cast<char*>(ones1 + 64 * 4);

    const bool upd = (flags & PVS_UPDATE_COORDS) && io.gxagg != nullptr;

    stage_weights_img<HB>(W2i, w.w2);
    if (upd) stage_weights_img<HB>(Wc1i, w.wc1);
    for (int c = threadIdx.x; c < H; c += NT) {
        b2t[c] = w.b2[c];
        bc1t[c] = upd ? w.bc1[c] : 0.f;
        wc2t[c] = upd ? w.wc2[c] : 0.f;
        wat[c] = EATT ? w.wa[c] : 0.f;
        wrhot[c] = w.w1[c * w.ld1 + w.off_rho];
        for (int t = 0; t < PVS_MAX_EDGE_ATTR; ++t)
            attrt[t * H + c] = t < w.n_attr ? w.w1[c * w.ld1 + w.off_rho + 1 + t] : 0.f;
    }
    // identity in the X layout's k order as a B operand: lane (col, hh), k-step s, element j' is
    // 1.0 iff channel ch(8s + j', hh) == col
    for (int i = threadIdx.x; i < 2 * 64 * 4; i += NT) {
        const int q = i & 3, l = (i >> 2) & 63, s = i >> 8;
        const int col = l & 31, lh = l >> 5;
        const unsigned lo16 = xch(8 * s + 2 * q, lh) == col ? 0x3f80u : 0u;
        const unsigned hi16 = xch(8 * s + 2 * q + 1, lh) == col ? 0x3f80u : 0u;
        idt[i] = lo16 | (hi16 << 16);
    }
    for (int i = threadIdx.x; i < 64 * 4; i += NT) {
        const int col = (i >> 2) & 31;
        ones0[i] = col == 0 ? 0x3f803f80u : 0u;
        ones1[i] = col == 1 ? 0x3f803f80u : 0u;
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int j = lane & 31, hh = lane >> 5;
    unsigned short* A1I = reinterpret_cast<unsigned short*>(wave_base + wv * Cfg::kWaveBytes);
    unsigned short* MI = A1I + kImgShorts;
    float* d1b = reinterpret_cast<float*>(MI + kImgShorts);    // SiLU'(z1), X layout, lane-private (H = 32 only)
    // once the m image is dead (after the Wc1 weight gradient) its slot holds the g_z1 tile
    float* T1 = reinterpret_cast<float*>(MI);
    float* tx = T1 + kTile * Cfg::kTS;
    int* rowbuf = reinterpret_cast<int*>(tx + kTile * 4);

    const float bac = EATT ? w.ba[0] : 0.f;
    float gate_raw = 0.f, gate = 1.f;
    if (ERES && (flags & (PVS_REZERO | PVS_GATED_RESIDUAL))) {
        gate_raw = w.edge_gate[0];
        gate = (flags & PVS_GATED_RESIDUAL) ? fmaxf(gate_raw, 0.f) : gate_raw;
    }
    const float res_a = (flags & (PVS_REZERO | PVS_GATED_RESIDUAL)) ? gate : 1.f;
    const float res_b = (flags & PVS_GATED_RESIDUAL) ? 1.f - gate : 1.f;

    // ---- accumulators that live for the whole kernel ----
    f32x16 gW2[HB][HB], gWc1[HB][HB];          // D layout: [c = 32bo + ch(r,hh)][k = 32bi + j]
    f32x16 gB[HB];                             // column 0: g_bc1, column 1: g_b2 (rows = channels, D layout)
    float g_wc2x[HB][16];                      // X layout (channel in the register, edges on lanes)
    float g_wax[EATT ? HB : 1][16];
#pragma unroll
    for (int bo = 0; bo < HB; ++bo) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { gB[bo][r] = 0.f; g_wc2x[bo][r] = 0.f; }
#pragma unroll
        for (int bi = 0; bi < HB; ++bi)
#pragma unroll
            for (int r = 0; r < 16; ++r) { gW2[bo][bi][r] = 0.f; gWc1[bo][bi][r] = 0.f; }
    }
#pragma unroll
    for (int b = 0; b < (EATT ? HB : 1); ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) g_wax[b][r] = 0.f;
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
        TileIdx I = load_tile_idx(g, w.n_attr, e_begin, e_begin, e_end, j);
        for (int e0 = e_begin; e0 < e_end; e0 += kTile) {
            const int e_next = (e0 + kTile < e_end) ? e0 + kTile : e0;
            const TileIdx In = load_tile_idx(g, w.n_attr, e_next, e_begin, e_end, j);
            const int e = I.e, ee = I.ee, i = I.i, ty = I.ty;
            const bool valid = I.valid;
            const float vm = valid ? 1.f : 0.f;
            const unsigned bmask = (unsigned)__ballot(valid && hh == 0 && i != I.prev_row);
            float d0, d1, d2, rho;
            Bf16Parts pb[HB];                 // parts of the tensor being pushed through a product
            float d1r[Cfg::kD1InLds ? 1 : HB][16];   // SiLU'(z1) in registers (H = 64)

            // ---- recompute: z1, a1 = SiLU(z1), SiLU'(z1); a1 image; z2 = W2 a1 + b2 ----
            f32x16 acc2[HB];
            {
                TileGather<HB> G;
                gather_tile<HB>(io.PQ, io.x, I, hh, G);
                d0 = G.d0; d1 = G.d1; d2 = G.d2;
                rho = d0 * d0 + d1 * d1 + d2 * d2;
                float a1[HB][16];
                assemble_z1<HB>(G, attrt, wrhot, ty, hh, rho, a1);
#pragma unroll
                for (int b = 0; b < HB; ++b)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        float dd[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float z = a1[b][4 * gq + q];
                            const float sg = pvs_sigmoid(z);
                            const float av = z * sg;
                            dd[q] = fmaf(av, 1.0f - sg, sg);       // SiLU'(z) = s + z s (1 - s)
                            a1[b][4 * gq + q] = av;
                        }
                        if constexpr (Cfg::kD1InLds) {
                            *reinterpret_cast<float4*>(d1b + ((b * 4 + gq) * 64 + lane) * 4) = make_float4(dd[0], dd[1], dd[2], dd[3]);
                        } else {
#pragma unroll
                            for (int q = 0; q < 4; ++q) d1r[b][4 * gq + q] = dd[q];
                        }
                    }
#pragma unroll
                for (int b = 0; b < HB; ++b) split_bf16x3(a1[b], pb[b]);
                write_image<HB>(A1I, j, hh, pb);
                float bias[HB][16];
                load_tab<HB>(b2t, hh, bias);
#pragma unroll
                for (int b = 0; b < HB; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc2[b][r] = bias[b][r];
                chain_parts<HB, false>(W2i, lane, pb, acc2);
            }
            float dz2[HB][16], m[HB][16];     // SiLU'(z2) and the message
            float m_new[ERES ? HB : 1][16], mp[ERES ? HB : 1][16];
#pragma unroll
            for (int b = 0; b < HB; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float z2 = acc2[b][r];
                    const float sg = pvs_sigmoid(z2);
                    m[b][r] = z2 * sg;
                    dz2[b][r] = fmaf(m[b][r], 1.0f - sg, sg);
                    if constexpr (ERES) m_new[b][r] = m[b][r];
                }
            if constexpr (ERES) {
                load_x<HB>(io.m_prev + (size_t)ee * H, hh, mp);
#pragma unroll
                for (int b = 0; b < HB; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) m[b][r] = fmaf(res_a, m_new[b][r], res_b * mp[b][r]);
            }

            // ---- gradient wrt m: the coordinate branch's term comes from the matrix core first; the external,
            // aggregated-message and attention terms are added AFTER it (g_m is then not live across the
            // coordinate branch: fewer registers where the pressure peaks; and with the terms added BEFORE
            // it the attention instantiation was not run-to-run reproducible) ----
            f32x16 gm[HB];
#pragma unroll
            for (int b = 0; b < HB; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) gm[b][r] = 0.f;
            float gMi[HB][16];
            auto load_row_terms = [&]() { load_x<HB>(io.gM + (size_t)i * H, hh, gMi); };
            auto add_row_terms = [&]() {
                if (io.g_m_out) {
                    float init[HB][16];
                    load_x<HB>(io.g_m_out + (size_t)ee * H, hh, init);
#pragma unroll
                    for (int b = 0; b < HB; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r) gm[b][r] = fmaf(init[b][r], vm, gm[b][r]);
                }
                if constexpr (EATT) {
                    float wax[HB][16];
                    load_tab<HB>(wat, hh, wax);
                    float logit = 0.f, dot = 0.f;
#pragma unroll
                    for (int b = 0; b < HB; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            logit = fmaf(wax[b][r], m[b][r], logit);
                            dot = fmaf(m[b][r], gMi[b][r], dot);
                        }
                    logit += __shfl_xor(logit, 32, 64);
                    dot += __shfl_xor(dot, 32, 64);
                    logit += bac;
                    const float aval = io.att[ee];
                    const float g_l = (flags & PVS_SOFTMAX_ATT) ? aval * (dot - io.softD[i]) * vm   // softD = M_i . g_M_i
                                                                : pvs_att_act_grad(att_act, logit, aval) * dot * vm;
                    if (hh == 0) g_ba += g_l;
#pragma unroll
                    for (int b = 0; b < HB; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            gm[b][r] += (aval * vm) * gMi[b][r] + g_l * wax[b][r];
                            g_wax[b][r] = fmaf(g_l, m[b][r], g_wax[b][r]);
                        }
                } else {
#pragma unroll
                    for (int b = 0; b < HB; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r) gm[b][r] = fmaf(vm, gMi[b][r], gm[b][r]);
                }
            };
            float s_coord = 0.f, nrm = 1.f;
            float gT0 = 0.f, gT1 = 0.f, gT2 = 0.f;
            if (upd) {
                gT0 = io.gxagg[3 * i]; gT1 = io.gxagg[3 * i + 1]; gT2 = io.gxagg[3 * i + 2];
                f32x16 accc[HB];
                {
                    float bias2[HB][16];
                    load_tab<HB>(bc1t, hh, bias2);
#pragma unroll
                    for (int b = 0; b < HB; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r) accc[b][r] = bias2[b][r];
                }
#pragma unroll
                for (int b = 0; b < HB; ++b) split_bf16x3(m[b], pb[b]);
                write_image<HB>(MI, j, hh, pb);
                chain_parts<HB, false>(Wc1i, lane, pb, accc);         // zc = Wc1 m + bc1
                float wc2x[HB][16];
                load_tab<HB>(wc2t, hh, wc2x);
                float q[HB][16], dq[HB][16];
                float s = 0.f;
#pragma unroll
                for (int b = 0; b < HB; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float zc = accc[b][r];
                        const float sg = pvs_sigmoid(zc);
                        q[b][r] = zc * sg;
                        dq[b][r] = fmaf(q[b][r], 1.0f - sg, sg);
                        s = fmaf(wc2x[b][r], q[b][r], s);
                    }
                s += __shfl_xor(s, 32, 64);
                float dact = 1.f;
                if (flags & PVS_TANH) { s = pvs_tanh(s); dact = 1.f - s * s; }
                if (flags & PVS_NORMALIZE) nrm = 1.f / (sqrtf(rho) + 1e-8f);
                s_coord = s;
                const float g_s = (d0 * gT0 + d1 * gT1 + d2 * gT2) * nrm * dact * vm;
                float g_zc[HB][16];
#pragma unroll
                for (int b = 0; b < HB; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        g_zc[b][r] = g_s * wc2x[b][r] * dq[b][r];
                        g_wc2x[b][r] = fmaf(g_s, q[b][r], g_wc2x[b][r]);
                    }
#pragma unroll
                for (int b = 0; b < HB; ++b) split_bf16x3(g_zc[b], pb[b]);
                chain_parts<HB, true>(Wc1i, lane, pb, gm);            // g_m += Wc1^T g_zc
                pvs_wave_lds_sync();                                  // the m image is complete
                wgrad_tile<HB>(pb, MI, idt, ones0, lane, gWc1, gB);   // gWc1 += g_zc (x) m ; g_bc1 += sum_e g_zc
                load_row_terms();
            } else {
                load_row_terms();
            }
            add_row_terms();
            // ---- edge residual; g_z2 = g_m_new * SiLU'(z2) ----
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
            // ---- g_a1 = W2^T g_z2 ; gW2 += g_z2 (x) a1 ; g_b2 += sum_e g_z2 ; g_z1 = g_a1 * SiLU'(z1) ----
#pragma unroll
            for (int b = 0; b < HB; ++b) split_bf16x3(g_z2[b], pb[b]);
            f32x16 ga1[HB];
#pragma unroll
            for (int b = 0; b < HB; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) ga1[b][r] = 0.f;
            chain_parts<HB, true>(W2i, lane, pb, ga1);
            if (!upd) pvs_wave_lds_sync();                            // (a1 image: no earlier sync on this path)
            wgrad_tile<HB>(pb, A1I, idt, ones1, lane, gW2, gB);
            float g_z1[HB][16];
#pragma unroll
            for (int b = 0; b < HB; ++b)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    float4 dd;
                    if constexpr (Cfg::kD1InLds) {
                        dd = *reinterpret_cast<const float4*>(d1b + ((b * 4 + gq) * 64 + lane) * 4);
                    } else {
                        dd = make_float4(d1r[b][4 * gq], d1r[b][4 * gq + 1], d1r[b][4 * gq + 2], d1r[b][4 * gq + 3]);
                    }
                    g_z1[b][4 * gq] = ga1[b][4 * gq] * dd.x;
                    g_z1[b][4 * gq + 1] = ga1[b][4 * gq + 1] * dd.y;
                    g_z1[b][4 * gq + 2] = ga1[b][4 * gq + 2] * dd.z;
                    g_z1[b][4 * gq + 3] = ga1[b][4 * gq + 3] * dd.w;
                }
            const float g_rho = dot_tab<HB>(wrhot, hh, g_z1);
            const float k1 = s_coord * nrm * vm;
            const float gd0 = fmaf(k1, gT0, 2.f * d0 * g_rho);
            const float gd1 = fmaf(k1, gT1, 2.f * d1 * g_rho);
            const float gd2 = fmaf(k1, gT2, 2.f * d2 * g_rho);
            pvs_wave_lds_sync();          // every read of the m image (its slot becomes the g_z1 tile) is done
            // per edge: grad wrt (x_row - x_col) and rho, 16 B, for the node gather kernel
            if (hh == 0) {
                *reinterpret_cast<float4*>(tx + j * 4) = make_float4(gd0, gd1, gd2, 0.f);
                rowbuf[j] = i;
                if (valid)
                    pvs_store_nt(io.gd + (size_t)e * 4, make_float4(gd0, gd1, gd2, pvs_pack_rho_type(rho, ty)));
            }
            // ---- g_z1 edge-major, then whole rows to HBM + the row-side sums from the same reads ----
#pragma unroll
            for (int b = 0; b < HB; ++b)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
                    *reinterpret_cast<float4*>(T1 + j * Cfg::kTS + 32 * b + 8 * gq + 4 * hh) =
                        make_float4(g_z1[b][4 * gq], g_z1[b][4 * gq + 1], g_z1[b][4 * gq + 2], g_z1[b][4 * gq + 3]);
            pvs_wave_lds_sync();
            reduce_rows_tile<HB>(T1, tx, rowbuf, bmask, lane, acc, accx, cur_row, flush,
                                 [&](int rl, int q, const float4& v) {
                                     if (e0 + rl < e_end)   // streamed once: non-temporal
                                         pvs_store_nt(io.gz1 + (size_t)(e0 + rl) * H + 4 * q, v);
                                 });
            I = In;
            pvs_wave_lds_sync();
        }
        flush(cur_row);
    }

    // ---- block reduction into one slab, fixed order ----
    const PvsSlabLayout L = pvs_slab_layout(H);
    __syncthreads();
    float* slab = smem;
    for (int i = threadIdx.x; i < L.total; i += NT) slab[i] = 0.f;
    __syncthreads();
    // X-layout vectors: sum over the 32 edge lanes of each half
    auto lanes32 = [](float v) {
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) v += __shfl_xor(v, o, 64);
        return v;
    };
#pragma unroll
    for (int b = 0; b < HB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) g_wc2x[b][r] = lanes32(g_wc2x[b][r]);
#pragma unroll
    for (int b = 0; b < (EATT ? HB : 1); ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) g_wax[b][r] = lanes32(g_wax[b][r]);
    g_ba += __shfl_xor(g_ba, 32, 64);          // only hh == 0 lanes accumulated
    g_ba = lanes32(g_ba);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) g_gate += __shfl_xor(g_gate, o, 64);
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
            if (j == 0) {
#pragma unroll
                for (int b = 0; b < HB; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int c = 32 * b + xch(r, hh);
                        slab[L.wc2 + c] += g_wc2x[b][r];
                        if constexpr (EATT) slab[L.wa + c] += g_wax[b][r];
                    }
            }
            if (j <= 1) {      // bias gradients: column 0 of gB is g_bc1, column 1 is g_b2
#pragma unroll
                for (int b = 0; b < HB; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) slab[(j == 0 ? L.bc1 : L.b2) + 32 * b + xch(r, hh)] += gB[b][r];
            }
            if (lane == 0) { slab[L.ba] += g_ba; slab[L.gate] += g_gate; }
        }
        __syncthreads();
    }
    float* dst = io.slabs + (size_t)blockIdx.x * L.total;
    for (int i = threadIdx.x; i < L.total; i += NT) dst[i] = slab[i];
}

template <int HB>
int launch_bwd_bf16(hipStream_t s, const PvsGraph& g, const PvsEdgeW& w, uint32_t flags, int att_act,
                    const PvsEdgeBwdIO& io, int e_lo, int e_hi, int* n_slabs) {
    using Cfg = BwdCfg<HB>;
    constexpr int nw = Cfg::kWavesPerBlock;
    int blocks, n_chunks;
    {
        const int E = e_hi - e_lo;
        long long b = ((long long)E + (long long)nw * 512 - 1) / ((long long)nw * 512);   // fill the chip first
        if (b < 1) b = 1;
        if (b > 256) b = 256;                      // one workgroup per CU (LDS)
        const long long waves = b * nw;
        long long per_wave = ((long long)E + waves * 4096 - 1) / (waves * 4096);
        if (per_wave < 1) per_wave = 1;
        blocks = (int)b;
        n_chunks = (int)(waves * per_wave);
    }
    *n_slabs = blocks;
    PvsProfScope prof(s, PVS_PROF_EDGE_BWD);
    const PvsSlabLayout L = pvs_slab_layout(Cfg::H);
    size_t lds = (size_t)Cfg::kSharedBytes + (size_t)nw * Cfg::kWaveBytes;
    if (lds < (size_t)L.total * 4) lds = (size_t)L.total * 4;
    const bool eres = (flags & PVS_EDGE_RESIDUAL) && io.m_prev != nullptr;
    const bool eatt = flags & PVS_EDGE_ATTENTION;
#define PVS_BWD_BF16_LAUNCH(ER, EA)                                                                         \
    do {                                                                                                   \
        if (set_lds(k_edge_bwd_bf16<HB, ER, EA>, lds)) return -2;                                          \
        k_edge_bwd_bf16<HB, ER, EA><<<blocks, Cfg::kThreadsPerBlock, lds, s>>>(g, w, flags, att_act, io, n_chunks, \
                                                                              e_lo, e_hi);               \
    } while (0)
    if (eres && eatt) PVS_BWD_BF16_LAUNCH(true, true);
    else if (eres) PVS_BWD_BF16_LAUNCH(true, false);
    else if (eatt) PVS_BWD_BF16_LAUNCH(false, true);
    else PVS_BWD_BF16_LAUNCH(false, false);
#undef PVS_BWD_BF16_LAUNCH
    PVS_CHECK_LAUNCH();
    return 0;
}

}  // namespace

// Same contract as pvs_launch_edge_bwd_mfma (edge_mfma.hip). Built for H = 32. The kernel is written for
// H = 32 * HB and its HB = 2 instantiation (one wave per SIMD, 512 registers) is correct (parity and
// reproducibility suites) but needs ~650 registers - 128 of weight-gradient accumulators alone - and runs
// cfg3's backward at 28.7 ms per step against 16.8 ms for the two-wave team kernel: measured once with
// -DPVS_BWD_BF16_H64, not built by default.
int pvs_launch_edge_bwd_bf16(hipStream_t s, int H, const PvsGraph& g, const PvsEdgeW& w, uint32_t flags, int att_act,
                             const PvsEdgeBwdIO& io, int e_lo, int e_hi, int* n_slabs) {
    PVS_REQUIRE(w.n_attr <= 3, "MFMA edge backward supports up to 3 edge classes (got %d)", w.n_attr);
    *n_slabs = 0;
    if (e_hi <= e_lo) return 0;
#ifdef PVS_BWD_BF16_H64
    if (H == 64) return launch_bwd_bf16<2>(s, g, w, flags, att_act, io, e_lo, e_hi, n_slabs);
#endif
    PVS_REQUIRE(H == 32, "all-bf16 one-wave-per-tile edge backward is built for H = 32 (got %d)", H);
    return launch_bwd_bf16<1>(s, g, w, flags, att_act, io, e_lo, e_hi, n_slabs);
}
