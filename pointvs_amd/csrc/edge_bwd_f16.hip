// Edge backward for H = 32 with every product as a THREE-term fp16 product on the matrix cores ("f16x2",
// edge_mfma_common.h), one wave per 32-edge tile, two waves per SIMD (round 3; its bf16x3 predecessor needed six terms per
// product and three parts per operand).
//
// Per 32-edge tile:
//   * an operand is split into two fp16 parts by conversions (pvs_f16_split2: scale, v_cvt_pk, two v_fma_mix, v_cvt_pk:
//     2.5 vector instructions per value since the scale multiply runs on register pairs); the power-of-two tile scale
//     rides in the split;
//   * a chain product (W2 a1, Wc1 m, Wc1^T g_zc, W2^T g_z2) is 6 MFMAs;
//   * activations AND gradients (g_zc, g_z2) go to LDS as row-major [edge][channel] fp16 images (two parts = 4 KB per
//     tensor and wave), and BOTH operands of the two weight-gradient products over the edge index come back through the
//     transposing read (ds_read_b64_tr_b16);
//   * a weight-gradient product is 6 MFMAs (+ 4 for the bias column, a product with a ones column).
// 44 MFMAs per tile. Scales: one per operand and tile; in the instantiations without edge attention they move LAZILY and
// the weight gradients accumulate in the MFMA accumulators at the images' scale (below: wgrad_tile_f16_acc); with edge
// attention every tile takes its own scale from the wave-wide maximum and the products go through a temporary
// accumulator (wgrad_tile_f16). Elementwise work (SiLU and SiLU', descales, the split's scale) is written on register
// pairs (common.h pvs_f2). What bounds the kernel, measured phase by phase: DESIGN.md §5 / §8 item 1,
// tools/tile_trace.py.
//
// Reference semantics: autograd of EGNNLayer.edge_model / coord_model / node_model's aggregation,
// /root/reference/point_vs/models/geometric/egnn_satorras.py:123-206 (SURVEY.md §8a "Backward spec").
#include "edge_mfma_common.h"
#ifndef PVS_SA_IDX
#define PVS_SA_IDX 0
#endif
#ifndef PVS_SA_GATHER
#define PVS_SA_GATHER 0
#endif
#ifndef PVS_SA_ROW
#define PVS_SA_ROW 0
#endif
#ifndef PVS_SA_STORE
#define PVS_SA_STORE 0
#endif

namespace {

// timing-only ablation (tools/variant_obj.sh -DPVS_ABL_F_SCATTER): the per-edge outputs at scattered positions of a
// 256k-edge region, the access pattern of a by-column layout
#ifdef PVS_ABL_F_SCATTER
#define PVS_ABL_SCR(e) pvs_abl_scr((e), g.n_edges)
__device__ __forceinline__ int pvs_abl_scr(int e, int E) {
    const int r = (e & ~262143) + (int)(((unsigned)(e & 262143) * 40503u) & 262143u);
    return r < E ? r : e;
}
#else
#define PVS_ABL_SCR(e) (e)
#endif

// timing-only instrumentation (tools/variant_obj.sh -DPVS_TILE_TRACE, tools/tile_trace.py): workgroup 0 writes the shader
// clock at phase boundaries of its first tiles; every point is a scheduling barrier, so the traced kernel is the
// phase-by-phase form of the shipped one (the tool reports both tile times)
#ifdef PVS_TILE_TRACE
constexpr int kTraceTiles = 128, kTracePoints = 24;
__device__ unsigned long long pvs_trace_buf[8 * kTraceTiles * kTracePoints];
#define PVS_TP(k)                                                                                        \
    do {                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        if ((k) == 1 || (k) == 21) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   /* 1: the gather has landed; 21: the previous tile's stores have */ \
        const unsigned long long t_ = __builtin_readcyclecounter();                                      \
        if (blockIdx.x == 0 && lane == 0 && tile_no < kTraceTiles)                                       \
            pvs_trace_buf[(wv * kTraceTiles + tile_no) * kTracePoints + (k)] = t_;                       \
        __builtin_amdgcn_sched_barrier(0);                                                               \
    } while (0)
#else
#define PVS_TP(k) ((void)0)
#endif

constexpr int kH = 32;
constexpr int kPartShorts = 32 * kH;              // one fp16 part image of a [32 edges][32 channels] tensor
constexpr int kImg2 = 2 * kPartShorts;            // hi + lo

template <bool TRANSPOSE>
__device__ __forceinline__ f16x8 frag_f16(const unsigned short* __restrict__ part, int lane, int s) {
    return __builtin_bit_cast(f16x8, img_fragment_bits<1, TRANSPOSE>(part, lane, 0, 0, s));
}

// acc += (W s_w) (v s_v)  (TRANSPOSE: W^T), v given as its two fp16 parts (B operand, X layout); three terms,
// the smallest first
template <bool TRANSPOSE>
__device__ __forceinline__ void chain_f16(const unsigned short* __restrict__ img, int lane, const F16Parts& b,
                                          f32x16& acc) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const f16x8 ah = frag_f16<TRANSPOSE>(img, lane, s);
        const f16x8 al = frag_f16<TRANSPOSE>(img + kH * kH, lane, s);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b.hi[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b.lo[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b.hi[s], acc, 0, 0, 0);
    }
}

// one fp16 part of a [32 edges][32 channels] tensor, X layout -> swizzled row-major image
__device__ __forceinline__ void write_part_f16(unsigned short* __restrict__ part, int j, int hh, const f16x8 (&p)[2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const uint4 u = __builtin_bit_cast(uint4, p[s]);
        // registers 8s..8s+3 hold channels 16s + 4hh + (0..3), registers 8s+4..8s+7 channels 16s + 8 + 4hh + (0..3)
#if PVS_IMG_PAIRED      // (the two chunks are neighbours in the image: img_off<1>)
        *reinterpret_cast<uint4*>(part + img_off<1>(j, 16 * s + 4 * hh)) = u;
#else
        *reinterpret_cast<uint2*>(part + img_off<1>(j, 16 * s + 4 * hh)) = make_uint2(u.x, u.y);
        *reinterpret_cast<uint2*>(part + img_off<1>(j, 16 * s + 8 + 4 * hh)) = make_uint2(u.z, u.w);
#endif
    }
}

__device__ __forceinline__ void write_image_f16(unsigned short* __restrict__ img, int j, int hh, const F16Parts& b) {
    write_part_f16(img, j, hh, b.hi);
    write_part_f16(img + kPartShorts, j, hh, b.lo);
}

// gW (D layout [c = ch(r,hh)][k = j]) += inv_w * sum over the tile's edges of G'[e][c] * Act'[e][k], and
// gB[.][col] += inv_b * sum over the tile's edges of G'[e][.]  (the bias gradient belonging to G, as a product
// with a B operand that is all ones in column `col`: the matrix core does the sum over the edges).
// G' and Act' are the scaled fp16 images of the gradient and activation tensors; both operands are read
// transposed (k = edge index). inv_w = 1 / (s_G s_Act), inv_b = 1 / s_G.
__device__ __forceinline__ void wgrad_tile_f16(const unsigned short* __restrict__ g_img,
                                               const unsigned short* __restrict__ act_img,
                                               const unsigned* __restrict__ ones, int lane, float inv_w, float inv_b,
                                               f32x16& gW, f32x16& gB) {
    f32x16 t, tb;
#pragma unroll
    for (int r = 0; r < 16; ++r) { t[r] = 0.f; tb[r] = 0.f; }
    const f16x8 one = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(ones + lane * 4));
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const f16x8 gh = frag_f16<true>(g_img, lane, s);
        const f16x8 gl = frag_f16<true>(g_img + kPartShorts, lane, s);
        const f16x8 ah = frag_f16<true>(act_img, lane, s);
        const f16x8 al = frag_f16<true>(act_img + kPartShorts, lane, s);
        t = __builtin_amdgcn_mfma_f32_32x32x16_f16(gl, ah, t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh, al, t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh, ah, t, 0, 0, 0);
        tb = __builtin_amdgcn_mfma_f32_32x32x16_f16(gl, one, tb, 0, 0, 0);
        tb = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh, one, tb, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        gW[r] = fmaf(t[r], inv_w, gW[r]);
        gB[r] = fmaf(tb[r], inv_b, gB[r]);
    }
}

// ---- weight gradients accumulated IN the matrix core's accumulator (PVS_LAZY_WSCALE, round 4) ----------------------
// wgrad_tile_f16 sums a tile into 32 temporary registers and folds them into the running sums with the tile's scales:
// 64 v_fma + 32 live registers per tile in a kernel whose register file is full. Here the running sums ARE the MFMA
// accumulators, kept at the scale of the operand images, and the images' scales move LAZILY: an operand keeps its
// power-of-two scale while the tile's largest magnitude stays within kLazyWindow binades below the scale's ceiling
// (s * max in [2^(13 - kLazyWindow), 2^14)); when it leaves the window the new scale puts it one binade below the
// ceiling, and the accumulators that carry the old scale are multiplied by the (power-of-two, exact) ratio - a rare,
// wave-uniform branch. Cost in accuracy: an element 2^-k below its tile's largest keeps 22 - max(0, k - 16 + d) bits,
// d <= kLazyWindow the distance of the tile maximum from the ceiling (d = 0 with per-tile scales).
// Scale ratios beyond 2^60 (a tile whose gradients are 2^-60 of what the accumulator holds, or the reverse) are not
// applied: the smaller side is below the fp32 resolution of the sum - the tile is skipped, or the accumulator restarts.
// Instantiations WITHOUT edge attention only: with it the bias tile gB also takes per-tile vector updates (the attention
// weight's gradient rides in its idle columns), which would wait out every MFMA that writes it (+22 % per launch), and a
// separate accumulator for those brings the spills back (5-57 VGPRs): they keep wgrad_tile_f16.
__device__ __forceinline__ void wgrad_tile_f16_acc(const unsigned short* __restrict__ g_img,
                                                   const unsigned short* __restrict__ act_img,
                                                   const unsigned* __restrict__ ones, int lane, f32x16& gW, f32x16& gB) {
    const f16x8 one = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(ones + lane * 4));
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const f16x8 gh = frag_f16<true>(g_img, lane, s);
        const f16x8 gl = frag_f16<true>(g_img + kPartShorts, lane, s);
        const f16x8 ah = frag_f16<true>(act_img, lane, s);
        const f16x8 al = frag_f16<true>(act_img + kPartShorts, lane, s);
        gW = __builtin_amdgcn_mfma_f32_32x32x16_f16(gl, ah, gW, 0, 0, 0);
        gW = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh, al, gW, 0, 0, 0);
        gW = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh, ah, gW, 0, 0, 0);
        gB = __builtin_amdgcn_mfma_f32_32x32x16_f16(gl, one, gB, 0, 0, 0);
        gB = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh, one, gB, 0, 0, 0);
    }
}

// timing-only ablations (tools/variant_obj.sh builds; never in the shipped library)
#ifdef PVS_ABL_F_NOWGRAD
#define F16_WGRAD(...) ((void)0)
#define F16_WGRAD_ACC(...) ((void)0)
#else
#define F16_WGRAD(...) wgrad_tile_f16(__VA_ARGS__)
#define F16_WGRAD_ACC(...) wgrad_tile_f16_acc(__VA_ARGS__)
#endif


struct F16Cfg {
    static constexpr int kThreadsPerBlock = 512;                          // two waves per SIMD
    static constexpr int kWavesPerBlock = kThreadsPerBlock / 64;
    static constexpr int kTS = kH + 4;                                    // g_z1 tile row stride (floats) of the padded form; the
                                                                          // BASELINE instantiation uses unpadded rows with rotated quads
    // per wave: a1 image, m image, gradient image (4 KB each), SiLU'(z1) (4 KB)
    static constexpr int kWaveBytes = 3 * kImg2 * 2 + 16 * 64 * 4;
    // shared: W2 and Wc1 images (hi + lo), tables, two ones columns, 4 words of weight maxima, one all-zero image
    // (what a weight-gradient product reads in place of an operand it must not add: pvs_rescale_acc)
    static constexpr int kSharedBytes = 2 * kImg2 * 2 + (5 + PVS_MAX_EDGE_ATTR) * kH * 4 + 2 * 64 * 16 + 16 + kImg2 * 2;
    static_assert(kTile * kTS * 4 + kTile * 16 + kTile * 4 <= 2 * kImg2 * 2, "g_z1 tile + tx + rowbuf must fit the m + gradient images");
};

// ERK: edge residual kind - 0 none; 1 the plain sum m + m_prev (nothing of the residual has to survive the tile's
// coordinate branch); 2 rezero (the gate's gradient needs the pre-residual message at the end); 3 gated (... and m_prev);
// 4 rezero or gated by the run-time flags (every kind >= 2 before round 4). Compile-time kinds took rezero from 18 / 31
// spilled VGPRs (without / with edge attention) to 2 / 9 and -11 % / -19 % per launch, gated + attention from 31 to 26 and
// -4 %; gated without attention is faster the old way (profiles/r04_variants_gated_rezero.txt).
template <int ERK, bool EATT>
__global__ void __launch_bounds__(F16Cfg::kThreadsPerBlock, 2)
k_edge_bwd_f16(PvsGraph g, PvsEdgeW w, uint32_t flags, int att_act, PvsEdgeBwdIO io, int n_chunks, int e_lo, int e_hi) {
    using Cfg = F16Cfg;
    constexpr int H = kH, NT = Cfg::kThreadsPerBlock, NW = Cfg::kWavesPerBlock;
    constexpr bool ERES = ERK != 0;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned short* W2i = reinterpret_cast<unsigned short*>(smem);         // hi, lo: [H][H] fp16 each
    unsigned short* Wc1i = W2i + kImg2;
    float* b2t = smem + kImg2;                                             // (2 images x kImg2 shorts = kImg2 floats)
    float* bc1t = b2t + H;
    float* wc2t = bc1t + H;
    float* wat = wc2t + H;
    float* wrhot = wat + H;
    float* attrt = wrhot + H;                                  // [PVS_MAX_EDGE_ATTR][H]
    unsigned* ones0 = reinterpret_cast<unsigned*>(attrt + PVS_MAX_EDGE_ATTR * H);   // [64 lanes][4]: fp16 ones, column 0 (g_bc1)
    unsigned* ones1 = ones0 + 64 * 4;               // ... column 1 (g_b2)
    unsigned* wmax = ones1 + 64 * 4;                // [0]: max |W2|, [1]: max |Wc1| (fp32 bits)
    unsigned short* ZI = reinterpret_cast<unsigned short*>(wmax + 4);      // kImg2 shorts of zeros
    char* wave_base = reinterpret_cast<char*>(ZI + kImg2);

    const bool upd = (flags & PVS_UPDATE_COORDS) && io.gxagg != nullptr;

    if (threadIdx.x < 4) wmax[threadIdx.x] = 0u;
    __syncthreads();
    pvs_block_absmax(w.w2, H * H, wmax);
    if (upd) pvs_block_absmax(w.wc1, H * H, wmax + 1);
    __syncthreads();
    float inv_sw2, inv_swc1;
    const float sw2 = pvs_f16_scale(wmax[0], &inv_sw2);
    const float swc1 = pvs_f16_scale(wmax[1], &inv_swc1);
    stage_weights_img_f16<1>(W2i, w.w2, sw2);
    if (upd) stage_weights_img_f16<1>(Wc1i, w.wc1, swc1);
    for (int c = threadIdx.x; c < H; c += NT) {
        b2t[c] = w.b2[c];
        bc1t[c] = upd ? w.bc1[c] : 0.f;
        wc2t[c] = upd ? w.wc2[c] : 0.f;
        wat[c] = EATT ? w.wa[c] : 0.f;
        wrhot[c] = w.w1[c * w.ld1 + w.off_rho];
        for (int t = 0; t < PVS_MAX_EDGE_ATTR; ++t)
            attrt[t * H + c] = t < w.n_attr ? w.w1[c * w.ld1 + w.off_rho + 1 + t] : 0.f;
    }
    for (int i = threadIdx.x; i < kImg2 / 2; i += NT) reinterpret_cast<unsigned*>(ZI)[i] = 0u;
    for (int i = threadIdx.x; i < 64 * 4; i += NT) {
        const int col = (i >> 2) & 31;
        ones0[i] = col == 0 ? 0x3c003c00u : 0u;      // fp16 1.0 pairs
        ones1[i] = col == 1 ? 0x3c003c00u : 0u;
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int j = lane & 31, hh = lane >> 5;
    unsigned short* A1I = reinterpret_cast<unsigned short*>(wave_base + wv * Cfg::kWaveBytes);
    unsigned short* MI = A1I + kImg2;
    unsigned short* GI = MI + kImg2;
    float* d1b = reinterpret_cast<float*>(GI + kImg2);         // SiLU'(z1), X layout, lane-private
    // once the m and gradient images are dead (after the W2 weight gradient) their slots hold the g_z1 tile
    float* T1 = reinterpret_cast<float*>(MI);
    // The conflict-free tile layout (pvs_tile_quad_off: four write addresses that differ by an XOR, i.e. three more
    // lane constants than base + immediate) only where the registers are there: the BASELINE instantiation. The others
    // went from 1 / 3 / 9 / 28 to 5 / 11 / 15 / 30 spilled VGPRs with it (+3...16 % per launch) and keep the padded rows.
    constexpr bool TSWZ = ERK == 0 && !EATT;
    float* tx = T1 + kTile * Cfg::kTS;
    int* rowbuf = reinterpret_cast<int*>(tx + kTile * 4);

    const float bac = EATT ? w.ba[0] : 0.f;
    float gate_raw = 0.f, gate = 1.f;
    // (kind 4: the kind is read from the flags at run time, as all of them were before round 4 - the same source)
    if (ERK == 4 && (flags & (PVS_REZERO | PVS_GATED_RESIDUAL))) {
        gate_raw = w.edge_gate[0];
        gate = (flags & PVS_GATED_RESIDUAL) ? fmaxf(gate_raw, 0.f) : gate_raw;
    }
    if constexpr (ERK == 2 || ERK == 3) {
        gate_raw = w.edge_gate[0];
        gate = ERK == 3 ? fmaxf(gate_raw, 0.f) : gate_raw;
    }
    const float res_a = ERK == 4 ? ((flags & (PVS_REZERO | PVS_GATED_RESIDUAL)) ? gate : 1.f) : (ERK >= 2 ? gate : 1.f);
    const float res_b = ERK == 4 ? ((flags & PVS_GATED_RESIDUAL) ? 1.f - gate : 1.f) : (ERK == 3 ? 1.f - gate : 1.f);

    // ---- accumulators that live for the whole kernel ----
    f32x16 gW2, gWc1;                          // D layout: [c = ch(r,hh)][k = j]
    // gB, D layout (rows = channels): column 0 = g_bc1, column 1 = g_b2 (ones-column products, wgrad_tile_f16). With
    // edge attention the OTHER 30 columns' lanes carry the attention weight's gradient g_wa as per-lane partial sums
    // (register r of lane (j, hh) is channel xch(r, hh) in the D layout and in the X layout alike, and the partial
    // sums are added over the lanes at the end anyway): the contributions of edges 0 and 1 go to lanes 2 and 3
    // through one DPP move. 16 kernel-lifetime registers less than a separate X-layout accumulator - what this
    // instantiation was spilling.
    f32x16 gB;
    float g_wc2x[16];                          // X layout (channel in the register, edges on lanes)
#pragma unroll
    for (int r = 0; r < 16; ++r) { gB[r] = 0.f; g_wc2x[r] = 0.f; gW2[r] = 0.f; gWc1[r] = 0.f; }
    float g_ba = 0.f, g_gate = 0.f;
    // (LAZY) scale exponents of the four operand images and what the accumulators carry
    constexpr bool LAZY = PVS_LAZY_WSCALE && !EATT;
    // elementwise work on register pairs (common.h pvs_f2): everywhere but gated residual + attention, the instantiation
    // with the most live values per tile (30 spilled registers instead of 26 with it, +7 % per launch)
    constexpr bool PAIR = PVS_PAIR_MATH && !(ERK == 3 && EATT);
    // (the four exponents share ONE scalar register, a byte each, 0 = not set yet: the kernel has no scalar register to
    // spare, and a spilled one costs a v_readlane / v_writelane pair per use)
    unsigned lazy_pack = 0u;
    auto lazy_scale = [&](const float (&v)[16], int slot, float* inv) {
        const int e = (int)((lazy_pack >> (8 * slot)) & 0xffu);
        LazyExp st{e ? e : -1};
        const float sc = pvs_lazy_tile_scale(v, st, inv);
        lazy_pack = (lazy_pack & ~(0xffu << (8 * slot))) | ((unsigned)st.e << (8 * slot));
        return sc;
    };
    auto lazy_e = [&](int slot) { return (int)((lazy_pack >> (8 * slot)) & 0xffu); };
    constexpr int kXa1 = 0, kXm = 1, kXg = 2, kXg2 = 3;
    AccUnits u_w2{-1}, u_b2{-1}, u_wc1{-1}, u_bc1{-1};

    const int total_waves = gridDim.x * NW;
#ifdef PVS_TILE_TRACE
    int tile_no = 0;
#endif
    for (int chunk = pvs_xcd_block(blockIdx.x, gridDim.x) * NW + wv; chunk < n_chunks; chunk += total_waves) {
        // (wave-uniform values that come out of global loads: into scalar registers, the vector file is full)
        const int e_begin = __builtin_amdgcn_readfirstlane(chunk_begin(g, chunk, n_chunks, e_lo, e_hi));
        const int e_end = __builtin_amdgcn_readfirstlane(chunk_begin(g, chunk + 1, n_chunks, e_lo, e_hi));
        int cur_row = -1;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), accx = acc;   // open row: lane = (row slot, quad)
        constexpr int QPR = H / 4;
        const int quad = lane % QPR, rsub = lane / QPR;
        auto flush = [&](int row_id) {
            if (row_id >= 0) {
                const float4 tot = sum_row_slots<1>(acc);
                if (rsub == 0) *reinterpret_cast<float4*>(PVS_SA_STORE ? pvs_off(io.gPQ + (size_t)row_id * 2 * H, 16u * quad) : io.gPQ + (size_t)row_id * 2 * H + 4 * quad) = tot;
                const float4 tx4 = sum_row_slots<1>(accx);
                if (lane == 0) {
                    io.gx_row[3 * row_id] = tx4.x;
                    io.gx_row[3 * row_id + 1] = tx4.y;
                    io.gx_row[3 * row_id + 2] = tx4.z;
                }
            }
            acc = make_float4(0.f, 0.f, 0.f, 0.f);
            accx = acc;
        };
        // A tile shares ONE power-of-two scale per operand, and two graphs' gradients can differ by many orders of
        // magnitude (a saturated loss), so a tile ends where a graph ends (PvsGraph.graph_eptr, when the caller knows
        // the batch's graphs): gb = the first graph boundary behind the current tile's start.
        int gk = 0, gb = e_hi;
        if (g.graph_eptr) {
            int lo = 0, hi = g.n_graphs;                // last k with graph_eptr[k] <= e_begin
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (__builtin_amdgcn_readfirstlane(g.graph_eptr[mid]) <= e_begin) lo = mid; else hi = mid;
            }
            gk = lo + 1;
            gb = __builtin_amdgcn_readfirstlane(g.graph_eptr[gk]);
        }
        auto tile_end = [&](int start, int bound) { return min(min(start + kTile, e_end), bound > start ? bound : e_end); };
        int t_end = tile_end(e_begin, gb);
        // only what comes out of memory is carried from tile to tile (4 registers); e / ee / valid are recomputed
        struct Loaded { int i, jn, ty, prev_row; };
        auto load_idx = [&](int start, int end) {
            const TileIdx t = PVS_SA_IDX ? load_tile_idx32(g, w.n_attr, start, e_begin, end, j) : load_tile_idx(g, w.n_attr, start, e_begin, end, j);
            return Loaded{t.i, t.jn, t.ty, t.prev_row};
        };
        Loaded I = load_idx(e_begin, t_end);
        for (int e0 = e_begin; e0 < e_end;) {
            const int e_this_end = t_end;
            PVS_TP(21);
            // the next tile: starts where this one ends; past a graph boundary the next boundary applies
            if (g.graph_eptr)    // (empty graphs repeat a boundary)
                while (gk < g.n_graphs && gb <= e_this_end) { ++gk; gb = __builtin_amdgcn_readfirstlane(g.graph_eptr[gk]); }
            const int e_next = e_this_end < e_end ? e_this_end : e0;
            const int n_end = e_this_end < e_end ? tile_end(e_this_end, gb) : e_this_end;
            const Loaded In = load_idx(e_next, n_end);
            PVS_TP(0);
            const int e = e0 + j, i = I.i, ty = I.ty;
            const bool valid = e < e_this_end;
            const int ee = min(max(valid ? e : e_this_end - 1, 0), g.n_edges - 1);   // (as load_tile_idx clamps)
            const float vm = valid ? 1.f : 0.f;
            const unsigned bmask = (unsigned)__ballot(valid && hh == 0 && i != I.prev_row);
            float d0, d1, d2, rho;
            F16Parts pb;                      // parts of the tensor being pushed through a product
            float inv_sa1, inv_sm = 1.f;      // 1 / tile scale of the a1 and m images

            // ---- recompute: z1, a1 = SiLU(z1), SiLU'(z1); a1 image; z2 = W2 a1 + b2 ----
            float z2[16];
            {
                TileGather<1> G;
                TileIdx Ig;
                Ig.i = I.i; Ig.jn = I.jn;
#ifdef PVS_ABL_F_QHOT          // timing-only: the column-side rows come from a 256 KB window (what hiding their latency could give)
                Ig.jn &= 1023;
#endif
                if (PVS_SA_GATHER) gather_tile32<1>(io.PQ, io.x, Ig, hh, G); else gather_tile<1>(io.PQ, io.x, Ig, hh, G);
                d0 = G.d0; d1 = G.d1; d2 = G.d2;
                rho = d0 * d0 + d1 * d1 + d2 * d2;
                float a1[1][16];
                assemble_z1<1>(G, attrt, wrhot, ty, hh, rho, a1);
                PVS_TP(1);
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    float dd[4];
                    if constexpr (PAIR) {
#pragma unroll
                        for (int q = 0; q < 4; q += 2) {
                            pvs_f2 av, dv;
                            pvs_silu_grad2(pvs_f2{a1[0][4 * gq + q], a1[0][4 * gq + q + 1]}, av, dv);
                            a1[0][4 * gq + q] = av.x; a1[0][4 * gq + q + 1] = av.y;
                            dd[q] = dv.x; dd[q + 1] = dv.y;
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float z = a1[0][4 * gq + q];
                            const float sg = pvs_sigmoid(z);
                            const float av = z * sg;
                            dd[q] = fmaf(av, 1.0f - sg, sg);       // SiLU'(z) = s + z s (1 - s)
                            a1[0][4 * gq + q] = av;
                        }
                    }
                    *reinterpret_cast<float4*>(d1b + (gq * 64 + lane) * 4) = make_float4(dd[0], dd[1], dd[2], dd[3]);
                }
                PVS_TP(18);
                const float sa1 = LAZY ? lazy_scale(a1[0], kXa1, &inv_sa1) : pvs_tile_scale(a1[0], &inv_sa1);
                PVS_TP(19);
                split_f16x2<PAIR>(a1[0], sa1, pb);
                PVS_TP(20);
                write_image_f16(A1I, j, hh, pb);
                PVS_TP(2);
                f32x16 acc2;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
                chain_f16<false>(W2i, lane, pb, acc2);
                float bias[1][16];
                load_tab<1>(b2t, hh, bias);
                const float k2 = inv_sa1 * inv_sw2;
                if constexpr (PAIR) {
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const pvs_f2 z = pvs_fma2(pvs_f2{acc2[r], acc2[r + 1]}, pvs_f2{k2, k2}, pvs_f2{bias[0][r], bias[0][r + 1]});
                        z2[r] = z.x; z2[r + 1] = z.y;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) z2[r] = fmaf(acc2[r], k2, bias[0][r]);
                }
                PVS_TP(3);
            }
            float dz2[16], m[1][16];          // SiLU'(z2) and the message
            float m_new[ERK >= 2 ? 16 : 1], mp[1][16];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                if constexpr (PAIR) {
                    pvs_f2 mv, dv;
                    pvs_silu_grad2(pvs_f2{z2[r], z2[r + 1]}, mv, dv);
                    m[0][r] = mv.x; m[0][r + 1] = mv.y;
                    dz2[r] = dv.x; dz2[r + 1] = dv.y;
                } else {
                    for (int t = r; t < r + 2; ++t) {
                        const float sg = pvs_sigmoid(z2[t]);
                        m[0][t] = z2[t] * sg;
                        dz2[t] = fmaf(m[0][t], 1.0f - sg, sg);
                    }
                }
                if constexpr (ERK >= 2) { m_new[r] = m[0][r]; m_new[r + 1] = m[0][r + 1]; }
            }
            if constexpr (ERES) {
                load_x<1>(io.m_prev + (size_t)ee * H, hh, mp);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if constexpr (ERK >= 2) m[0][r] = fmaf(res_a, m_new[r], res_b * mp[0][r]);
                    else m[0][r] += mp[0][r];
                }
            }

            // ---- gradient wrt m: the coordinate branch's term comes from the matrix core first; the external,
            // aggregated-message and attention terms are added AFTER it (g_m is then not live across the
            // coordinate branch) ----
            PVS_TP(4);
            float gm[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) gm[r] = 0.f;
            float gMi[1][16];
            auto load_row_terms = [&]() { if (PVS_SA_ROW) load_x<1>(pvs_off(io.gM, (unsigned)i * (4u * H) + 16u * hh), 0, gMi); else load_x<1>(io.gM + (size_t)i * H, hh, gMi); };
            // Edge attention: everything that needs the message itself - the logit, m . g_M, the gate's gradient g_l
            // and its weight gradient g_wa += g_l m - is evaluated HERE, while m is live anyway; two scalars (the gate
            // value and g_l) cross the coordinate branch and m dies with its split, as in the plain kernel. The g_M
            // row is fetched a second time behind the branch (an L1 hit: a tile's edges share their row).
            float att_v = 1.f, g_l = 0.f;
            if constexpr (EATT) {
                float logit = 0.f, dot = 0.f;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {       // (quad by quad: eight registers of operands at a time)
                    const float4 gq4 = *reinterpret_cast<const float4*>(io.gM + (size_t)i * H + 8 * gq + 4 * hh);
                    const float4 wq4 = *reinterpret_cast<const float4*>(wat + 8 * gq + 4 * hh);
                    logit = fmaf(wq4.x, m[0][4 * gq], logit); dot = fmaf(m[0][4 * gq], gq4.x, dot);
                    logit = fmaf(wq4.y, m[0][4 * gq + 1], logit); dot = fmaf(m[0][4 * gq + 1], gq4.y, dot);
                    logit = fmaf(wq4.z, m[0][4 * gq + 2], logit); dot = fmaf(m[0][4 * gq + 2], gq4.z, dot);
                    logit = fmaf(wq4.w, m[0][4 * gq + 3], logit); dot = fmaf(m[0][4 * gq + 3], gq4.w, dot);
                }
                logit = pvs_xor32_sum(logit);
                dot = pvs_xor32_sum(dot);
                logit += bac;
                const float aval = io.att[ee];
                g_l = (flags & PVS_SOFTMAX_ATT) ? aval * (dot - io.softD[i]) * vm   // softD = M_i . g_M_i
                                                : pvs_att_act_grad(att_act, logit, aval) * dot * vm;
                att_v = aval * vm;
                if (hh == 0) g_ba += g_l;
                const float g_own = j >= 2 ? g_l : 0.f;                    // this lane's own edge, columns 2..31
                const float k_23 = (j == 2 || j == 3) ? 1.f : 0.f;         // lanes that also take edges 0 / 1
                const float g_in = k_23 * __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(g_l), 0x4E, 0xf, 0xf, true));
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    // m of lane j ^ 2 (quad_perm [2,3,0,1]); only lanes 2 and 3 use it (g_in is zero elsewhere)
                    const float mt = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m[0][r]), 0x4E, 0xf, 0xf, true));
                    gB[r] = fmaf(g_own, m[0][r], gB[r]);
                    gB[r] = fmaf(g_in, mt, gB[r]);
                }
            }
            auto add_row_terms = [&]() {
                if (io.g_m_out) {
                    float init[1][16];
                    load_x<1>(io.g_m_out + (size_t)ee * H, hh, init);
#pragma unroll
                    for (int r = 0; r < 16; ++r) gm[r] = fmaf(init[0][r], vm, gm[r]);
                }
                if constexpr (EATT) {
                    float wax[1][16];
                    load_tab<1>(wat, hh, wax);
#pragma unroll
                    for (int r = 0; r < 16; ++r) gm[r] += att_v * gMi[0][r] + g_l * wax[0][r];
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) gm[r] = fmaf(vm, gMi[0][r], gm[r]);
                }
            };
            float s_coord = 0.f, nrm = 1.f;
            float gT0 = 0.f, gT1 = 0.f, gT2 = 0.f;
            if (upd) {
                const float* gT = PVS_SA_ROW ? pvs_off(io.gxagg, 12u * (unsigned)i) : io.gxagg + 3 * i;
                gT0 = gT[0]; gT1 = gT[1]; gT2 = gT[2];
                const float sm = LAZY ? lazy_scale(m[0], kXm, &inv_sm) : pvs_tile_scale(m[0], &inv_sm);
                PVS_TP(16);
                split_f16x2<PAIR>(m[0], sm, pb);
                PVS_TP(17);
                write_image_f16(MI, j, hh, pb);
                PVS_TP(5);
                f32x16 accc;
#pragma unroll
                for (int r = 0; r < 16; ++r) accc[r] = 0.f;
                chain_f16<false>(Wc1i, lane, pb, accc);               // (Wc1 m) s_m s_wc1
                float bias2[1][16], wc2x[1][16];
                load_tab<1>(bc1t, hh, bias2);
                load_tab<1>(wc2t, hh, wc2x);
                const float kc = inv_sm * inv_swc1;
                float q[16], dq[16];
                float s = 0.f;
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    if constexpr (PAIR) {
                        const pvs_f2 zc = pvs_fma2(pvs_f2{accc[r], accc[r + 1]}, pvs_f2{kc, kc}, pvs_f2{bias2[0][r], bias2[0][r + 1]});
                        pvs_f2 qv, dv;
                        pvs_silu_grad2(zc, qv, dv);
                        q[r] = qv.x; q[r + 1] = qv.y;
                        dq[r] = dv.x; dq[r + 1] = dv.y;
                    } else {
                        for (int t = r; t < r + 2; ++t) {
                            const float zc = fmaf(accc[t], kc, bias2[0][t]);   // zc = Wc1 m + bc1
                            const float sg = pvs_sigmoid(zc);
                            q[t] = zc * sg;
                            dq[t] = fmaf(q[t], 1.0f - sg, sg);
                        }
                    }
                    // (one running sum: a pair of partial sums, or a wave-uniform fast path for full tiles behind the
                    // chain, each cost this instantiation 3-5 spilled registers)
                    s = fmaf(wc2x[0][r], q[r], s);
                    s = fmaf(wc2x[0][r + 1], q[r + 1], s);
                }
                s = pvs_xor32_sum(s);
                float dact = 1.f;
                if (flags & PVS_TANH) { s = pvs_tanh(s); dact = 1.f - s * s; }
                if (flags & PVS_NORMALIZE) nrm = 1.f / (sqrtf(rho) + 1e-8f);
                s_coord = s;
                const float g_s = (d0 * gT0 + d1 * gT1 + d2 * gT2) * nrm * dact * vm;
                float g_zc[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    g_zc[r] = g_s * wc2x[0][r] * dq[r];
                    g_wc2x[r] = fmaf(g_s, q[r], g_wc2x[r]);
                }
                PVS_TP(6);
                float inv_sg;
                const float sg_ = LAZY ? lazy_scale(g_zc, kXg, &inv_sg) : pvs_tile_scale(g_zc, &inv_sg);
                split_f16x2<PAIR>(g_zc, sg_, pb);
                write_image_f16(GI, j, hh, pb);
                PVS_TP(7);
                f32x16 accg;
#pragma unroll
                for (int r = 0; r < 16; ++r) accg[r] = 0.f;
                chain_f16<true>(Wc1i, lane, pb, accg);                // (Wc1^T g_zc) s_g s_wc1
                const float kg = inv_sg * inv_swc1;
#pragma unroll
                for (int r = 0; r < 16; ++r) gm[r] = accg[r] * kg;
                PVS_TP(8);
                pvs_wave_lds_sync();                                  // the m and g_zc images are complete
                // gWc1 += g_zc (x) m ; g_bc1 += sum_e g_zc
                if constexpr (LAZY) {
                    const int what = pvs_rescale_acc(gWc1, gB, j == 0, u_wc1, u_bc1, lazy_e(kXg), lazy_e(kXm));
                    // (an operand whose product is not to be added is read from the all-zero image instead: the product
                    // itself is never inside a branch, so that its MFMAs can be scheduled among the vector work behind it)
                    F16_WGRAD_ACC(GI, (what & 1) ? MI : ZI, (what & 2) ? ones0 : reinterpret_cast<unsigned*>(ZI), lane, gWc1, gB);
                } else {
                    F16_WGRAD(GI, MI, ones0, lane, inv_sg * inv_sm, inv_sg, gWc1, gB);
                }
                load_row_terms();
            } else {
                load_row_terms();
            }
            PVS_TP(9);
            add_row_terms();
            // ---- edge residual; g_z2 = g_m_new * SiLU'(z2) ----
            float g_z2[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float gmv = gm[r];
                float gnew = gmv;
                if constexpr (ERK == 1) mp[0][r] = gmv;      // (plain sum: m_prev receives g_m as it is)
                if constexpr (ERK == 2) {           // rezero: m = m_prev + g m_new
                    gnew = gate * gmv;
                    g_gate = fmaf(gmv, m_new[r], g_gate);
                    mp[0][r] = gmv;
                }
                if constexpr (ERK == 3) {           // gated: m = relu(g) m_new + (1 - relu(g)) m_prev
                    gnew = gate * gmv;
                    if (gate_raw > 0.f) g_gate = fmaf(gmv, m_new[r] - mp[0][r], g_gate);
                    mp[0][r] = (1.f - gate) * gmv;
                }
                if constexpr (ERK == 4) {
                    if (flags & PVS_REZERO) {
                        gnew = gate * gmv;
                        g_gate = fmaf(gmv, m_new[r], g_gate);
                        mp[0][r] = gmv;
                    } else if (flags & PVS_GATED_RESIDUAL) {
                        gnew = gate * gmv;
                        if (gate_raw > 0.f) g_gate = fmaf(gmv, m_new[r] - mp[0][r], g_gate);
                        mp[0][r] = (1.f - gate) * gmv;
                    } else {
                        mp[0][r] = gmv;
                    }
                }
#ifdef PVS_MASK_GZ2          // (A/B only: what the lanes past a tile's end hold - profiles/r06_gated_residual_backward_defect.txt)
                g_z2[r] = valid ? gnew * dz2[r] : 0.f;
#else
                g_z2[r] = gnew * dz2[r];
#endif
            }
            if constexpr (ERES) {
                if (valid) store_x<1>(io.g_m_prev + (size_t)e * H, hh, mp);
            }
            // ---- g_a1 = W2^T g_z2 ; gW2 += g_z2 (x) a1 ; g_b2 += sum_e g_z2 ; g_z1 = g_a1 * SiLU'(z1) ----
            PVS_TP(10);
            float inv_sg2;
            const float sg2 = LAZY ? lazy_scale(g_z2, kXg2, &inv_sg2) : pvs_tile_scale(g_z2, &inv_sg2);
            split_f16x2<PAIR>(g_z2, sg2, pb);
            pvs_wave_lds_sync();                                      // the g_zc image has been read
            write_image_f16(GI, j, hh, pb);
            f32x16 ga1;
#pragma unroll
            for (int r = 0; r < 16; ++r) ga1[r] = 0.f;
            chain_f16<true>(W2i, lane, pb, ga1);
            PVS_TP(11);
            pvs_wave_lds_sync();                                      // the a1 and g_z2 images are complete
            if constexpr (LAZY) {
                const int what = pvs_rescale_acc(gW2, gB, j == 1, u_w2, u_b2, lazy_e(kXg2), lazy_e(kXa1));
                F16_WGRAD_ACC(GI, (what & 1) ? A1I : ZI, (what & 2) ? ones1 : reinterpret_cast<unsigned*>(ZI), lane, gW2, gB);
            } else {
                F16_WGRAD(GI, A1I, ones1, lane, inv_sg2 * inv_sa1, inv_sg2, gW2, gB);
            }
            PVS_TP(12);
            float g_z1[1][16];
            const float k1g = inv_sg2 * inv_sw2;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const float4 dd = *reinterpret_cast<const float4*>(d1b + (gq * 64 + lane) * 4);
                g_z1[0][4 * gq] = ga1[4 * gq] * (dd.x * k1g);
                g_z1[0][4 * gq + 1] = ga1[4 * gq + 1] * (dd.y * k1g);
                g_z1[0][4 * gq + 2] = ga1[4 * gq + 2] * (dd.z * k1g);
                g_z1[0][4 * gq + 3] = ga1[4 * gq + 3] * (dd.w * k1g);
            }
            const float g_rho = dot_tab<1>(wrhot, hh, g_z1);
            const float k1 = s_coord * nrm * vm;
            const float gd0 = fmaf(k1, gT0, 2.f * d0 * g_rho);
            const float gd1 = fmaf(k1, gT1, 2.f * d1 * g_rho);
            const float gd2 = fmaf(k1, gT2, 2.f * d2 * g_rho);
            PVS_TP(13);
            pvs_wave_lds_sync();          // every read of the m / gradient images (their slots become the g_z1 tile) is done
            // per edge: grad wrt (x_row - x_col) and rho, 16 B, for the node gather kernel
            if (hh == 0) {
                *reinterpret_cast<float4*>(tx + j * 4) = make_float4(gd0, gd1, gd2, 0.f);
                rowbuf[j] = i;
                if (valid) {
#if defined(PVS_ABL_F_SCATTER) || !PVS_SA_STORE
                    float* gd_at = io.gd + (size_t)PVS_ABL_SCR(e) * 4;
#else
                    float* gd_at = pvs_off(io.gd + (size_t)e0 * 4, 16u * j);       // (e0 is wave-uniform: a scalar base)
#endif
#ifndef PVS_ABL_F_NOGZ1      // timing-only: no per-edge outputs (what a column side fused into this kernel would not write)
                    pvs_store_nt(gd_at, make_float4(gd0, gd1, gd2, pvs_pack_rho_type(rho, ty)));
#endif
                }
            }
            // ---- g_z1 edge-major, then whole rows to HBM + the row-side sums from the same reads ----
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
                *reinterpret_cast<float4*>(T1 + (TSWZ ? pvs_tile_quad_off<1>(j, 2 * gq + hh) : j * Cfg::kTS + 8 * gq + 4 * hh)) =
                    make_float4(g_z1[0][4 * gq], g_z1[0][4 * gq + 1], g_z1[0][4 * gq + 2], g_z1[0][4 * gq + 3]);
            pvs_wave_lds_sync();
            PVS_TP(14);
#ifndef PVS_ABL_F_NOREDUCE
            reduce_rows_tile<1, false, true, TSWZ>(T1, tx, rowbuf, bmask, lane, acc, accx, cur_row, flush,
                                [&](int rl, int q, const float4& v) {
                                    if (e0 + rl < e_this_end) {  // streamed once: non-temporal
#if defined(PVS_ABL_F_SCATTER) || !PVS_SA_STORE
                                        float* at = io.gz1 + (size_t)PVS_ABL_SCR(e0 + rl) * H + 4 * q;
#else
                                        float* at = pvs_off(io.gz1 + (size_t)e0 * H, (unsigned)rl * (4u * H) + 16u * q);
#endif
#ifndef PVS_ABL_F_NOGZ1
                                        pvs_store_nt(at, v);
#endif
                                    }
                                });
#endif
            I = In;
            e0 = e_this_end;
            t_end = n_end;
            pvs_wave_lds_sync();
            PVS_TP(15);
#ifdef PVS_TILE_TRACE
            ++tile_no;
#endif
        }
        flush(cur_row);
    }

    // ---- block reduction into one slab, fixed order ----
    const PvsSlabLayout L = pvs_slab_layout(H);
    __syncthreads();
    float* slab = smem;
    for (int i = threadIdx.x; i < L.total; i += NT) slab[i] = 0.f;
    if constexpr (LAZY) {       // the accumulators back to true values: an exact power of two each (in two factors:
        // 1 / (s_G s_Act) = 2^(units - 280), units = the sum of two exponents in [17, 254], need not be a normal float)
        auto pw = [](int u, int n) { return u < 0 ? 1.f : __uint_as_float((unsigned)(n ? u - u / 2 - 13 : u / 2 - 13) << 23); };
        const float a2 = pw(u_w2.cur(), 0), b2 = pw(u_w2.cur(), 1), ac = pw(u_wc1.cur(), 0), bc = pw(u_wc1.cur(), 1);
        const float kb2 = u_b2.cur() < 0 ? 1.f : __uint_as_float((unsigned)(u_b2.cur() - 13) << 23);
        const float kbc = u_bc1.cur() < 0 ? 1.f : __uint_as_float((unsigned)(u_bc1.cur() - 13) << 23);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            gW2[r] = gW2[r] * a2 * b2;
            gWc1[r] = gWc1[r] * ac * bc;
            gB[r] *= j == 0 ? kbc : (j == 1 ? kb2 : 1.f);
        }
    }
    __syncthreads();
    // X-layout vectors: sum over the 32 edge lanes of each half
    auto lanes32 = [](float v) {
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) v += __shfl_xor(v, o, 64);
        return v;
    };
#pragma unroll
    for (int r = 0; r < 16; ++r) g_wc2x[r] = lanes32(g_wc2x[r]);
    float g_wax[EATT ? 16 : 1];
    if constexpr (EATT) {
#pragma unroll
        for (int r = 0; r < 16; ++r) g_wax[r] = lanes32(j >= 2 ? gB[r] : 0.f);      // columns 2..31 of gB
    }
    g_ba += __shfl_xor(g_ba, 32, 64);          // only hh == 0 lanes accumulated
    g_ba = lanes32(g_ba);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) g_gate += __shfl_xor(g_gate, o, 64);
    for (int turn = 0; turn < NW; ++turn) {
        if (wv == turn) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = xch(r, hh), k = j;
                slab[L.w2 + c * H + k] += gW2[r];
                slab[L.wc1 + c * H + k] += gWc1[r];
            }
            if (j == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int c = xch(r, hh);
                    slab[L.wc2 + c] += g_wc2x[r];
                    if constexpr (EATT) slab[L.wa + c] += g_wax[r];
                }
            }
            if (j <= 1) {      // bias gradients: column 0 of gB is g_bc1, column 1 is g_b2
#pragma unroll
                for (int r = 0; r < 16; ++r) slab[(j == 0 ? L.bc1 : L.b2) + xch(r, hh)] += gB[r];
            }
            if (lane == 0) { slab[L.ba] += g_ba; slab[L.gate] += g_gate; }
        }
        __syncthreads();
    }
    float* dst = io.slabs + (size_t)blockIdx.x * L.total;
    for (int i = threadIdx.x; i < L.total; i += NT) dst[i] = slab[i];
}

}  // namespace

#ifdef PVS_PAIR_PROBE
#include "probe_pair_bwd.h"
#endif

#ifdef PVS_TILE_TRACE
extern "C" int pvs_debug_tile_trace(unsigned long long* dst, size_t count) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(pvs_trace_buf), count * sizeof(unsigned long long));
}
#endif

// Same contract as pvs_launch_edge_bwd_mfma (edge_mfma.hip). H = 32 only.
int pvs_launch_edge_bwd_f16(hipStream_t s, int H, const PvsGraph& g, const PvsEdgeW& w, uint32_t flags, int att_act,
                            const PvsEdgeBwdIO& io, int e_lo, int e_hi, int* n_slabs) {
    PVS_REQUIRE(w.n_attr <= 3, "MFMA edge backward supports up to 3 edge classes (got %d)", w.n_attr);
    PVS_REQUIRE(H == 32, "f16x2 edge backward is built for H = 32 (got %d)", H);
    *n_slabs = 0;
    if (e_hi <= e_lo) return 0;
    using Cfg = F16Cfg;
    constexpr int nw = Cfg::kWavesPerBlock;
    int blocks, n_chunks;
    {
        const int E = e_hi - e_lo;
        const long long per = pvs_edges_per_wave();
        long long b = ((long long)E + (long long)nw * per - 1) / ((long long)nw * per);   // fill the chip first
        if (b < 1) b = 1;
        if (b > 256) b = 256;                      // one workgroup per CU (LDS)
        const long long waves = b * nw;
        const long long ce = pvs_chunk_edges(8192);     // (one chunk per wave at cfg2: edge_mfma_common.h)
        long long per_wave = ((long long)E + waves * ce - 1) / (waves * ce);
        if (per_wave < 1) per_wave = 1;
        blocks = (int)b;
        n_chunks = (int)(waves * per_wave);
    }
    *n_slabs = blocks;
    PvsProfScope prof(s, PVS_PROF_EDGE_BWD);
    const PvsSlabLayout L = pvs_slab_layout(kH);
    size_t lds = (size_t)Cfg::kSharedBytes + (size_t)nw * Cfg::kWaveBytes;
    if (lds < (size_t)L.total * 4) lds = (size_t)L.total * 4;
    const bool eres = (flags & PVS_EDGE_RESIDUAL) && io.m_prev != nullptr;
    const bool eatt = flags & PVS_EDGE_ATTENTION;
#ifdef PVS_PAIR_PROBE       // timing-only: the channel-split wave-pair kernel in place of <0, false> (probe_pair_bwd.h)
    if (!eres && !eatt) {
        const int pblocks = 256 * PVS_PAIR_PROBE_OCC, ppairs = 2 * pblocks;
        long long per = ((long long)(e_hi - e_lo) + (long long)ppairs * 2048 - 1) / ((long long)ppairs * 2048);
        if (per < 1) per = 1;
        *n_slabs = pblocks;
        if (set_lds(pairprobe::k_edge_bwd_pair_probe, (size_t)pairprobe::kLds)) return -2;
        pairprobe::k_edge_bwd_pair_probe<<<pblocks, 256, pairprobe::kLds, s>>>(g, w, flags, io, (int)(ppairs * per), e_lo, e_hi);
        PVS_CHECK_LAUNCH();
        return 0;
    }
#endif
#define PVS_BWD_F16_LAUNCH(ER, EA)                                                                          \
    do {                                                                                                   \
        if (set_lds(k_edge_bwd_f16<ER, EA>, lds)) return -2;                                               \
        k_edge_bwd_f16<ER, EA><<<blocks, Cfg::kThreadsPerBlock, lds, s>>>(g, w, flags, att_act, io, n_chunks, \
                                                                          e_lo, e_hi);                   \
    } while (0)
    // (as the reference orders them: rezero wins over gated, egnn_satorras.py:194-202)
    const bool rezero = flags & PVS_REZERO, gated = !rezero && (flags & PVS_GATED_RESIDUAL);
    if (eres && rezero && eatt) PVS_BWD_F16_LAUNCH(2, true);
    else if (eres && rezero) PVS_BWD_F16_LAUNCH(2, false);
    else if (eres && gated && eatt) PVS_BWD_F16_LAUNCH(3, true);
    // Gated residual without attention runs the COMPILE-TIME kind 3 since round 6. Until then it ran kind 4 (the kind read
    // from the flags at run time: 11 spilled VGPRs against 13) - and kind 4 with BOTH the lazy scales and
    // the pair arithmetic compiled in gave g_z2-derived outputs (g_h, g_x, the edge_mlp gradients) that were 1e-3 ... 1e-1 off
    // and changed from run to run, while g_m_prev and the coordinate branch stayed right; either feature off, or kind 3,
    // and it is bit-reproducible and within 1e-6 of the exact family (profiles/r06_gated_residual_backward_defect.txt:
    // found by fuzz seed 116; no golden case ran that instantiation on more than a few tiles). The cause inside the
    // instantiation was not found (no asm-related hazard in its ISA, spills outside the tile loop); the kind-4 code path is
    // no longer instantiated. tools/backward_instantiations_probe.py and the GPU test of the same name run all 24
    // instantiations of the H = 32 / 64 backward twice on a multi-tile graph.
#ifdef PVS_GATED_KIND4          // (A/B only: the defective run-time kind, for whoever looks for the cause; never shipped)
    else if (eres && gated) PVS_BWD_F16_LAUNCH(4, false);
#else
    else if (eres && gated) PVS_BWD_F16_LAUNCH(3, false);
#endif
    else if (eres && eatt) PVS_BWD_F16_LAUNCH(1, true);
    else if (eres) PVS_BWD_F16_LAUNCH(1, false);
    else if (eatt) PVS_BWD_F16_LAUNCH(0, true);
    else PVS_BWD_F16_LAUNCH(0, false);
#undef PVS_BWD_F16_LAUNCH
    PVS_CHECK_LAUNCH();
    return 0;
}
