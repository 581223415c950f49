// Edge backward for H = 64: one wave per 16-EDGE tile, every product on the bf16 matrix pipe ("bf16x3",
// edge_mfma_common.h), no inter-wave exchange and no workgroup barrier inside the edge loop.
//
// Why not the H = 32 kernel's shape (edge_bwd_bf16.hip: one wave per 32-edge tile, 32x32x16 MFMAs): at
// H = 64 a lane of that layout carries 32 channels of every live tensor and the kernel needs ~650
// registers; the round-1 answer - a TEAM of two waves that own 32 channels each (k_edge_bwd_team_parts,
// edge_mfma.hip) - hands every activation to the partner through LDS behind workgroup barriers and, with
// 36 KB of tiles per team, fits one wave per SIMD: 22k SIMD cycles per tile against ~10k of issue work
// (profiles/r02_ab_h64_team_ablations.txt).
// Here the chain products use v_mfma_f32_16x16x32_bf16: A = a 16-channel row block of the weights, B = the
// activations of 16 edges, so a lane (edge n = lane & 15, group g = lane >> 4) carries 16 channels of
// every tensor - the register footprint of the H = 32 kernel - for all 64 channels of its edges:
//   "Y layout": register r of lane (n, g) holds channel 16 (r >> 2) + 4 g + (r & 3) of edge n.
// It is the accumulator layout of four 16x16x32 products (row block b = r >> 2) and, read as registers
// 8s .. 8s+7, the B operand of k-step s of the next product when the weights' A operand uses the same
// channel order (frag16). The two weight gradients are products over the EDGE index (K = 16 = one k-step
// of v_mfma_f32_32x32x16_bf16) whose A and B operands are transposing reads (ds_read_b64_tr_b16) of
// row-major [edge][channel] part images: the activation images are written anyway, the gradient tensors
// get a third image slot. 128 accumulator registers hold the two 64x64 weight gradients, which is why this
// kernel runs one wave per SIMD (512 registers); what it gives up in latency hiding it wins back by never
// waiting for another wave: the next tile's node rows are gathered while the current tile is computed.
//
// Reference semantics: autograd of EGNNLayer.edge_model / coord_model / node_model's aggregation,
// /root/reference/point_vs/models/geometric/egnn_satorras.py:123-206 (SURVEY.md §8a "Backward spec").
#include "edge_mfma_common.h"
// Pair arithmetic (common.h pvs_f2) in the elementwise blocks that have no product to run beside: one wave per SIMD
// issues one instruction per ~5 cycles whatever it is, so two results per instruction is two issue slots for one
// (cfg3: -1...1.7 % per launch). NOT in the block woven beside the gW2 product (wgrad16_beside): paired there, the
// pinned units no longer fill the MFMAs' shadows evenly (23 more s_nop, +0.5 % instead of -1 %).
#ifndef PVS_PAIR_H64
#define PVS_PAIR_H64 1
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kH = 64;
constexpr int kT16 = 16;                       // edges per tile
constexpr int kPart16 = kT16 * kH;             // bf16 elements of one part image of a [16][64] tensor
constexpr int kImg16 = 3 * kPart16;            // hi, mid, lo
constexpr int kTS16 = kH + 4;                  // row stride (floats) of the fp32 g_z1 tile
constexpr int kWaveBytes64 = 3 * kImg16 * 2;   // a1, m (then the g_z1 tile, tx, rowbuf), g (g_zc then g_z2)
constexpr int kSharedBytes64 = 2 * 3 * kH * kH * 2 + (5 + PVS_MAX_EDGE_ATTR) * kH * 4;
static_assert(kT16 * kTS16 * 4 + kT16 * 16 + kT16 * 4 <= kImg16 * 2, "g_z1 tile + tx + rowbuf must fit the m image");

__device__ __forceinline__ int ych(int r, int gr) { return 16 * (r >> 2) + 4 * gr + (r & 3); }

// ---- LDS images of [rows][64] bf16 tensors: the 8-byte chunks of a row permuted by an XOR that depends on the
// row. ds_read_b64 / ds_read_b64_tr_b16 are served per 32-lane half over 64 four-byte banks, so the 32 lanes of a
// half must hit 32 different 8-byte units mod 256 B.
// The swizzle of the H = 32 kernels (f = row / 2) leaves this kernel's reads 2-way (row reads) and 4-way
// (transposed reads) conflicting - SQ_LDS_BANK_CONFLICT was half of all LDS cycles - so each image gets the f
// its own access patterns need:
//  * weights (frag16): per 16-row block TWO PLANES of [16 rows][32 bf16] - plane (c >> 4) & 1 holds the channels
//    whose Y-layout register index has bit 2 set / clear - so that the two 8-byte halves of a lane's k-step
//    fragment (channels 32 s + 4 g + . and 32 s + 16 + 4 g + .) sit a CONSTANT 1 KB apart, closer than any other
//    pair of the product's loads: the compiler fuses loads by offset distance, and this pair becomes one
//    ds_read2st64_b64 that lands in the operand's four consecutive registers. (With both halves in one
//    128-byte row under a row-dependent XOR it paired loads of DIFFERENT fragments and rebuilt every operand
//    with four v_mov: 130 per tile.) Inside a plane a row is 64 B = 8 chunks (s, g); row reads vary row bits
//    0-3 and g0, transposed reads vary row bits 0-2 and chunk bits 0-1 (p):  f = (r2, r3) -> chunk bits (2, 1)
//    makes both hit 32 different 8-byte units mod 256 B (unit = 8 (row & 3) + (chunk ^ f)).
//  * activations / gradients ([16][64], 128-byte rows, unit = 16 (row & 1) + (chunk ^ f)): writes vary the row (16
//    lanes of one group: f must be a bijection of the row's 4 bits), transposed reads vary row bits 0-1 (q) and
//    chunk bits 0-2 (p, g0):  f = (r0, r3, r2, r1) -> bits (0, 1, 2, 3)
constexpr int kWPlane = 16 * 32;
__device__ __forceinline__ int w_off(int r, int c) {
    // (round 5: + row bit 1 -> chunk bit 0. The paired row read ds_read2st64_b64 is NOT served like ds_read_b64 (32 lanes
    // over 64 banks) but as two accesses of 4 x 16 contiguous lanes over 32 banks (MI355X_MICROARCH.md, LDS): 16 lanes =
    // the 16 rows of a block at one chunk, two 64-byte rows per 32 banks, so f must be a bijection of (r1, r2, r3) -
    // with (r2, r3) alone rows r and r + 2 collided, 2-way on all 48 paired reads of a tile = the 27 % of LDS time
    // SQ_LDS_BANK_CONFLICT showed; tools/lds_conflicts.py. The transposed reads vary r0-r2 and chunk bits 0-1 and stay
    // conflict-free: r1 is constant inside one of their 64-byte slots.)
    const int f = (((r >> 2) & 1) << 2) | (((r >> 3) & 1) << 1) | ((r >> 1) & 1);
    const int chunk = ((c >> 5) << 2) | ((c >> 2) & 3);
    return (r >> 4) * 2 * kWPlane + ((c >> 4) & 1) * kWPlane + (r & 15) * 32 + 4 * (chunk ^ f) + (c & 3);
}
__device__ __forceinline__ int a_off(int n, int c) {
    const int f = (n & 1) | (((n >> 3) & 1) << 1) | (((n >> 2) & 1) << 2) | (((n >> 1) & 1) << 3);
    return n * kH + 4 * ((c >> 2) ^ f) + (c & 3);
}

// the three bf16 parts of W [64][64] (row-major) as three images
__device__ __forceinline__ void stage_weights_y(unsigned short* img, const float* __restrict__ W) {
    unsigned* hi = reinterpret_cast<unsigned*>(img);
    unsigned* mid = reinterpret_cast<unsigned*>(img + kH * kH);
    unsigned* lo = reinterpret_cast<unsigned*>(img + 2 * kH * kH);
    for (int i = threadIdx.x; i < kH * kH / 2; i += blockDim.x) {
        const int r = (2 * i) / kH, c = (2 * i) % kH;
        const float x0 = W[r * kH + c], x1 = W[r * kH + c + 1];
        const float r0 = x0 - __uint_as_float(__float_as_uint(x0) & 0xffff0000u);
        const float r1 = x1 - __uint_as_float(__float_as_uint(x1) & 0xffff0000u);
        const float t0 = r0 - __uint_as_float(__float_as_uint(r0) & 0xffff0000u);
        const float t1 = r1 - __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
        const int o = w_off(r, c) >> 1;
        hi[o] = pvs_pack_hi16(x0, x1);
        mid[o] = pvs_pack_hi16(r0, r1);
        lo[o] = pvs_pack_hi16(t0, t1);
    }
}

// operand of v_mfma_f32_32x32x16_bf16 over the EDGE index from one part image [16 edges][64 channels]:
// lane (c = lane & 31, hh) receives channel 32 blk + c of edges 4 hh + (0..3) and 8 + 4 hh + (0..3)
__device__ __forceinline__ bf16x8 edge_fragment(const unsigned short* __restrict__ part, int lane, int blk) {
    const int hh = lane >> 5, li = lane & 15, q = li >> 2, p = li & 3;
    const int r0 = 4 * hh, col = 32 * blk + 16 * ((lane >> 4) & 1) + 4 * p;
    typedef pvs_v4s __attribute__((address_space(3))) * lds_v4s;
    const pvs_v4s ta = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(part + a_off(r0 + q, col)));
    const pvs_v4s tb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(part + a_off(r0 + 8 + q, col)));
    const uint2 a = __builtin_bit_cast(uint2, ta), b = __builtin_bit_cast(uint2, tb);
    return __builtin_bit_cast(bf16x8, make_uint4(a.x, a.y, b.x, b.y));
}

// a row of a row-major [rows][64] array <-> Y layout: 4 floats at 16 q + 4 g, q = 0..3
__device__ __forceinline__ void load_y(const float* __restrict__ base, int gr, float (&out)[16]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(base + 16 * q + 4 * gr);
        out[4 * q] = v.x; out[4 * q + 1] = v.y; out[4 * q + 2] = v.z; out[4 * q + 3] = v.w;
    }
}

__device__ __forceinline__ void store_y(float* __restrict__ base, int gr, const float (&v)[16]) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(base + 16 * q + 4 * gr) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}

// A operand of v_mfma_f32_16x16x32_bf16 (lane: row m = lane & 15, k = 8 (lane >> 4) + j) for output block b
// and k-step s from one part image of W [64][64] (w_off swizzle): k index (g, j) is channel
// 32 s + 16 (j >> 2) + 4 g + (j & 3), the Y layout's order of registers 8s .. 8s+7.
// TRANSPOSE = false: A[m][k] = W[16 b + m][channel]  (W v);  true: A[m][k] = W[channel][16 b + m]  (W^T v).
template <bool TRANSPOSE>
__device__ __forceinline__ bf16x8 frag16(const unsigned short* __restrict__ part, int lane, int b, int s) {
    const int gr = lane >> 4;
    uint2 a, c;
    if constexpr (!TRANSPOSE) {
        const int r = 16 * b + (lane & 15), c0 = 32 * s + 4 * gr;
        const unsigned short* q = part + w_off(r, c0);
        a = *reinterpret_cast<const uint2*>(q);
        c = *reinterpret_cast<const uint2*>(q + kWPlane);
    } else {
        // 16-lane group: lane 4q+p supplies row q, columns 4p..4p+3 of a 4x16 block and receives column
        // (lane & 15) of its 4 rows
        const int li = lane & 15, q = li >> 2, p = li & 3;
        const int r0 = 32 * s + 4 * gr, col = 16 * b + 4 * p;
        typedef pvs_v4s __attribute__((address_space(3))) * lds_v4s;
        const pvs_v4s ta = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(part + w_off(r0 + q, col)));
        const pvs_v4s tb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(part + w_off(r0 + 16 + q, col)));
        a = __builtin_bit_cast(uint2, ta);
        c = __builtin_bit_cast(uint2, tb);
    }
    return __builtin_bit_cast(bf16x8, make_uint4(a.x, a.y, c.x, c.y));
}

// acc[b] (channels 16 b + 4 g + .) += W v  (TRANSPOSE: W^T v), v given as its bf16 parts in Y layout.
// 48 MFMAs in six groups - (lo, s), (mid, s), (hi, s) for the two k-steps s, the small terms first - each on
// the four row blocks' fragments of one part image. One wave per SIMD has nobody to hide an LDS round trip
// behind, so the fragments are fetched TWO groups ahead of their use (three buffers of 16 registers): with
// the loads placed next to their uses the compiler emitted load - wait(0) - 4 MFMAs twelve times per product,
// ~5k exposed cycles per tile.
template <bool TRANSPOSE>
__device__ __forceinline__ void chain16(const unsigned short* __restrict__ img, int lane, const Bf16Parts& v,
                                        f32x4 (&acc)[4]) {
    const unsigned short* lo = img + 2 * kH * kH;
    const unsigned short* mid = img + kH * kH;
    const unsigned short* hi = img;
    bf16x8 f0[4], f1[4], f2[4];
    auto ld = [&](const unsigned short* part, int s, bf16x8 (&f)[4]) {
#pragma unroll
        for (int b = 0; b < 4; ++b) f[b] = frag16<TRANSPOSE>(part, lane, b, s);
    };
    auto mm = [&](const bf16x8 (&f)[4], const bf16x8& x) {
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[b], x, acc[b], 0, 0, 0);
    };
    ld(lo, 0, f0);
    ld(lo, 1, f1);
    ld(mid, 0, f2);
    mm(f0, v.hi[0]);                                   // lo, s = 0
    ld(mid, 1, f0);
    mm(f1, v.hi[1]);                                   // lo, s = 1
    ld(hi, 0, f1);
    mm(f2, v.mid[0]); mm(f2, v.hi[0]);                 // mid, s = 0
    ld(hi, 1, f2);
    mm(f0, v.mid[1]); mm(f0, v.hi[1]);                 // mid, s = 1
    mm(f1, v.lo[0]); mm(f1, v.mid[0]); mm(f1, v.hi[0]);   // hi, s = 0
    mm(f2, v.lo[1]); mm(f2, v.mid[1]); mm(f2, v.hi[1]);   // hi, s = 1
}

// chain16 with the operand's split INSIDE the product. One wave per SIMD: the only vector work an MFMA can run
// beside is this wave's own, and a 16x16x32 MFMA leaves room for two vector instructions (8 of its 16 cycles hold
// the issue port). The split is 88 instructions per tensor - 8 v_perm for the hi parts, (v_and, v_sub) per value
// for each residual, 8 v_perm per further part - and only the hi parts are needed by the first products, so the
// order below is: hi parts of k-step 0, then every MFMA followed by ONE unit (two instructions) of the remaining
// split, pinned in program order by sched_barrier(0) after every MFMA (left to itself the scheduler finishes the
// split first and then issues the 48 MFMAs back to back with the vector ALU idle). Product order: every group of
// four MFMAs needs only parts that are ready - (lo,0)hi (lo,1)hi (mid,0)hi (mid,0)mid (mid,1)hi (hi,0)hi (mid,1)mid
// (hi,0)mid (hi,0)lo (hi,1)hi (hi,1)mid (hi,1)lo; fragment loads stay two groups ahead as in chain16.
// GLOBAL (timing-only probe, -DPVS_ABL_H64_WC1_GLOBAL, profiles/r06_ab_h64_lds_residency.txt): the weight's operand
// fragments as PRE-ARRANGED 16-byte words in global memory - fragment (part, s, b) of lane l at word
// ((part * 2 + s) * 4 + b) * 64 + l, one fully coalesced 1 KB load per fragment, what csrc/edge_bwd_wide.hip does for
// coord_mlp.0 - instead of reads of the LDS image: what freeing that weight's 24.5 KB of LDS would cost.
template <bool TRANSPOSE, bool GLOBAL = false>
__device__ __forceinline__ void chain16s(const unsigned short* __restrict__ img, int lane, const float (&x)[16],
                                         Bf16Parts& pb, f32x4 (&acc)[4]) {
    const unsigned short* lo = img + 2 * kH * kH;
    const unsigned short* mid = img + kH * kH;
    const unsigned short* hi = img;
    bf16x8 f0[4], f1[4], f2[4];
    unsigned ph[8], pm[8], pl[8];
    float r[16], t[16];
    auto ld = [&](const unsigned short* part, int s, bf16x8 (&f)[4]) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if constexpr (GLOBAL) {
                const int pi = (int)((part - img) / (kH * kH));
                // (a wave-uniform base + the lane's 32-bit byte offset: one address register for all 24 fragments)
                f[b] = __builtin_bit_cast(bf16x8, *pvs_off(reinterpret_cast<const uint4*>(img) + (((pi * 2 + (TRANSPOSE ? 1 : 0)) * 2 + s) * 4 + b) * 64, 16u * (unsigned)lane));
            } else {
                f[b] = frag16<TRANSPOSE>(part, lane, b, s);
            }
        }
    };
    auto res = [](float v) { return v - __uint_as_float(__float_as_uint(v) & 0xffff0000u); };
    auto part = [](const unsigned (&p)[8], int s) {
        return __builtin_bit_cast(bf16x8, make_uint4(p[4 * s], p[4 * s + 1], p[4 * s + 2], p[4 * s + 3]));
    };
    // one unit = two vector instructions
    auto HI = [&](int i) { ph[i] = pvs_pack_hi16(x[2 * i], x[2 * i + 1]); ph[i + 1] = pvs_pack_hi16(x[2 * i + 2], x[2 * i + 3]); };
    auto MID = [&](int i) { pm[i] = pvs_pack_hi16(r[2 * i], r[2 * i + 1]); pm[i + 1] = pvs_pack_hi16(r[2 * i + 2], r[2 * i + 3]); };
    auto LO = [&](int i) { pl[i] = pvs_pack_hi16(t[2 * i], t[2 * i + 1]); pl[i + 1] = pvs_pack_hi16(t[2 * i + 2], t[2 * i + 3]); };
    auto R = [&](int k) { r[k] = res(x[k]); };
    auto T = [&](int k) { t[k] = res(r[k]); };
#define PVS_MM(f, b, v) do { acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[b], v, acc[b], 0, 0, 0); } while (0)
#define PVS_PIN() __builtin_amdgcn_sched_barrier(0)
    HI(0); HI(2);
    ld(lo, 0, f0);
    ld(lo, 1, f1);
    ld(mid, 0, f2);
    PVS_PIN();
    {   // (lo, 0) hi0
        const bf16x8 h0 = part(ph, 0);
        PVS_MM(f0, 0, h0); HI(4); PVS_PIN();
        PVS_MM(f0, 1, h0); HI(6); PVS_PIN();
        PVS_MM(f0, 2, h0); R(0); PVS_PIN();
        PVS_MM(f0, 3, h0); R(1); PVS_PIN();
    }
    ld(mid, 1, f0);
    const bf16x8 h0 = part(ph, 0), h1 = part(ph, 1);
    PVS_MM(f1, 0, h1); R(2); PVS_PIN();          // (lo, 1) hi1
    PVS_MM(f1, 1, h1); R(3); PVS_PIN();
    PVS_MM(f1, 2, h1); R(4); PVS_PIN();
    PVS_MM(f1, 3, h1); R(5); PVS_PIN();
    ld(hi, 0, f1);
    PVS_MM(f2, 0, h0); R(6); PVS_PIN();          // (mid, 0) hi0
    PVS_MM(f2, 1, h0); R(7); PVS_PIN();
    PVS_MM(f2, 2, h0); MID(0); PVS_PIN();
    PVS_MM(f2, 3, h0); MID(2); PVS_PIN();
    const bf16x8 m0 = part(pm, 0);
    PVS_MM(f2, 0, m0); R(8); PVS_PIN();          // (mid, 0) mid0
    PVS_MM(f2, 1, m0); R(9); PVS_PIN();
    PVS_MM(f2, 2, m0); R(10); PVS_PIN();
    PVS_MM(f2, 3, m0); R(11); PVS_PIN();
    ld(hi, 1, f2);
    PVS_MM(f0, 0, h1); R(12); PVS_PIN();         // (mid, 1) hi1
    PVS_MM(f0, 1, h1); R(13); PVS_PIN();
    PVS_MM(f0, 2, h1); R(14); PVS_PIN();
    PVS_MM(f0, 3, h1); R(15); PVS_PIN();
    PVS_MM(f1, 0, h0); MID(4); PVS_PIN();        // (hi, 0) hi0
    PVS_MM(f1, 1, h0); MID(6); PVS_PIN();
    PVS_MM(f1, 2, h0); T(0); PVS_PIN();
    PVS_MM(f1, 3, h0); T(1); PVS_PIN();
    const bf16x8 m1 = part(pm, 1);
    PVS_MM(f0, 0, m1); T(2); PVS_PIN();          // (mid, 1) mid1
    PVS_MM(f0, 1, m1); T(3); PVS_PIN();
    PVS_MM(f0, 2, m1); T(4); PVS_PIN();
    PVS_MM(f0, 3, m1); T(5); PVS_PIN();
    PVS_MM(f1, 0, m0); T(6); PVS_PIN();          // (hi, 0) mid0
    PVS_MM(f1, 1, m0); T(7); PVS_PIN();
    PVS_MM(f1, 2, m0); LO(0); PVS_PIN();
    PVS_MM(f1, 3, m0); LO(2); PVS_PIN();
    const bf16x8 l0 = part(pl, 0);
    PVS_MM(f1, 0, l0); T(8); PVS_PIN();          // (hi, 0) lo0
    PVS_MM(f1, 1, l0); T(9); PVS_PIN();
    PVS_MM(f1, 2, l0); T(10); PVS_PIN();
    PVS_MM(f1, 3, l0); T(11); PVS_PIN();
    PVS_MM(f2, 0, h1); T(12); PVS_PIN();         // (hi, 1) hi1
    PVS_MM(f2, 1, h1); T(13); PVS_PIN();
    PVS_MM(f2, 2, h1); T(14); PVS_PIN();
    PVS_MM(f2, 3, h1); T(15); PVS_PIN();
    PVS_MM(f2, 0, m1); LO(4); PVS_PIN();         // (hi, 1) mid1
    PVS_MM(f2, 1, m1); LO(6); PVS_PIN();
    PVS_MM(f2, 2, m1);
    PVS_MM(f2, 3, m1);
    const bf16x8 l1 = part(pl, 1);
    PVS_MM(f2, 0, l1); PVS_MM(f2, 1, l1); PVS_MM(f2, 2, l1); PVS_MM(f2, 3, l1);   // (hi, 1) lo1
#undef PVS_MM
#undef PVS_PIN
    pb.hi[0] = h0; pb.hi[1] = h1; pb.mid[0] = m0; pb.mid[1] = m1; pb.lo[0] = l0; pb.lo[1] = l1;
}

// the three bf16 parts of a Y-layout tensor -> row-major [edge][channel] images (a_off swizzle)
__device__ __forceinline__ void write_image16(unsigned short* __restrict__ img, int n, int gr, const Bf16Parts& p) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        // registers 8s..8s+3: channels 32 s + 4 g + (0..3); registers 8s+4..8s+7: channels 32 s + 16 + 4 g + (0..3)
        const int o0 = a_off(n, 32 * s + 4 * gr), o1 = a_off(n, 32 * s + 16 + 4 * gr);
        const uint4 h = __builtin_bit_cast(uint4, p.hi[s]);
        const uint4 m = __builtin_bit_cast(uint4, p.mid[s]);
        const uint4 l = __builtin_bit_cast(uint4, p.lo[s]);
        *reinterpret_cast<uint2*>(img + o0) = make_uint2(h.x, h.y);
        *reinterpret_cast<uint2*>(img + o1) = make_uint2(h.z, h.w);
        *reinterpret_cast<uint2*>(img + kPart16 + o0) = make_uint2(m.x, m.y);
        *reinterpret_cast<uint2*>(img + kPart16 + o1) = make_uint2(m.z, m.w);
        *reinterpret_cast<uint2*>(img + 2 * kPart16 + o0) = make_uint2(l.x, l.y);
        *reinterpret_cast<uint2*>(img + 2 * kPart16 + o1) = make_uint2(l.z, l.w);
    }
}

// gW[bo][bi] (D layout of the 32x32 instruction: [c = 32 bo + ch(r, hh)][k = 32 bi + (lane & 31)]) +=
// sum over the tile's 16 edges of G[e][c] * Act[e][k]: both operands are transposing reads of the images
// (edge = k index of ONE 32x32x16 k-step; the same edge order on both sides).
__device__ __forceinline__ void wgrad16(const unsigned short* __restrict__ gimg, const unsigned short* __restrict__ aimg,
                                        int lane, f32x16 (&gW)[2][2]) {
#pragma unroll
    for (int bo = 0; bo < 2; ++bo) {
        const bf16x8 gh = edge_fragment(gimg, lane, bo);
        const bf16x8 gm = edge_fragment(gimg + kPart16, lane, bo);
        const bf16x8 gl = edge_fragment(gimg + 2 * kPart16, lane, bo);
#pragma unroll
        for (int bi = 0; bi < 2; ++bi) {
            const bf16x8 ah = edge_fragment(aimg, lane, bi);
            const bf16x8 am = edge_fragment(aimg + kPart16, lane, bi);
            const bf16x8 al = edge_fragment(aimg + 2 * kPart16, lane, bi);
            gW[bo][bi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gl, ah, gW[bo][bi], 0, 0, 0);
            gW[bo][bi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gh, al, gW[bo][bi], 0, 0, 0);
            gW[bo][bi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gm, am, gW[bo][bi], 0, 0, 0);
            gW[bo][bi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gm, ah, gW[bo][bi], 0, 0, 0);
            gW[bo][bi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gh, am, gW[bo][bi], 0, 0, 0);
            gW[bo][bi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gh, ah, gW[bo][bi], 0, 0, 0);
        }
    }
}

// gW2 += g_z2 (x) a1 of the PREVIOUS tile (images gimg, aimg) beside THIS tile's z1 -> a1 = SiLU(z1), SiLU'(z1):
// 24 MFMAs of 32 cycles, each followed by its share of the 16 values' vector work (two values per three MFMAs),
// pinned like chain16s. Fragment sets (three parts, 12 registers) are read one block pair ahead; the block order
// (0,0) (0,1) (1,1) (1,0) reuses the activation fragments of block 1.
template <class S1, class S2>
__device__ __forceinline__ void wgrad16_beside(const unsigned short* __restrict__ gimg, const unsigned short* __restrict__ aimg,
                                               int lane, f32x16 (&gW)[2][2], S1&& s1, S2&& s2) {
    bf16x8 g0[3], g1[3], a0[3], a1[3];
    auto ld = [&](const unsigned short* img, int blk, bf16x8 (&f)[3]) {
#pragma unroll
        for (int p = 0; p < 3; ++p) f[p] = edge_fragment(img + p * kPart16, lane, blk);
    };
    // MFMA K (0..23) and its share of the vector work: values 2 (K / 3) and 2 (K / 3) + 1 over three MFMAs.
    // (hi, mid, lo) = parts 0, 1, 2; small terms first.
#define PVS_WG(acc, gp, ap, K)                                                          \
    do {                                                                                \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gp, ap, acc, 0, 0, 0);            \
        asm volatile("" : "+a"(acc));                                                   \
        if constexpr ((K) % 3 == 0) { s1(2 * ((K) / 3)); }                              \
        else if constexpr ((K) % 3 == 1) { s2(2 * ((K) / 3)); s1(2 * ((K) / 3) + 1); }  \
        else { s2(2 * ((K) / 3) + 1); }                                                 \
        __builtin_amdgcn_sched_barrier(0);                                              \
    } while (0)
#define PVS_SIX(acc, g, a, K0)                                                          \
    PVS_WG(acc, g[2], a[0], K0); PVS_WG(acc, g[0], a[2], K0 + 1); PVS_WG(acc, g[1], a[1], K0 + 2); \
    PVS_WG(acc, g[1], a[0], K0 + 3); PVS_WG(acc, g[0], a[1], K0 + 4); PVS_WG(acc, g[0], a[0], K0 + 5)
    ld(gimg, 0, g0);
    ld(aimg, 0, a0);
    ld(aimg, 1, a1);
    __builtin_amdgcn_sched_barrier(0);
    PVS_SIX(gW[0][0], g0, a0, 0);
    ld(gimg, 1, g1);
    __builtin_amdgcn_sched_barrier(0);
    PVS_SIX(gW[0][1], g0, a1, 6);
    ld(aimg, 0, a0);
    __builtin_amdgcn_sched_barrier(0);
    PVS_SIX(gW[1][1], g1, a1, 12);
    PVS_SIX(gW[1][0], g1, a0, 18);
#undef PVS_SIX
#undef PVS_WG
}

// per-lane partial dot over the lane's 16 channels, summed over the 4 lane groups of the edge
// (lane-permute instructions, not ds_bpermute: edge_mfma_common.h, pvs_xor32_sum)
__device__ __forceinline__ float sum_groups(float s) { return pvs_xor32_sum(pvs_xor16_sum(s)); }

// node rows and per-row / per-edge scalars of one tile, fetched one tile ahead. EVERYTHING the first half of a
// tile reads from HBM is in here: a load issued at the top of the tile would sit in front of the wait for this
// struct's registers (the loop-carried wait is vmcnt(0): one in-order counter), exposing its whole latency.
struct Gather16 {
    float P[16], Q[16];
    float xi0, xi1, xi2, xj0, xj1, xj2;
    float gT0, gT1, gT2;      // gxagg row (coordinate update)
    float aval, softd;        // attention value of the edge, softmax row term
};

template <bool EATT>
__device__ __forceinline__ void gather16(const PvsEdgeBwdIO& io, const TileIdx& t, int gr, bool upd, bool softmax,
                                         Gather16& G) {
    load_y(io.PQ + (size_t)t.i * 2 * kH, gr, G.P);
    load_y(io.PQ + (size_t)t.jn * 2 * kH + kH, gr, G.Q);
    const float* x = io.x;
    G.xi0 = x[3 * t.i]; G.xi1 = x[3 * t.i + 1]; G.xi2 = x[3 * t.i + 2];
    G.xj0 = x[3 * t.jn]; G.xj1 = x[3 * t.jn + 1]; G.xj2 = x[3 * t.jn + 2];
    // UNCONDITIONAL loads from a pointer chosen by the (uniform) condition - the consumer selects: a load under
    // a branch reaches its consumer through a phi, and the register copies that implement the phi wait for the
    // load where it was issued (s_waitcnt vmcnt(0) right behind the prefetch: the whole latency, every tile)
    const float* gx = upd ? io.gxagg : io.x;          // (x is [N,3] like gxagg)
    G.gT0 = gx[3 * t.i]; G.gT1 = gx[3 * t.i + 1]; G.gT2 = gx[3 * t.i + 2];
    G.aval = 1.f; G.softd = 0.f;
    if constexpr (EATT) {
        G.aval = io.att[t.ee];
        G.softd = *(softmax ? io.softD + t.i : io.att + t.ee);
    }
}

// Row (segment) reduction of the 16-edge g_z1 tile: reduce_rows_tile (edge_mfma_common.h) with 16 rows -
// lane = (row slot rsub = lane / 16, 16-byte quad = lane % 16), four rows per lane.
template <class Flush, class RowStore>
__device__ __forceinline__ void reduce_rows16(const float* __restrict__ T, const float* __restrict__ tx,
                                              const int* __restrict__ rowbuf, unsigned bmask, int lane, float4& acc,
                                              float4& accx, int& cur_row, Flush&& flush, RowStore&& store_row) {
    const int quad = lane & 15, rsub = lane >> 4;
    float4 v[4], dx[4];
    int seg[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int rl = 4 * k + rsub;
        v[k] = *reinterpret_cast<const float4*>(T + rl * kTS16 + 4 * quad);
        dx[k] = *reinterpret_cast<const float4*>(tx + rl * 4);
        store_row(rl, quad, v[k]);
        seg[k] = __popc(bmask & ((2u << rl) - 1u));
    }
    auto add4 = [](float4& a, const float4& b) {
#if PVS_PAIR_H64
        const pvs_f2 lo = pvs_f2{a.x, a.y} + pvs_f2{b.x, b.y}, hi = pvs_f2{a.z, a.w} + pvs_f2{b.z, b.w};
        a.x = lo.x; a.y = lo.y; a.z = hi.x; a.w = hi.y;
#else
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
#endif
    };
    if (bmask == 0u) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { add4(acc, v[k]); add4(accx, dx[k]); }
        return;
    }
    unsigned bm = bmask;
    for (int s = 0;; ++s) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float m = seg[k] == s ? 1.f : 0.f;
            acc.x = fmaf(m, v[k].x, acc.x); acc.y = fmaf(m, v[k].y, acc.y);
            acc.z = fmaf(m, v[k].z, acc.z); acc.w = fmaf(m, v[k].w, acc.w);
            accx.x = fmaf(m, dx[k].x, accx.x); accx.y = fmaf(m, dx[k].y, accx.y);
            accx.z = fmaf(m, dx[k].z, accx.z);
        }
        if (bm == 0u) break;            // the last segment stays open (carried to the next tile)
        flush(cur_row);
        const int pos = __builtin_ctz(bm);
        bm &= bm - 1u;
        cur_row = __builtin_amdgcn_readfirstlane(rowbuf[pos]);
    }
}

// timing-only ablations (tools/ab.py variants; never built into the shipped library)
#ifdef PVS_ABL_H_NOWGRAD
#define H64_WGRAD(...) ((void)0)
#else
#define H64_WGRAD(...) wgrad16(__VA_ARGS__)
#endif
#ifdef PVS_ABL_H_NOCHAIN
#define H64_CHAIN(T, img, lane, v, acc) do { for (int b_ = 0; b_ < 4; ++b_) acc[b_][0] += __builtin_bit_cast(uint4, v.hi[0]).x * 1e-30f; } while (0)
#define H64_CHAINS(T, img, lane, x, pb, acc) do { split_bf16x3(x, pb); H64_CHAIN(T, img, lane, pb, acc); } while (0)
#else
#define H64_CHAIN(T, img, lane, v, acc) chain16<T>(img, lane, v, acc)
// (the pinned form costs registers: with an edge residual - m_prev and, for the gates, the pre-residual message
// live through the tile - it spills 8-22 VGPRs where split + chain16 spills 0-10; those instantiations keep the
// two-step form)
#define H64_CHAINS(T, img, lane, x, pb, acc)                                        \
    do {                                                                            \
        if constexpr (ERK == 0) chain16s<T>(img, lane, x, pb, acc);                 \
        else { split_bf16x3(x, pb); chain16<T>(img, lane, pb, acc); }               \
    } while (0)
#endif
#ifdef PVS_ABL_H64_WC1_GLOBAL      // timing only (wrong numbers: the words read are node rows, not weight fragments)
#define H64_CHAINS_WC1(T, img, lane, x, pb, acc)                                                                        \
    do {                                                                                                                \
        if constexpr (ERK == 0) chain16s<T, true>(reinterpret_cast<const unsigned short*>(io.PQ), lane, x, pb, acc);    \
        else H64_CHAINS(T, img, lane, x, pb, acc);                                                                      \
    } while (0)
#else
#define H64_CHAINS_WC1(T, img, lane, x, pb, acc) H64_CHAINS(T, img, lane, x, pb, acc)
#endif

// ERK: edge residual kind - 0 none; 1 the plain sum m + m_prev (nothing of the residual has to survive the tile's
// coordinate branch); 2 rezero / gated (the gate's gradient needs the pre-residual message and m_prev at the end).
template <int ERK, bool EATT>
__global__ void __launch_bounds__(256, 1)
k_edge_bwd_h64(PvsGraph g, PvsEdgeW w, uint32_t flags, int att_act, PvsEdgeBwdIO io, int n_chunks, int e_lo, int e_hi) {
    constexpr int H = kH, NT = 256, NW = 4;
    constexpr bool ERES = ERK != 0;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned short* W2i = reinterpret_cast<unsigned short*>(smem);         // 3 parts x [H][H] bf16
    unsigned short* Wc1i = W2i + 3 * H * H;
    float* b2t = smem + 3 * H * H;                                         // (2 images x 3 H^2 shorts = 3 H^2 floats)
    float* bc1t = b2t + H;
    float* wc2t = bc1t + H;
    float* wat = wc2t + H;
    float* wrhot = wat + H;
    float* attrt = wrhot + H;                                  // [PVS_MAX_EDGE_ATTR][H]
    char* wave_base = reinterpret_cast<char*>(attrt + PVS_MAX_EDGE_ATTR * H);

    const bool upd = (flags & PVS_UPDATE_COORDS) && io.gxagg != nullptr;

    stage_weights_y(W2i, w.w2);
    if (upd) stage_weights_y(Wc1i, w.wc1);
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
    const int n = lane & 15, gr = lane >> 4;
    unsigned short* A1I = reinterpret_cast<unsigned short*>(wave_base + wv * kWaveBytes64);
    unsigned short* MI = A1I + kImg16;
    unsigned short* GI = MI + kImg16;
    // once the m image is dead (after the Wc1 weight gradient) its slot holds the g_z1 tile
    float* T1 = reinterpret_cast<float*>(MI);
    float* tx = T1 + kT16 * kTS16;
    int* rowbuf = reinterpret_cast<int*>(tx + kT16 * 4);

    const float bac = EATT ? w.ba[0] : 0.f;
    float gate_raw = 0.f, gate = 1.f;
    if (ERK == 2 && (flags & (PVS_REZERO | PVS_GATED_RESIDUAL))) {
        gate_raw = w.edge_gate[0];
        gate = (flags & PVS_GATED_RESIDUAL) ? fmaxf(gate_raw, 0.f) : gate_raw;
    }
    const float res_a = (flags & (PVS_REZERO | PVS_GATED_RESIDUAL)) ? gate : 1.f;
    const float res_b = (flags & PVS_GATED_RESIDUAL) ? 1.f - gate : 1.f;

    // ---- accumulators that live for the whole kernel ----
    f32x16 gW2[2][2], gWc1[2][2];              // D layout: [c = 32bo + ch(r,hh)][k = 32bi + (lane & 31)]
    float g_b2y[16], g_bc1y[16], g_wc2y[16];   // Y layout (channel in the register, edges on the lanes)
    float g_way[EATT ? 16 : 1];
#pragma unroll
    for (int bo = 0; bo < 2; ++bo)
#pragma unroll
        for (int bi = 0; bi < 2; ++bi)
#pragma unroll
            for (int r = 0; r < 16; ++r) { gW2[bo][bi][r] = 0.f; gWc1[bo][bi][r] = 0.f; }
#pragma unroll
    for (int r = 0; r < 16; ++r) { g_b2y[r] = 0.f; g_bc1y[r] = 0.f; g_wc2y[r] = 0.f; }
#pragma unroll
    for (int r = 0; r < (EATT ? 16 : 1); ++r) g_way[r] = 0.f;
    float g_ba = 0.f, g_gate = 0.f;

    const int total_waves = gridDim.x * NW;
    for (int chunk = pvs_xcd_block(blockIdx.x, gridDim.x) * NW + wv; chunk < n_chunks; chunk += total_waves) {
        const int e_begin = chunk_begin(g, chunk, n_chunks, e_lo, e_hi);
        const int e_end = chunk_begin(g, chunk + 1, n_chunks, e_lo, e_hi);
        int cur_row = -1;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), accx = acc;   // open row: lane = (row slot, quad)
        const int quad = lane & 15, rsub = lane >> 4;
        auto flush = [&](int row_id) {
            if (row_id >= 0) {
                const float4 tot = sum_row_slots<2>(acc);
                if (rsub == 0) *reinterpret_cast<float4*>(io.gPQ + (size_t)row_id * 2 * H + 4 * quad) = tot;
                const float4 tx4 = sum_row_slots<2>(accx);
                if (lane == 0) {
                    io.gx_row[3 * row_id] = tx4.x;
                    io.gx_row[3 * row_id + 1] = tx4.y;
                    io.gx_row[3 * row_id + 2] = tx4.z;
                }
            }
            acc = make_float4(0.f, 0.f, 0.f, 0.f);
            accx = acc;
        };
        // software pipeline: the indices of tile t+2 and the node rows of tile t+1 are in flight while tile t
        // is processed (load_tile_idx clamps every index into the chunk / the arrays)
        TileIdx I = load_tile_idx(g, w.n_attr, e_begin, e_begin, e_end, n);
        TileIdx In = load_tile_idx(g, w.n_attr, e_begin + kT16, e_begin, e_end, n);
        Gather16 G;
        if (e_begin < e_end) gather16<EATT>(io, I, gr, upd, flags & PVS_SOFTMAX_ATT, G);
        // The tile's HBM stores (g_z1 rows, gd records, finished rows of gPQ / gx_row) are issued at the START of
        // the next tile, after that tile's first loads: vmcnt counts loads and stores in one in-order queue, so
        // a wait for a load issued after a store also waits for the store's acknowledgement from HBM, and with
        // one wave per SIMD nothing else runs meanwhile (s_waitcnt was 26 % of the wave's cycles). The g_z1
        // tile, tx and rowbuf stay in LDS until then (their slot - the m image - is next written mid-tile).
        int pend_e0 = -1;
        unsigned pend_bmask = 0u;
        // gW2 += g_z2 (x) a1 of tile t is issued at the top of tile t+1 (wgrad16_beside); the first tile of a chunk
        // multiplies two zeroed images
        for (int k = lane; k < kImg16 * 2 / 16; k += 64) {
            reinterpret_cast<uint4*>(A1I)[k] = make_uint4(0u, 0u, 0u, 0u);
            reinterpret_cast<uint4*>(GI)[k] = make_uint4(0u, 0u, 0u, 0u);
        }
        pvs_wave_lds_sync();
        auto store_phase = [&]() {
            if (gr == 0 && pend_e0 + n < e_end)       // per edge: grad wrt (x_row - x_col), rho and class: 16 B
                pvs_store_nt(io.gd + (size_t)(pend_e0 + n) * 4, *reinterpret_cast<const float4*>(tx + n * 4));
#ifndef PVS_ABL_H_NOSEG
            const int pe0 = pend_e0;
            reduce_rows16(T1, tx, rowbuf, pend_bmask, lane, acc, accx, cur_row, flush,
                          [&](int rl, int q, const float4& v) {
                              if (pe0 + rl < e_end)   // streamed once: non-temporal
                                  pvs_store_nt(io.gz1 + (size_t)(pe0 + rl) * H + 4 * q, v);
                          });
#endif
            pvs_wave_lds_sync();
        };
        for (int e0 = e_begin; e0 < e_end; e0 += kT16) {
            const int e = I.e, ee = I.ee, i = I.i, ty = I.ty;
            const bool valid = I.valid;
            const float vm = valid ? 1.f : 0.f;
            const unsigned bmask = (unsigned)__ballot(valid && gr == 0 && i != I.prev_row) & 0xffffu;
            // (the 16-register row operands are fetched mid-tile, where the pressure allows)
            float gMi[16];
            const float gT0 = upd ? G.gT0 : 0.f, gT1 = upd ? G.gT1 : 0.f, gT2 = upd ? G.gT2 : 0.f;
            const float aval = G.aval, softd = G.softd;      // (softd is only read under PVS_SOFTMAX_ATT)
            float mp[ERES ? 16 : 1];
            if constexpr (ERES) load_y(io.m_prev + (size_t)ee * H, gr, mp);
            if (pend_e0 >= 0) store_phase();           // the previous tile's stores, behind this tile's loads

            const float d0 = G.xi0 - G.xj0, d1 = G.xi1 - G.xj1, d2 = G.xi2 - G.xj2;
            const float rho = d0 * d0 + d1 * d1 + d2 * d2;
            Bf16Parts pb;                     // parts of the tensor being pushed through a product
            float d1r[16];                    // SiLU'(z1)

            // ---- recompute: z1, a1 = SiLU(z1), SiLU'(z1); a1 image; z2 = W2 a1 + b2 ----
            f32x4 acc2[4];
            {
                float a1[16], aa[16], rr[16];
                load_y(attrt + ty * H, gr, aa);
                load_y(wrhot, gr, rr);
#ifndef PVS_ABL_H_NOWGRAD
                {
                    float z[16], ex[16];
                    wgrad16_beside(GI, A1I, lane, gW2,
                                   // (the empty asm statements tie each value's work to its place between two
                                   // MFMAs: sched_barrier pins the machine scheduler, but the optimiser before it
                                   // moves pure arithmetic across the barriers and clumped the MFMAs again)
                                   [&](int r) {
                                       float p = G.P[r];
                                       asm volatile("" : "+v"(p));
                                       z[r] = p + G.Q[r] + fmaf(rr[r], rho, aa[r]);
                                       ex[r] = pvs_exp(-z[r]);
                                       asm volatile("" : "+v"(ex[r]));
                                   },
                                   [&](int r) {
                                       const float sg = pvs_rcp(1.0f + ex[r]);
                                       const float av = z[r] * sg;
                                       d1r[r] = fmaf(av, 1.0f - sg, sg);       // SiLU'(z) = s + z s (1 - s)
                                       a1[r] = av;
                                       asm volatile("" : "+v"(d1r[r]), "+v"(a1[r]));
                                   });
                }
#else
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float z = G.P[r] + G.Q[r] + fmaf(rr[r], rho, aa[r]);
                    const float sg = pvs_sigmoid(z);
                    const float av = z * sg;
                    d1r[r] = fmaf(av, 1.0f - sg, sg);
                    a1[r] = av;
                }
#endif
                float bias[16];
                load_y(b2t, gr, bias);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[r >> 2][r & 3] = bias[r];
                H64_CHAINS(false, W2i, lane, a1, pb, acc2);
                pvs_wave_lds_sync();                                  // the previous tile's a1 image has been read
                write_image16(A1I, n, gr, pb);
            }
            float dz2[16], m[16];             // SiLU'(z2) and the message
            float m_new[ERK == 2 ? 16 : 1];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
#if PVS_PAIR_H64
                pvs_f2 mv, dv;
                pvs_silu_grad2(pvs_f2{acc2[r >> 2][r & 3], acc2[r >> 2][(r & 3) + 1]}, mv, dv);
                m[r] = mv.x; m[r + 1] = mv.y;
                dz2[r] = dv.x; dz2[r + 1] = dv.y;
#else
                for (int t = r; t < r + 2; ++t) {
                    const float z2 = acc2[t >> 2][t & 3];
                    const float sg = pvs_sigmoid(z2);
                    m[t] = z2 * sg;
                    dz2[t] = fmaf(m[t], 1.0f - sg, sg);
                }
#endif
                if constexpr (ERK == 2) { m_new[r] = m[r]; m_new[r + 1] = m[r + 1]; }
            }
            if constexpr (ERES) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if constexpr (ERK == 2) m[r] = fmaf(res_a, m_new[r], res_b * mp[r]);
                    else m[r] += mp[r];
                }
            }

            // ---- gradient wrt m: the coordinate branch's term first, then the external, aggregated-message
            // and attention terms (same order as the H = 32 kernel) ----
            f32x4 gm[4];
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) gm[b][r] = 0.f;
            float s_coord = 0.f, nrm = 1.f;
            if (upd) {
                f32x4 accc[4];
                {
                    float bias2[16];
                    load_y(bc1t, gr, bias2);
#pragma unroll
                    for (int r = 0; r < 16; ++r) accc[r >> 2][r & 3] = bias2[r];
                }
                H64_CHAINS_WC1(false, Wc1i, lane, m, pb, accc);        // zc = Wc1 m + bc1
                write_image16(MI, n, gr, pb);
                float wc2y[16];
                load_y(wc2t, gr, wc2y);
                float q[16], dq[16];
                float s = 0.f;
#if PVS_PAIR_H64
                pvs_f2 s2{0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    pvs_f2 qv, dv;
                    pvs_silu_grad2(pvs_f2{accc[r >> 2][r & 3], accc[r >> 2][(r & 3) + 1]}, qv, dv);
                    q[r] = qv.x; q[r + 1] = qv.y;
                    dq[r] = dv.x; dq[r + 1] = dv.y;
                    s2 = pvs_fma2(pvs_f2{wc2y[r], wc2y[r + 1]}, qv, s2);
                }
                s = s2.x + s2.y;
#else
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float zc = accc[r >> 2][r & 3];
                    const float sg = pvs_sigmoid(zc);
                    q[r] = zc * sg;
                    dq[r] = fmaf(q[r], 1.0f - sg, sg);
                    s = fmaf(wc2y[r], q[r], s);
                }
#endif
                s = sum_groups(s);
                float dact = 1.f;
                if (flags & PVS_TANH) { s = pvs_tanh(s); dact = 1.f - s * s; }
                if (flags & PVS_NORMALIZE) nrm = 1.f / (sqrtf(rho) + 1e-8f);
                s_coord = s;
                const float g_s = (d0 * gT0 + d1 * gT1 + d2 * gT2) * nrm * dact * vm;
                float g_zc[16];
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
#if PVS_PAIR_H64
                    const pvs_f2 gz = pvs_f2{wc2y[r], wc2y[r + 1]} * g_s * pvs_f2{dq[r], dq[r + 1]};
                    const pvs_f2 gw = pvs_fma2(pvs_f2{q[r], q[r + 1]}, pvs_f2{g_s, g_s}, pvs_f2{g_wc2y[r], g_wc2y[r + 1]});
                    const pvs_f2 gb = pvs_f2{g_bc1y[r], g_bc1y[r + 1]} + gz;
                    g_zc[r] = gz.x; g_zc[r + 1] = gz.y;
                    g_wc2y[r] = gw.x; g_wc2y[r + 1] = gw.y;
                    g_bc1y[r] = gb.x; g_bc1y[r + 1] = gb.y;
#else
                    for (int t = r; t < r + 2; ++t) {
                        g_zc[t] = g_s * wc2y[t] * dq[t];
                        g_wc2y[t] = fmaf(g_s, q[t], g_wc2y[t]);
                        g_bc1y[t] += g_zc[t];
                    }
#endif
                }
                load_y(io.gM + (size_t)i * H, gr, gMi);               // (in flight behind the two products below)
                H64_CHAINS_WC1(true, Wc1i, lane, g_zc, pb, gm);        // g_m += Wc1^T g_zc
                write_image16(GI, n, gr, pb);
                pvs_wave_lds_sync();                                  // the m and g_zc images are complete
                H64_WGRAD(GI, MI, lane, gWc1);                          // gWc1 += g_zc (x) m
            } else {
                load_y(io.gM + (size_t)i * H, gr, gMi);
            }
            {   // external, aggregated-message and attention terms
                if (io.g_m_out) {
                    float init[16];
                    load_y(io.g_m_out + (size_t)ee * H, gr, init);
#pragma unroll
                    for (int r = 0; r < 16; ++r) gm[r >> 2][r & 3] = fmaf(init[r], vm, gm[r >> 2][r & 3]);
                }
                if constexpr (EATT) {
                    float way[16];
                    load_y(wat, gr, way);
                    float logit = 0.f, dot = 0.f;
#if PVS_PAIR_H64
                    pvs_f2 l2{0.f, 0.f}, d2p{0.f, 0.f};
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const pvs_f2 mv{m[r], m[r + 1]};
                        l2 = pvs_fma2(pvs_f2{way[r], way[r + 1]}, mv, l2);
                        d2p = pvs_fma2(mv, pvs_f2{gMi[r], gMi[r + 1]}, d2p);
                    }
                    logit = l2.x + l2.y;
                    dot = d2p.x + d2p.y;
#else
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        logit = fmaf(way[r], m[r], logit);
                        dot = fmaf(m[r], gMi[r], dot);
                    }
#endif
                    logit = sum_groups(logit) + bac;
                    dot = sum_groups(dot);
                    const float g_l = (flags & PVS_SOFTMAX_ATT) ? aval * (dot - softd) * vm     // softD = M_i . g_M_i
                                                                : pvs_att_act_grad(att_act, logit, aval) * dot * vm;
                    if (gr == 0) g_ba += g_l;
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
#if PVS_PAIR_H64
                        const pvs_f2 t = pvs_fma2(pvs_f2{gMi[r], gMi[r + 1]}, pvs_f2{aval * vm, aval * vm}, pvs_f2{way[r], way[r + 1]} * g_l);
                        const pvs_f2 gw = pvs_fma2(pvs_f2{m[r], m[r + 1]}, pvs_f2{g_l, g_l}, pvs_f2{g_way[r], g_way[r + 1]});
                        gm[r >> 2][r & 3] += t.x; gm[r >> 2][(r & 3) + 1] += t.y;
                        g_way[r] = gw.x; g_way[r + 1] = gw.y;
#else
                        for (int t = r; t < r + 2; ++t) {
                            gm[t >> 2][t & 3] += (aval * vm) * gMi[t] + g_l * way[t];
                            g_way[t] = fmaf(g_l, m[t], g_way[t]);
                        }
#endif
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
#if PVS_PAIR_H64
                        const pvs_f2 t = pvs_fma2(pvs_f2{gMi[r], gMi[r + 1]}, pvs_f2{vm, vm}, pvs_f2{gm[r >> 2][r & 3], gm[r >> 2][(r & 3) + 1]});
                        gm[r >> 2][r & 3] = t.x; gm[r >> 2][(r & 3) + 1] = t.y;
#else
                        gm[r >> 2][r & 3] = fmaf(vm, gMi[r], gm[r >> 2][r & 3]);
                        gm[(r + 1) >> 2][(r + 1) & 3] = fmaf(vm, gMi[r + 1], gm[(r + 1) >> 2][(r + 1) & 3]);
#endif
                    }
                }
            }
            // ---- software pipeline: the node rows of tile t+1 and the indices of tile t+2 are fetched from here
            // on: the register peak - the coordinate branch - is over, ~4k cycles of work remain to cover them, and the
            // wait for gM_i above did not have to count (or, after the branch merge, drain) these loads ----
            Gather16 Gn;
            gather16<EATT>(io, In, gr, upd, flags & PVS_SOFTMAX_ATT, Gn);
            const TileIdx Inn = load_tile_idx(g, w.n_attr, e0 + 2 * kT16, e_begin, e_end, n);
            // ---- edge residual; g_z2 = g_m_new * SiLU'(z2) ----
            float g_z2[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float gmv = gm[r >> 2][r & 3];
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
                if (!(PVS_PAIR_H64 && ERK == 0)) g_b2y[r] += g_z2[r];
            }
            if constexpr (PVS_PAIR_H64 && ERK == 0) {      // (g_z2 = g_m * SiLU'(z2) again, two at a time: the loop above folds away)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const pvs_f2 z = pvs_f2{gm[r >> 2][r & 3], gm[r >> 2][(r & 3) + 1]} * pvs_f2{dz2[r], dz2[r + 1]};
                    const pvs_f2 b = pvs_f2{g_b2y[r], g_b2y[r + 1]} + z;
                    g_z2[r] = z.x; g_z2[r + 1] = z.y;
                    g_b2y[r] = b.x; g_b2y[r + 1] = b.y;
                }
            }
            if constexpr (ERES) {
                if (valid) store_y(io.g_m_prev + (size_t)e * H, gr, mp);
            }
            // ---- g_a1 = W2^T g_z2 ; gW2 += g_z2 (x) a1 ; g_z1 = g_a1 * SiLU'(z1) ----
            f32x4 ga1[4];
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) ga1[b][r] = 0.f;
            H64_CHAINS(true, W2i, lane, g_z2, pb, ga1);
            pvs_wave_lds_sync();                                      // the g_zc image has been read
            write_image16(GI, n, gr, pb);
            // (gW2 += g_z2 (x) a1: at the top of the next tile / behind the loop)
            float g_z1[16];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
#if PVS_PAIR_H64
                const pvs_f2 z = pvs_f2{ga1[r >> 2][r & 3], ga1[r >> 2][(r & 3) + 1]} * pvs_f2{d1r[r], d1r[r + 1]};
                g_z1[r] = z.x; g_z1[r + 1] = z.y;
#else
                g_z1[r] = ga1[r >> 2][r & 3] * d1r[r];
                g_z1[r + 1] = ga1[(r + 1) >> 2][(r + 1) & 3] * d1r[r + 1];
#endif
            }
            float g_rho;
            {
                float rr[16];
                load_y(wrhot, gr, rr);
                float s = 0.f;
#if PVS_PAIR_H64
                pvs_f2 s2{0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 16; r += 2) s2 = pvs_fma2(pvs_f2{rr[r], rr[r + 1]}, pvs_f2{g_z1[r], g_z1[r + 1]}, s2);
                s = s2.x + s2.y;
#else
#pragma unroll
                for (int r = 0; r < 16; ++r) s = fmaf(rr[r], g_z1[r], s);
#endif
                g_rho = sum_groups(s);
            }
            const float k1 = s_coord * nrm * vm;
            const float gd0 = fmaf(k1, gT0, 2.f * d0 * g_rho);
            const float gd1 = fmaf(k1, gT1, 2.f * d1 * g_rho);
            const float gd2 = fmaf(k1, gT2, 2.f * d2 * g_rho);
            pvs_wave_lds_sync();          // every read of the m image (its slot becomes the g_z1 tile) is done
            // per edge: grad wrt (x_row - x_col) and rho (+ class), 16 B, for the node gather kernel
            if (gr == 0) {
                *reinterpret_cast<float4*>(tx + n * 4) = make_float4(gd0, gd1, gd2, pvs_pack_rho_type(rho, ty));
                rowbuf[n] = i;
            }
            // ---- g_z1 edge-major: whole rows go to HBM, and the row-side sums come from the same reads, in the
            // deferred store phase ----
            store_y(T1 + n * kTS16, gr, g_z1);
            pend_e0 = e0;
            pend_bmask = bmask;
            I = In;
            In = Inn;
            G = Gn;
            pvs_wave_lds_sync();
        }
        if (pend_e0 >= 0) {
            pvs_wave_lds_sync();
            H64_WGRAD(GI, A1I, lane, gW2);                              // the last tile's product
            store_phase();
        }
        flush(cur_row);
    }

    // ---- block reduction into one slab, fixed order ----
    const PvsSlabLayout L = pvs_slab_layout(H);
    __syncthreads();
    float* slab = smem;
    for (int i = threadIdx.x; i < L.total; i += NT) slab[i] = 0.f;
    __syncthreads();
    // Y-layout vectors: sum over the 16 edge lanes of each group
    auto lanes16 = [](float v) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o, 64);
        return v;
    };
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        g_b2y[r] = lanes16(g_b2y[r]);
        g_bc1y[r] = lanes16(g_bc1y[r]);
        g_wc2y[r] = lanes16(g_wc2y[r]);
    }
#pragma unroll
    for (int r = 0; r < (EATT ? 16 : 1); ++r) g_way[r] = lanes16(g_way[r]);
    g_ba = lanes16(g_ba);                      // only group 0 accumulated
    g_ba += __shfl_xor(g_ba, 16, 64);
    g_ba += __shfl_xor(g_ba, 32, 64);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) g_gate += __shfl_xor(g_gate, o, 64);
    const int j = lane & 31, hh = lane >> 5;
    for (int turn = 0; turn < NW; ++turn) {
        if (wv == turn) {
#pragma unroll
            for (int bo = 0; bo < 2; ++bo)
#pragma unroll
                for (int bi = 0; bi < 2; ++bi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int c = 32 * bo + xch(r, hh), k = 32 * bi + j;
                        slab[L.w2 + c * H + k] += gW2[bo][bi][r];
                        slab[L.wc1 + c * H + k] += gWc1[bo][bi][r];
                    }
            if (n == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int c = ych(r, gr);
                    slab[L.b2 + c] += g_b2y[r];
                    slab[L.bc1 + c] += g_bc1y[r];
                    slab[L.wc2 + c] += g_wc2y[r];
                    if constexpr (EATT) slab[L.wa + c] += g_way[r];
                }
            }
            if (lane == 0) { slab[L.ba] += g_ba; slab[L.gate] += g_gate; }
        }
        __syncthreads();
    }
    float* dst = io.slabs + (size_t)blockIdx.x * L.total;
    for (int i = threadIdx.x; i < L.total; i += NT) dst[i] = slab[i];
}

}  // namespace

// Same contract as pvs_launch_edge_bwd_mfma (edge_mfma.hip), H = 64 only.
int pvs_launch_edge_bwd_h64(hipStream_t s, const PvsGraph& g, const PvsEdgeW& w, uint32_t flags, int att_act,
                            const PvsEdgeBwdIO& io, int e_lo, int e_hi, int* n_slabs) {
    PVS_REQUIRE(w.n_attr <= 3, "MFMA edge backward supports up to 3 edge classes (got %d)", w.n_attr);
    *n_slabs = 0;
    if (e_hi <= e_lo) return 0;
    constexpr int nw = 4;
    int blocks, n_chunks;
    {
        const int E = e_hi - e_lo;
        const long long per = pvs_edges_per_wave();
        long long b = ((long long)E + (long long)nw * per - 1) / ((long long)nw * per);   // fill the chip first
        if (b < 1) b = 1;
        if (b > 256) b = 256;                      // one workgroup per CU (registers: one wave per SIMD)
        const long long waves = b * nw;
        long long per_wave = ((long long)E + waves * 4096 - 1) / (waves * 4096);
        if (per_wave < 1) per_wave = 1;
        blocks = (int)b;
        n_chunks = (int)(waves * per_wave);
    }
    *n_slabs = blocks;
    PvsProfScope prof(s, PVS_PROF_EDGE_BWD);
    const PvsSlabLayout L = pvs_slab_layout(kH);
    size_t lds = (size_t)kSharedBytes64 + (size_t)nw * kWaveBytes64;
    if (lds < (size_t)L.total * 4) lds = (size_t)L.total * 4;
    const bool eres = (flags & PVS_EDGE_RESIDUAL) && io.m_prev != nullptr;
    const bool eatt = flags & PVS_EDGE_ATTENTION;
#define PVS_BWD_H64_LAUNCH(ER, EA)                                                                      \
    do {                                                                                               \
        if (set_lds(k_edge_bwd_h64<ER, EA>, lds)) return -2;                                           \
        k_edge_bwd_h64<ER, EA><<<blocks, 256, lds, s>>>(g, w, flags, att_act, io, n_chunks, e_lo, e_hi); \
    } while (0)
    const bool gated = flags & (PVS_REZERO | PVS_GATED_RESIDUAL);
    if (eres && gated && eatt) PVS_BWD_H64_LAUNCH(2, true);
    else if (eres && gated) PVS_BWD_H64_LAUNCH(2, false);
    else if (eres && eatt) PVS_BWD_H64_LAUNCH(1, true);
    else if (eres) PVS_BWD_H64_LAUNCH(1, false);
    else if (eatt) PVS_BWD_H64_LAUNCH(0, true);
    else PVS_BWD_H64_LAUNCH(0, false);
#undef PVS_BWD_H64_LAUNCH
    PVS_CHECK_LAUNCH();
    return 0;
}
