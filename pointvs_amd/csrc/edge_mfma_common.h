// Device helpers shared by the MFMA edge kernels (edge_mfma_fwd.hip, edge_mfma.hip): weight staging,
// X-layout MFMA chains, tile index/gather helpers, the row (segment) reduction of an edge-major LDS
// tile and the bf16x3 split. See edge_mfma_fwd.hip for the layout description.
#pragma once
#include "edge_kernels.h"
#include "mfma_common.h"
#include "profile.h"

#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = 4;
constexpr int kTile = 32;

// Timing-only ablation switches (PVS_ABLATE env, tools/ablate.py): results are wrong when set.
// internal: the forward writes the raw coordinate sums (no x, no 1/deg) into x_out (edge_sums)
constexpr uint32_t kFwdRawXsum = 1u << 23;
constexpr uint32_t kAblNoMfma = 1u << 24, kAblNoSilu = 1u << 25, kAblNoReduce = 1u << 26,
                   kAblNoGather = 1u << 27;

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
                                           const float (&v)[HB][16], f32x16 (&acc)[HB],
                                           bool skip = false) {
    if (skip) {   // ablation: keep the operands live, issue no MFMA
#pragma unroll
        for (int b = 0; b < HB; ++b) acc[b][0] += v[b][0];
        return;
    }
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

// Natural row-major staging W[c*(H+1) + k] (row stride padded by one word): ONE copy serves both
// Z = W V (lanes vary the row: stride H+1 -> distinct banks) and Z = W^T V (lanes vary the column:
// consecutive words), halving the LDS the backward needs for its four operand orientations.
template <int HB>
__device__ __forceinline__ void stage_weights_nat(float* dst, const float* __restrict__ W) {
    constexpr int H = 32 * HB;
    for (int i = threadIdx.x; i < H * H; i += blockDim.x) dst[(i / H) * (H + 1) + (i % H)] = W[i];
}

template <int HB, bool TRANSPOSE>
__device__ __forceinline__ void mfma_chain_nat(const float* __restrict__ Wn, int lane,
                                               const float (&v)[HB][16], f32x16 (&acc)[HB],
                                               bool skip = false) {
    constexpr int H = 32 * HB, LD = H + 1;
    if (skip) {
#pragma unroll
        for (int b = 0; b < HB; ++b) acc[b][0] += v[b][0];
        return;
    }
    const int j = lane & 31, hh = lane >> 5;
    const float* base = TRANSPOSE ? Wn + (4 * hh) * LD + j : Wn + j * LD + 4 * hh;
#pragma unroll
    for (int bo = 0; bo < HB; ++bo)
#pragma unroll
        for (int bi = 0; bi < HB; ++bi)
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int kc = 32 * bi + (t & 3) + 8 * (t >> 2);   // + 4hh folded into base
                const float a = TRANSPOSE ? base[kc * LD + 32 * bo] : base[(32 * bo) * LD + kc];
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

// v + (the same register of lane ^ 32 / ^ 16 / ^ 8) without LDS. v_permlane32_swap / v_permlane16_swap with both
// operands the same register return the low-half (even-row) value in one result and the high-half (odd-row) value
// in the other, in EVERY lane; lane ^ 8 is a rotation by 8 inside a row of 16 (DPP row_ror:8). Bit-identical to
// v + __shfl_xor(v, o) (one commutative fp32 addition per lane), but a vector instruction instead of a
// ds_bpermute round trip through LDS - which a wave that shares its SIMD with at most one other waits out.
__device__ __forceinline__ float pvs_xor32_sum(float v) {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float pvs_xor16_sum(float v) {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float pvs_xor8_sum(float v) {
    const int u = __float_as_int(v);
    return v + __int_as_float(__builtin_amdgcn_update_dpp(u, u, 0x128 /* row_ror:8 */, 0xf, 0xf, false));
}
template <int O>
__device__ __forceinline__ float pvs_xor_sum(float v) {
    static_assert(O == 8 || O == 16 || O == 32, "lane distance");
    if constexpr (O == 8) return pvs_xor8_sum(v);
    else if constexpr (O == 16) return pvs_xor16_sum(v);
    else return pvs_xor32_sum(v);
}

// Pair arithmetic (common.h pvs_f2): the elementwise work of the edge kernels on two adjacent registers per instruction.
#ifndef PVS_PAIR_MATH
#define PVS_PAIR_MATH 1
#endif
// H = 32 only. At 128 channels register pairs cost the softmax forward its second wave per SIMD; the 64-channel forward
// (768 threads, 168 registers) is 1 % slower with them (profiles/r05_ab_pair_math.txt; -DPVS_PAIR_MAX_HB=2 builds it).
#ifndef PVS_PAIR_MAX_HB
#define PVS_PAIR_MAX_HB 1
#endif
template <int HB> constexpr bool pvs_pair_math = PVS_PAIR_MATH && HB <= PVS_PAIR_MAX_HB;

template <int HB>
__device__ __forceinline__ float dot_tab(const float* __restrict__ tab, int hh, const float (&v)[HB][16]) {
    float s = 0.f;
    if constexpr (pvs_pair_math<HB>) {
        pvs_f2 s2{0.f, 0.f};            // (even and odd channels apart: half the instructions of one running sum)
#pragma unroll
        for (int b = 0; b < HB; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 w = *reinterpret_cast<const float4*>(tab + 32 * b + 8 * g + 4 * hh);
                s2 = pvs_fma2(pvs_f2{w.x, w.y}, pvs_f2{v[b][4 * g], v[b][4 * g + 1]}, s2);
                s2 = pvs_fma2(pvs_f2{w.z, w.w}, pvs_f2{v[b][4 * g + 2], v[b][4 * g + 3]}, s2);
            }
        s = s2.x + s2.y;
    } else {
#pragma unroll
        for (int b = 0; b < HB; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 w = *reinterpret_cast<const float4*>(tab + 32 * b + 8 * g + 4 * hh);
                s = fmaf(w.x, v[b][4 * g], s); s = fmaf(w.y, v[b][4 * g + 1], s);
                s = fmaf(w.z, v[b][4 * g + 2], s); s = fmaf(w.w, v[b][4 * g + 3], s);
            }
    }
    return pvs_xor32_sum(s);            // other half holds the other 16 channels of each block
}

// Wave chunks of the edge range [e_lo, e_hi) (row-aligned ends): chunk k starts at the row that
// contains edge e_lo + k*(e_hi-e_lo)/n_chunks, so every row is owned by exactly one wave.
__device__ __forceinline__ int chunk_begin(const PvsGraph& g, int k, int n_chunks, int e_lo, int e_hi) {
    if (k <= 0) return e_lo;
    if (k >= n_chunks) return e_hi;
    const long long t = e_lo + (long long)k * (e_hi - e_lo) / n_chunks;
#ifdef PVS_ABL_UNALIGNED_CHUNKS       // timing only (rows that straddle a boundary are summed wrongly): what perfectly equal chunks would give
    return (int)t;
#endif
    return max(e_lo, g.rowptr[g.row[t]]);
}

// Indices of one 32-edge tile (lane = edge slot j, both halves hold the same values).
struct TileIdx {
    int e, ee, i, jn, ty, prev_row;
    bool valid;
};

__device__ __forceinline__ TileIdx load_tile_idx(const PvsGraph& g, int n_attr, int e0, int e_begin,
                                                 int e_end, int j) {
    TileIdx t;
    t.e = e0 + j;
    t.valid = t.e < e_end;
    t.ee = t.valid ? t.e : e_end - 1;
    // an EMPTY chunk at the start of the range (a first row longer than several chunk slots) has
    // e_end = 0: its index loads are never used, but they must stay inside the arrays
    t.ee = min(max(t.ee, 0), g.n_edges - 1);
    t.i = g.row[t.ee];
    t.jn = g.col[t.ee];
    if (n_attr & 0x100) { t.i &= 7; t.jn &= 7; }   // ablation: every gather hits 8 hot rows
    t.ty = (n_attr & 0xff) ? (int)g.etype[t.ee] : 0;
    t.prev_row = (t.ee == e_begin || t.ee == 0) ? -1 : g.row[t.ee - 1];
    return t;
}

// ---- 32-bit lane offsets (round 5) -----------------------------------------------------------------------------------
// base + (size_t)index * stride makes the compiler form a 64-bit address PER LANE (a sign extension and a
// v_lshl_add_u64 per access, two registers per live address: 37 vector instructions and ten kernel-lifetime registers in
// the H = 32 backward's tile loop). A wave-uniform base (kernel argument, or argument + a wave-uniform element offset,
// formed in scalar registers) plus an UNSIGNED 32-bit byte offset selects the `global_load v, v_off, s[base:base+1]`
// form instead: one register and at most one instruction per address. Callers bound the offsets: node tables below
// 2^32 bytes (the launchers check N), per-edge arrays addressed relative to the tile's first edge.
#ifndef PVS_SADDR
#define PVS_SADDR 1
#endif
template <class T>
__device__ __forceinline__ T* pvs_off(T* base, unsigned bytes) {
    if (!PVS_SADDR) return reinterpret_cast<T*>(reinterpret_cast<char*>(const_cast<typename std::remove_const<T>::type*>(base)) + (long long)(int)bytes);
    return reinterpret_cast<T*>(reinterpret_cast<char*>(const_cast<typename std::remove_const<T>::type*>(base)) + bytes);
}

// load_tile_idx with 32-bit offsets (the same values)
__device__ __forceinline__ TileIdx load_tile_idx32(const PvsGraph& g, int n_attr, int e0, int e_begin, int e_end, int j) {
    TileIdx t;
    t.e = e0 + j;
    t.valid = t.e < e_end;
    t.ee = t.valid ? t.e : e_end - 1;
    t.ee = min(max(t.ee, 0), g.n_edges - 1);
    const unsigned o = 4u * (unsigned)t.ee;
    t.i = *pvs_off(g.row, o);
    t.jn = *pvs_off(g.col, o);
    if (n_attr & 0x100) { t.i &= 7; t.jn &= 7; }   // ablation: every gather hits 8 hot rows
    t.ty = (n_attr & 0xff) ? (int)*pvs_off(g.etype, (unsigned)t.ee) : 0;
    t.prev_row = (t.ee == e_begin || t.ee == 0) ? -1 : *pvs_off(g.row, o - 4u);
    return t;
}

// Gathered node data of one tile in X layout: P_i and Q_j rows, coordinate difference.
template <int HB>
struct TileGather {
    float P[HB][16], Q[HB][16];
    float d0, d1, d2;
};

template <int HB>
__device__ __forceinline__ void gather_tile(const float* __restrict__ PQ, const float* __restrict__ x,
                                            const TileIdx& t, int hh, TileGather<HB>& G) {
    constexpr int H = 32 * HB;
    const float* Pp = PQ + (size_t)t.i * 2 * H + 4 * hh;
    const float* Qp = PQ + (size_t)t.jn * 2 * H + H + 4 * hh;
#pragma unroll
    for (int b = 0; b < HB; ++b)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const float4 p = *reinterpret_cast<const float4*>(Pp + 32 * b + 8 * gq);
            const float4 q = *reinterpret_cast<const float4*>(Qp + 32 * b + 8 * gq);
            G.P[b][4 * gq] = p.x; G.P[b][4 * gq + 1] = p.y; G.P[b][4 * gq + 2] = p.z; G.P[b][4 * gq + 3] = p.w;
            G.Q[b][4 * gq] = q.x; G.Q[b][4 * gq + 1] = q.y; G.Q[b][4 * gq + 2] = q.z; G.Q[b][4 * gq + 3] = q.w;
        }
    G.d0 = x[3 * t.i] - x[3 * t.jn];
    G.d1 = x[3 * t.i + 1] - x[3 * t.jn + 1];
    G.d2 = x[3 * t.i + 2] - x[3 * t.jn + 2];
}

// gather_tile with 32-bit offsets (node tables below 2^32 bytes)
template <int HB>
__device__ __forceinline__ void gather_tile32(const float* __restrict__ PQ, const float* __restrict__ x,
                                              const TileIdx& t, int hh, TileGather<HB>& G) {
    constexpr int H = 32 * HB;
    const float* Pp = pvs_off(PQ, (unsigned)t.i * (8u * H) + 16u * hh);
    const float* Qp = pvs_off(PQ, (unsigned)t.jn * (8u * H) + 4u * H + 16u * hh);
#pragma unroll
    for (int b = 0; b < HB; ++b)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const float4 p = *reinterpret_cast<const float4*>(Pp + 32 * b + 8 * gq);
            const float4 q = *reinterpret_cast<const float4*>(Qp + 32 * b + 8 * gq);
            G.P[b][4 * gq] = p.x; G.P[b][4 * gq + 1] = p.y; G.P[b][4 * gq + 2] = p.z; G.P[b][4 * gq + 3] = p.w;
            G.Q[b][4 * gq] = q.x; G.Q[b][4 * gq + 1] = q.y; G.Q[b][4 * gq + 2] = q.z; G.Q[b][4 * gq + 3] = q.w;
        }
    const float* xi = pvs_off(x, 12u * (unsigned)t.i);
    const float* xj = pvs_off(x, 12u * (unsigned)t.jn);
    G.d0 = xi[0] - xj[0];
    G.d1 = xi[1] - xj[1];
    G.d2 = xi[2] - xj[2];
}

// z1 = P_i + Q_j + w_rho * rho + W_a[type]  (X layout)
template <int HB>
__device__ __forceinline__ void assemble_z1(const TileGather<HB>& G, const float* __restrict__ attrt,
                                            const float* __restrict__ wrhot, int ty, int hh, float rho,
                                            float (&z1)[HB][16]) {
    constexpr int H = 32 * HB;
    const float* At = attrt + ty * H + 4 * hh;
    const float* Rt = wrhot + 4 * hh;
#pragma unroll
    for (int b = 0; b < HB; ++b)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const float4 a = *reinterpret_cast<const float4*>(At + 32 * b + 8 * gq);
            const float4 r = *reinterpret_cast<const float4*>(Rt + 32 * b + 8 * gq);
            if constexpr (pvs_pair_math<HB>) {
                const pvs_f2 rho2{rho, rho};
                const pvs_f2 lo = pvs_f2{G.P[b][4 * gq], G.P[b][4 * gq + 1]} + pvs_f2{G.Q[b][4 * gq], G.Q[b][4 * gq + 1]} +
                                  pvs_fma2(pvs_f2{r.x, r.y}, rho2, pvs_f2{a.x, a.y});
                const pvs_f2 hi = pvs_f2{G.P[b][4 * gq + 2], G.P[b][4 * gq + 3]} + pvs_f2{G.Q[b][4 * gq + 2], G.Q[b][4 * gq + 3]} +
                                  pvs_fma2(pvs_f2{r.z, r.w}, rho2, pvs_f2{a.z, a.w});
                z1[b][4 * gq] = lo.x; z1[b][4 * gq + 1] = lo.y; z1[b][4 * gq + 2] = hi.x; z1[b][4 * gq + 3] = hi.y;
                continue;
            }
            z1[b][4 * gq] = G.P[b][4 * gq] + G.Q[b][4 * gq] + fmaf(r.x, rho, a.x);
            z1[b][4 * gq + 1] = G.P[b][4 * gq + 1] + G.Q[b][4 * gq + 1] + fmaf(r.y, rho, a.y);
            z1[b][4 * gq + 2] = G.P[b][4 * gq + 2] + G.Q[b][4 * gq + 2] + fmaf(r.z, rho, a.z);
            z1[b][4 * gq + 3] = G.P[b][4 * gq + 3] + G.Q[b][4 * gq + 3] + fmaf(r.w, rho, a.w);
        }
}


// ---- row (segment) reduction of one edge-major LDS tile -------------------------------------------
// T[32][TS] holds one H-vector per edge of the tile, tx[32][4] one float4 per edge, rowbuf[32] the
// row id of each edge. Lane = (row slot rsub, 16-byte quad): each lane reads whole float4s, so the
// same LDS reads feed both the per-row sums and (optionally) fully coalesced 128-byte row stores to
// HBM. bmask bit e = "edge e starts a new row" (wave-uniform), so segments are handled by scalar
// control flow: the first segment continues the carried row, every later one starts at a set bit.
// acc/accx carry the open row's partial sums (per lane: its quad, summed over its row slots);
// flush(row) reduces them over the row slots, stores and clears.
// SWZ (round 5, H = 32 backward): T[32][H] without padding, the 16-byte quads of row r rotated by r & (H/4 - 1): the b128
// reads below are served in lane groups {0-3, 12-15, 20-27}, ... (MI355X_MICROARCH.md, LDS), for which the padded stride
// H + 4 is a 2-way conflict and the rotated unpadded rows are conflict-free (tools/lds_conflicts.py); the writer uses
// pvs_tile_quad_off() too.
template <int HB>
__device__ __forceinline__ int pvs_tile_quad_off(int row, int quad) {      // float offset of quad `quad` of row `row`
    constexpr int H = 32 * HB;
    return row * H + 4 * (quad ^ (row & (H / 4 - 1)));
}

template <int HB, bool WSUM = false, bool ROWS = true, bool SWZ = false, class Flush, class RowStore>
__device__ __forceinline__ void reduce_rows_tile(const float* __restrict__ T, const float* __restrict__ tx,
                                                 const int* __restrict__ rowbuf, unsigned bmask, int lane,
                                                 float4& acc, float4& accx, int& cur_row, Flush&& flush,
                                                 RowStore&& store_row) {
    constexpr int H = 32 * HB, TS = H + 4;
    constexpr int QPR = H / 4, RPI = 64 / QPR, NK = kTile / RPI;
    const int quad = lane % QPR, rsub = lane / QPR;
    float4 v[NK], dx[NK];
    int seg[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int rl = k * RPI + rsub;
        if constexpr (ROWS) v[k] = *reinterpret_cast<const float4*>(T + (SWZ ? pvs_tile_quad_off<HB>(rl, quad) : rl * TS + 4 * quad));      // (!ROWS: only the
        else v[k] = make_float4(0.f, 0.f, 0.f, 0.f);                        // per-edge float4 of tx is reduced)
        dx[k] = *reinterpret_cast<const float4*>(tx + rl * 4);
        store_row(rl, quad, v[k]);
        const unsigned upto = rl == 31 ? 0xffffffffu : ((2u << rl) - 1u);
        seg[k] = __popc(bmask & upto);
    }
    auto add4 = [](float4& a, const float4& b) {
        if constexpr (pvs_pair_math<HB>) {
            const pvs_f2 lo = pvs_f2{a.x, a.y} + pvs_f2{b.x, b.y}, hi = pvs_f2{a.z, a.w} + pvs_f2{b.z, b.w};
            a.x = lo.x; a.y = lo.y; a.z = hi.x; a.w = hi.y;
        } else {
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
    };
    if (bmask == 0u) {
#pragma unroll
        for (int k = 0; k < NK; ++k) { add4(acc, v[k]); add4(accx, dx[k]); }
        return;
    }
    unsigned bm = bmask;
    for (int s = 0;; ++s) {
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const float m = seg[k] == s ? 1.f : 0.f;
            acc.x = fmaf(m, v[k].x, acc.x); acc.y = fmaf(m, v[k].y, acc.y);
            acc.z = fmaf(m, v[k].z, acc.z); acc.w = fmaf(m, v[k].w, acc.w);
            accx.x = fmaf(m, dx[k].x, accx.x); accx.y = fmaf(m, dx[k].y, accx.y);
            accx.z = fmaf(m, dx[k].z, accx.z);
            if constexpr (WSUM) accx.w = fmaf(m, dx[k].w, accx.w);   // (softmax: the weights' sum)
        }
        if (bm == 0u) break;            // the last segment stays open (carried to the next tile)
        flush(cur_row);
        const int pos = __builtin_ctz(bm);
        bm &= bm - 1u;
        cur_row = __builtin_amdgcn_readfirstlane(rowbuf[pos]);
    }
}

// sum a float4 over the row slots (lanes that share a quad)
template <int HB>
__device__ __forceinline__ float4 sum_row_slots(float4 a) {
    auto all = [&](auto f) { a.x = f(a.x); a.y = f(a.y); a.z = f(a.z); a.w = f(a.w); };
    if constexpr (HB == 1) all([](float v) { return pvs_xor8_sum(v); });
    if constexpr (HB <= 2) all([](float v) { return pvs_xor16_sum(v); });
    all([](float v) { return pvs_xor32_sum(v); });
    return a;
}


// ---- fp32 products as 6 bf16 MFMA terms ("bf16x3") -------------------------------------------------
// x = hi + mid + lo with each part the next 8 significant bits of x (truncation: exact 24-bit
// split), so a*b = hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi + O(2^-25 |a||b|): fp32-level
// accuracy from v_mfma_f32_32x32x16_bf16 (fp32 accumulate), 12 bf16 MFMAs of 32 cycles per
// 32x32x32 block instead of 16 fp32 MFMAs of 64 cycles, on the matrix cores proper.
// k-step s of the bf16 instruction takes X-layout registers 8s..8s+7 as its 8 B-operand elements
// (k = 8*hh + j'), and the A operand staged with the same channel order.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Bf16Parts { bf16x8 hi[2], mid[2], lo[2]; };

__device__ __forceinline__ unsigned pvs_pack_hi16(float x0, float x1) {
    // bf16 (truncated) of x0 in the low half, of x1 in the high half
    return __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u);
}

__device__ __forceinline__ void split_bf16x3(const float (&v)[16], Bf16Parts& out) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        uint4 ph, pm, pl;
        unsigned* h = reinterpret_cast<unsigned*>(&ph);
        unsigned* m = reinterpret_cast<unsigned*>(&pm);
        unsigned* l = reinterpret_cast<unsigned*>(&pl);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float x0 = v[8 * s + 2 * q], x1 = v[8 * s + 2 * q + 1];
            const float r0 = x0 - __uint_as_float(__float_as_uint(x0) & 0xffff0000u);
            const float r1 = x1 - __uint_as_float(__float_as_uint(x1) & 0xffff0000u);
            const float t0 = r0 - __uint_as_float(__float_as_uint(r0) & 0xffff0000u);
            const float t1 = r1 - __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
            h[q] = pvs_pack_hi16(x0, x1);
            m[q] = pvs_pack_hi16(r0, r1);
            l[q] = pvs_pack_hi16(t0, t1);
        }
        out.hi[s] = __builtin_bit_cast(bf16x8, ph);
        out.mid[s] = __builtin_bit_cast(bf16x8, pm);
        out.lo[s] = __builtin_bit_cast(bf16x8, pl);
    }
}

// ---- bf16x3 weights as ONE swizzled row-major image per part, read row-wise for W v and, through
// ds_read_b64_tr_b16 (gfx950's transposing LDS read), column-wise for W^T v: half the LDS of two
// pre-transposed copies. Element (r, c) of W [H][H] sits at 16-bit index
//   r*H + 4*((c >> 2) ^ swz(r)) + (c & 3),   swz(r) = (r / (128/H)) & (H/4 - 1)   (128/H rows span the 64 banks):
// the 8-byte chunks of a row are permuted so that 32 lanes reading the same logical chunk of 32
// consecutive rows hit 32 different bank pairs (row reads conflict-free, transposed reads 2-way).
typedef short pvs_v4s __attribute__((ext_vector_type(4)));

// H = 32 (round 5, tools/lds_conflicts.py): with swz(r) = (r / 4) & 7 the image WRITES (ds_write_b64: lane groups of 16
// contiguous lanes = 16 consecutive rows, 32 banks) were 2-way conflicts - rows r and r + 2 of a group share their banks
// and their swizzle. swz(r) = bits (r2, r3, r1 ^ r4) of the row is a bijection of (r1, r2, r3) for every r4 (the writes:
// 16 rows of one group) AND of (r2, r3, r4) for every r1 (the row reads: 32 rows, 64 banks); the transposed reads take
// four consecutive rows whole and do not care. All three conflict-free.
#ifndef PVS_IMG_PAIRED
#define PVS_IMG_PAIRED 1
#endif
template <int HB>
__device__ __forceinline__ int img_off(int r, int c) {
    constexpr int H = 32 * HB, NCH = H / 4, RPC = 128 / H;
    if constexpr (HB == 1) {
#if PVS_IMG_PAIRED
        // The two 8-byte chunks one lane's fragment is made of (channels 16 s + 4 hh + 0..3 and 16 s + 8 + 4 hh + 0..3: chunk
        // numbers q and q + 2) sit SIDE BY SIDE: a row fragment is one ds_read_b128 and a part write one ds_write_b128
        // instead of two 8-byte accesses each (and hipcc no longer pairs a row read with the other part image's, which cost
        // six v_mov per k-step to sort out). The 16-byte pairs P = 2 s + hh of a row are permuted by the two row bits
        // (r2, r1 ^ r3): conflict-free for the ds_write_b128 (8-lane groups, 32 banks), the ds_read_b128 (its own lane
        // groups, 64 banks) and the transposing reads alike (tools/lds_conflicts.py; 144 of the 992 linear 2-bit swizzles are).
        const int q = c >> 2, pair = ((q >> 2) << 1) | (q & 1), t = (q >> 1) & 1;
        const int f = ((r >> 2) & 1) | ((((r >> 1) ^ (r >> 3)) & 1) << 1);
        return r * H + 8 * (pair ^ f) + 4 * t + (c & 3);
#else
        return r * H + 4 * ((c >> 2) ^ (((r >> 2) & 3) | ((((r >> 1) ^ (r >> 4)) & 1) << 2))) + (c & 3);
#endif
    }
    return r * H + 4 * ((c >> 2) ^ ((r / RPC) & (NCH - 1))) + (c & 3);
}

// A-operand fragment (8 bf16 in the k order of the X layout) of block (bo, bi), k-step s, of one part
// image: TRANSPOSE = false: rows of W (W v); true: columns of W (W^T v) through the transposing read.
template <int HB, bool TRANSPOSE>
__device__ __forceinline__ uint4 img_fragment_bits(const unsigned short* __restrict__ part, int lane, int bo,
                                                   int bi, int s) {
    const int hh = lane >> 5;
    uint2 a, b;
    if constexpr (!TRANSPOSE) {
        const int r = 32 * bo + (lane & 31), c0 = 32 * bi + 16 * s + 4 * hh;
        if constexpr (HB == 1 && PVS_IMG_PAIRED) return *reinterpret_cast<const uint4*>(part + img_off<HB>(r, c0));
        a = *reinterpret_cast<const uint2*>(part + img_off<HB>(r, c0));
        b = *reinterpret_cast<const uint2*>(part + img_off<HB>(r, c0 + 8));
    } else {
        // 16-lane group: lane 4q+p supplies row q, columns 4p..4p+3 of a 4x16 block and receives
        // column (lane & 15) of its 4 rows
        const int li = lane & 15, q = li >> 2, p = li & 3;
        const int r0 = 32 * bi + 16 * s + 4 * hh, col = 32 * bo + 16 * ((lane >> 4) & 1) + 4 * p;
        typedef pvs_v4s __attribute__((address_space(3))) * lds_v4s;
        const pvs_v4s ta = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(part + img_off<HB>(r0 + q, col)));
        const pvs_v4s tb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(part + img_off<HB>(r0 + 8 + q, col)));
        a = __builtin_bit_cast(uint2, ta);
        b = __builtin_bit_cast(uint2, tb);
    }
    return make_uint4(a.x, a.y, b.x, b.y);
}

template <int HB, bool TRANSPOSE>
__device__ __forceinline__ bf16x8 img_fragment(const unsigned short* __restrict__ part, int lane, int bo,
                                               int bi, int s) {
    return __builtin_bit_cast(bf16x8, img_fragment_bits<HB, TRANSPOSE>(part, lane, bo, bi, s));
}

// ---- fp32 products as 3 fp16 MFMA terms ("f16x2", round 3) ------------------------------------------
// x*s = hi + lo with hi = fp16(x*s) (round to nearest) and lo = fp16(x*s - hi): 22 significant bits in two
// parts instead of 24 in three, and  a*b = (hi*hi + hi*lo + lo*hi) / (s_a s_b) + O(2^-22 |a||b|)  with the
// three terms summed in ONE fp32 accumulator of v_mfma_f32_32x32x16_f16 (same cycles as the bf16 form): half
// the MFMAs of bf16x3 and 3 VALU instructions per split value instead of 5.5 (pvs_f16_split2: scale, v_cvt_pk_f16_f32
// for a pair, v_fma_mix_f32 for each residual, v_cvt_pk again; the quarter-rate v_fma_mixlo/hi_f16 form of round 3's
// first version - 2 instructions per value - measured slower: profiles/r03_micro_valu_issue.txt).
// fp16 has 5 exponent bits, so every operand is scaled by a power of two s chosen from the maximum of the
// TILE it belongs to (activations reach 1e3, gradients 1e-9: neither fits fp16 unscaled): s * max in
// [2^13, 2^14). Then hi never overflows and the representation error of an element is
// max(2^-22 |x s|, 2^-25) - the second term (fp16 subnormal spacing) is 2^-38 of the tile maximum, so small
// elements of a tile keep more absolute accuracy than fp32 rounding of the sums they enter. Power-of-two
// scales are exact; products are divided by s_a s_b where the accumulator is consumed (folded into the FMA
// that adds the bias / multiplies by the activation derivative).
// Emulated against fp64 on random tiles with 2^+-6 dynamic range inside a tile (tools/f16x2_numerics.py):
// max error 2e-7 of sum|a||b|, between a sequential fp32 FMA chain (3e-7) and bf16x3 (1.5e-7).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct F16Parts { f16x8 hi[2], lo[2]; };

// maximum over the wave of a per-lane u32 (all 64 lanes active), wave-uniform result
__device__ __forceinline__ unsigned pvs_wave_max_u32(unsigned v) {
    v = max(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true));     // quad_perm [1,0,3,2]
    v = max(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true));     // quad_perm [2,3,0,1]
    v = max(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xf, 0xf, true));    // row_half_mirror
    v = max(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xf, 0xf, true));    // row_mirror
    const unsigned a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const unsigned c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return max(max(a, b), max(c, d));
}

// power-of-two scale for values whose largest magnitude has the fp32 bit pattern `max_bits`:
// s * max in [2^13, 2^14);  *inv = 1 / s. All-zero / denormal input: s = 2^124.
__device__ __forceinline__ float pvs_f16_scale(unsigned max_bits, float* inv) {
    int e = (int)((max_bits >> 23) & 0xffu);
    e = e < 16 ? 16 : e;
    *inv = __uint_as_float((unsigned)(e - 13) << 23);
    return __uint_as_float((unsigned)(267 - e) << 23);
}

__device__ __forceinline__ float pvs_absmax16(const float (&v)[16]) {
    float m = fmaxf(fabsf(v[0]), fabsf(v[1]));
#pragma unroll
    for (int t = 2; t < 16; t += 2) m = fmaxf(fmaxf(m, fabsf(v[t])), fabsf(v[t + 1]));
    return m;
}

// scale of one 32-edge tile operand held in X layout (16 values per lane, whole wave)
__device__ __forceinline__ float pvs_tile_scale(const float (&v)[16], float* inv) {
    return pvs_f16_scale(pvs_wave_max_u32(__float_as_uint(pvs_absmax16(v))), inv);
}

// fp16 pair (x0*s, x1*s) rounded to nearest, low half = x0
__device__ __forceinline__ unsigned pvs_f16_hi2(float x0, float x1, float s) {
    unsigned h;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(x0), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(x1), "v"(s));
    return h;
}
// fp16 pair (x0*s - hi.lo, x1*s - hi.hi): the residuals (exact in fp32) rounded to fp16
__device__ __forceinline__ unsigned pvs_f16_lo2(float x0, float x1, float s, unsigned h) {
    unsigned l;
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(x0), "v"(s), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(x1), "v"(s), "v"(h));
    return l;
}

// The same two words from conversions instead of v_fma_mixlo/hi_f16. Measured on MI355X
// (tools/micro/valu_issue_bench.hip, profiles/r03_micro_valu_issue.txt; SIMD cycles per wave-instruction at two
// or more waves per SIMD): v_fma_mixlo/hi_f16 8.4-8.9 - a quarter-rate instruction like v_exp_f32 - against
// v_mul_f32 2.5, v_cvt_pk_f16_f32 4.6, v_fma_mix_f32 4.3: 33.5 cycles per pair of values for the four mixlo/hi
// against 22.9 for  y = x s (2 x v_mul, exact: s is a power of two), h = cvt_pk(y0, y1) (round to nearest even,
// the rounding of mixlo), r = y - f32(h half) (v_fma_mix_f32: exact, the difference has at most 13 significant
// bits), l = cvt_pk(r0, r1). Bit for bit the words of pvs_f16_hi2 / pvs_f16_lo2.
template <bool PAIR = false>
__device__ __forceinline__ void pvs_f16_split2(float x0, float x1, float s, unsigned& h, unsigned& l) {
    float y0, y1;
    if constexpr (PAIR) {
        const pvs_f2 y = pvs_f2{x0, x1} * s;        // one v_pk_mul_f32
        y0 = y.x; y1 = y.y;
    } else {
        y0 = x0 * s; y1 = x1 * s;
    }
    // The two conversions are the COMPILER's (fptrunc <2 x float> selects v_cvt_pk_f16_f32 on gfx950, round to nearest
    // even like the instruction), not asm statements as until round 5: hipcc's hazard recognizer does not look inside
    // inline asm, and h / l are MFMA operands - an asm-written register that an MFMA reads with nothing but an s_waitcnt or
    // s_nop 0 in between is the "VALU write -> MFMA read" hazard unprotected, and the MFMA may take the register's old
    // content. Every shipped kernel had such places (tools/asm_mfma_hazard_scan.py) and passed because those waits
    // happened to stall; the 64-channel forward compiled with the pair arithmetic did not (profiles/r05_ab_pair_math.txt,
    // section 3; the NaN rows of the withdrawn PVS_FWD_SADDR build were the same thing). Only the residuals stay asm
    // (v_fma_mix_f32 has no pattern the compiler selects here): their consumer is the second conversion, a vector
    // instruction the compiler does guard behind an asm definition.
    typedef _Float16 pvs_h2 __attribute__((ext_vector_type(2)));
    float r0, r1;
    h = __builtin_bit_cast(unsigned, __builtin_convertvector(pvs_f2{y0, y1}, pvs_h2));
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(y0), "v"(h));
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(y1), "v"(h));
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(pvs_f2{r0, r1}, pvs_h2));
}

template <bool PAIR = false>
__device__ __forceinline__ void split_f16x2(const float (&v)[16], float s, F16Parts& out) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        uint4 ph, pl;
        unsigned* h = reinterpret_cast<unsigned*>(&ph);
        unsigned* l = reinterpret_cast<unsigned*>(&pl);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float x0 = v[8 * ks + 2 * q], x1 = v[8 * ks + 2 * q + 1];
#ifdef PVS_SPLIT_MIXLO      // (A/B only: round 3's first form. Its parts are written by asm statements, i.e. NOT guarded against the
            // VALU-write -> MFMA-read hazard, see pvs_f16_split2: never ship a build with this defined)
            h[q] = pvs_f16_hi2(x0, x1, s);
            l[q] = pvs_f16_lo2(x0, x1, s, h[q]);
#else
            pvs_f16_split2<PAIR>(x0, x1, s, h[q], l[q]);
#endif
        }
        out.hi[ks] = __builtin_bit_cast(f16x8, ph);
        out.lo[ks] = __builtin_bit_cast(f16x8, pl);
    }
}

// W [H][H] (row-major) * s as two swizzled row-major fp16 images (hi at img, lo at img + H*H), the layout of
// img_off: rows through ds_read_b64 for W v, columns through ds_read_b64_tr_b16 for W^T v.
template <int HB>
__device__ __forceinline__ void stage_weights_img_f16(unsigned short* img, const float* __restrict__ W, float s) {
    constexpr int H = 32 * HB;
    unsigned* hi = reinterpret_cast<unsigned*>(img);
    unsigned* lo = reinterpret_cast<unsigned*>(img + H * H);
    for (int i = threadIdx.x; i < H * H / 2; i += blockDim.x) {
        const int r = (2 * i) / H, c = (2 * i) % H;
        const float x0 = W[r * H + c], x1 = W[r * H + c + 1];
        const unsigned h = pvs_f16_hi2(x0, x1, s);
        const int o = img_off<HB>(r, c) >> 1;
        hi[o] = h;
        lo[o] = pvs_f16_lo2(x0, x1, s, h);
    }
}

// A-operand order for the forward kernels (one ds_read_b128 per fragment): the 32x32 block W[row0 + .][col0 + .]
// of a matrix with row stride ld, times s, as dst[((part*2 + ks)*64 + l)*4 .. +3] (uint words) = 8 fp16 of
// W[l&31][ch(8ks + j', l>>5)], part in {hi, lo}: 4 KB per block.
__device__ __forceinline__ void stage_weights_f16x2(unsigned* dst, const float* __restrict__ W, float s, int ld,
                                                    int row0, int col0) {
    W += (size_t)row0 * ld + col0;
    for (int i = threadIdx.x; i < 2 * 64 * 4; i += blockDim.x) {
        const int q = i & 3, l = (i >> 2) & 63, ks = i >> 8;
        const int o = l & 31, hh = l >> 5;
        const float x0 = W[o * ld + xch(8 * ks + 2 * q, hh)], x1 = W[o * ld + xch(8 * ks + 2 * q + 1, hh)];
        const unsigned h = pvs_f16_hi2(x0, x1, s);
        dst[((0 * 2 + ks) * 64 + l) * 4 + q] = h;
        dst[((1 * 2 + ks) * 64 + l) * 4 + q] = pvs_f16_lo2(x0, x1, s, h);
    }
}

template <int HB>
__device__ __forceinline__ void stage_weights_f16x2_blocks(unsigned* dst, const float* __restrict__ W, float s) {
#pragma unroll
    for (int bo = 0; bo < HB; ++bo)
#pragma unroll
        for (int bi = 0; bi < HB; ++bi)
            stage_weights_f16x2(dst + (bo * HB + bi) * (4 * 64 * 4), W, s, 32 * HB, 32 * bo, 32 * bi);
}

// acc[bo] += sum_bi (W s_w)[bo][bi] (v s_v)[bi]: v (X layout, fp32) is split here with the tile scale s_v
template <int HB>
__device__ __forceinline__ void mfma_chain_f16x2_blocks(const unsigned* __restrict__ Wb, int lane,
                                                        const float (&v)[HB][16], float s_v, f32x16 (&acc)[HB]) {
#pragma unroll
    for (int bi = 0; bi < HB; ++bi) {
        F16Parts b;
        split_f16x2<pvs_pair_math<HB>>(v[bi], s_v, b);
#pragma unroll
        for (int bo = 0; bo < HB; ++bo) {
            const unsigned* Wblk = Wb + (bo * HB + bi) * (4 * 64 * 4);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const f16x8 ah = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(Wblk + ((0 * 2 + ks) * 64 + lane) * 4));
                const f16x8 al = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(Wblk + ((1 * 2 + ks) * 64 + lane) * 4));
                acc[bo] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b.hi[ks], acc[bo], 0, 0, 0);
                acc[bo] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b.lo[ks], acc[bo], 0, 0, 0);
                acc[bo] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b.hi[ks], acc[bo], 0, 0, 0);
            }
        }
    }
}

// ---- lazily moving tile scales (round 4; edge_bwd_f16.hip "weight gradients accumulated IN the matrix core's
// accumulator" explains why). H = 32 backward only: in the 512-register kernels (H = 64, the wide team backward) the
// weight-gradient accumulators live in the accumulator half of the register file, and the rescaling branch - vector
// instructions that touch them inside the tile loop - makes the allocator move them out of it: the wide backward went
// from 25-163 to 145-237 spilled VGPRs with this scheme (the same wall as profiles/r03_h64_backward_f16x2_rejected.txt) ----
#ifndef PVS_LAZY_WSCALE
#define PVS_LAZY_WSCALE 1
#endif
constexpr int kLazyWindow = 2;
struct LazyExp { int e; };                 // exponent of the scale's ceiling (scale = 2^(140 - e)); < 0: not set yet

// (max_bits: the fp32 bit pattern of the tile's - or the team's - largest magnitude, wave-uniform)
__device__ __forceinline__ float pvs_lazy_scale_from_max(unsigned max_bits, LazyExp& st, float* inv) {
    int e_t = (int)((max_bits >> 23) & 0xffu);
    e_t = e_t < 16 ? 16 : e_t;
    if (st.e < 0 || e_t > st.e || st.e - e_t > kLazyWindow) st.e = e_t + 1 > 254 ? 254 : e_t + 1;
    *inv = __uint_as_float((unsigned)(st.e - 13) << 23);
    return __uint_as_float((unsigned)(267 - st.e) << 23);
}
// The scale stays while  e_t <= st.e  and  st.e - e_t <= kLazyWindow  (e_t = the wave maximum's exponent, at least 16):
// that is "every lane's maximum is below 2^(st.e - 126) and some lane's is at least 2^(st.e - kLazyWindow - 127)" - two
// compares of the lanes' OWN maxima against wave-uniform bit patterns (round 5). The wave-wide maximum (four DPP steps,
// four v_readlane and the scalar work behind them) is only formed on the tiles where the scale moves; same decisions, same
// scales, bit for bit.
__device__ __forceinline__ float pvs_lazy_tile_scale(const float (&v)[16], LazyExp& st, float* inv) {
    const unsigned m = __float_as_uint(pvs_absmax16(v));
    const unsigned hi = (unsigned)(st.e + 1) << 23;
    const unsigned lo = st.e - kLazyWindow <= 16 ? 0u : (unsigned)(st.e - kLazyWindow) << 23;
#ifndef PVS_WINDOW_CHECK
#define PVS_WINDOW_CHECK 1
#endif
    if (!PVS_WINDOW_CHECK || st.e < 0 || __ballot(m >= hi) != 0ull || __ballot(m >= lo) == 0ull)
        return pvs_lazy_scale_from_max(pvs_wave_max_u32(m), st, inv);
    *inv = __uint_as_float((unsigned)(st.e - 13) << 23);
    return __uint_as_float((unsigned)(267 - st.e) << 23);
}

// What an accumulator that lives in MFMA registers carries. Its content is  true value x 2^(-units)  up to a fixed
// offset, units = the sum of the exponents of its operand images' scale ceilings (LazyExp.e; the bias column: of the
// gradient image alone). `cur` = the units it is in now; `top` = the COARSEST units (largest magnitudes) it has been in
// since it (re)started; cur < 0: nothing accumulated yet. Wave-uniform, in scalar registers.
// (one scalar register: cur in the low half, top in the high half - the H = 32 backward has no scalar register to spare)
struct AccUnits {
    int v;
    __device__ __forceinline__ int cur() const { return v < 0 ? -1 : (v & 0xffff); }
    __device__ __forceinline__ int top() const { return v >> 16; }
    __device__ __forceinline__ void set(int cur_, int top_) { v = cur_ | (top_ << 16); }
};
constexpr int kAccSpan = 60;

// Decide what happens to an accumulator when the next tile's product comes in `units`:
//   0  skip the tile: it lies more than 2^kAccSpan below the coarsest tile accumulated so far - below the fp32
//      resolution of the sum (a tile's largest operand entries fill [2^11, 2^14) of their image, so its product sums
//      are comparable to the accumulator's content in the same units);
//   1  accumulate, after multiplying the accumulator by *factor (an exact power of two, 2^-120 ... 2^120; 1 when the
//      units did not move; 0 = restart: the tile lies more than 2^kAccSpan ABOVE everything accumulated so far).
// The window is measured from `top`, not from `cur` (round 5): measured from the current units, a run of tiles that
// each step DOWN by less than the span multiplied the content by the product of all the steps (2^150 over three steps
// of 2^-50: infinity), and a restart could drop content that an earlier, coarser tile had left. From `top` the content
// is never scaled up by more than 2^kAccSpan in total and a restart drops only what is 2^-60 of the new tile.
__device__ __forceinline__ int pvs_acc_step(AccUnits& st, int units, float* factor) {
    *factor = 1.f;
    if (st.v < 0) { st.set(units, units); return 1; }
    const int cur = st.cur(), top = st.top();
    if (units == cur) return 1;
    if (top - units > kAccSpan) return 0;
    if (units - top > kAccSpan) { st.set(units, units); *factor = 0.f; return 1; }
    *factor = __uint_as_float((unsigned)(127 + cur - units) << 23);
    st.set(units, units > top ? units : top);
    return 1;
}

// The weight-gradient accumulator(s) `a` (operands G and Act) and the bias column that rides in lanes `col_lane` of gB
// (operand G alone) are brought to the scales (eg, ea) of the tile's images SEPARATELY (one shared decision
// dropped a tile's bias sums whenever its ACTIVATION image alone was far below the accumulator's - an all-zero m tile
// beside ordinary ones - and could skip where the bias column had to restart).
// Returns bit 0: issue the weight-gradient MFMAs, bit 1: issue the bias-column MFMAs.
template <int NB>
__device__ __forceinline__ int pvs_rescale_acc(f32x16 (&a)[NB], f32x16& gB, bool col_lane, AccUnits& uw, AccUnits& ub,
                                               int eg, int ea) {
    if (uw.cur() == eg + ea && ub.cur() == eg) return 3;      // (the common case: no scale moved)
    float fw, fb;
    const int do_w = pvs_acc_step(uw, eg + ea, &fw), do_b = pvs_acc_step(ub, eg, &fb);
    if (do_w && fw != 1.f) {
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) a[b][r] *= fw;
    }
    if (do_b && fb != 1.f) {
#pragma unroll
        for (int r = 0; r < 16; ++r) gB[r] *= col_lane ? fb : 1.f;
    }
    return do_w | (do_b << 1);
}
__device__ __forceinline__ int pvs_rescale_acc(f32x16& a, f32x16& gB, bool col_lane, AccUnits& uw, AccUnits& ub, int eg,
                                               int ea) {
    return pvs_rescale_acc<1>(reinterpret_cast<f32x16 (&)[1]>(a), gB, col_lane, uw, ub, eg, ea);
}


// tile scale of an H = 32*HB channel operand (all blocks share one scale)
template <int HB>
__device__ __forceinline__ float pvs_tile_scale_blocks(const float (&v)[HB][16], float* inv) {
    float m = pvs_absmax16(v[0]);
#pragma unroll
    for (int b = 1; b < HB; ++b) m = fmaxf(m, pvs_absmax16(v[b]));
    return pvs_f16_scale(pvs_wave_max_u32(__float_as_uint(m)), inv);
}

// PER-EDGE scale of an operand of a CHAIN product (round 4). Z = W V has one column per edge, so a scale may differ
// from edge to edge: column e of (W s_w)(V s_e) is s_w s_e Z[:, e], and in the X / D layouts a lane holds exactly one
// edge's column - the accumulator is descaled with the lane's own 1 / s_e. Every edge then keeps 22 bits relative to
// ITS OWN largest channel, however large its tile-mates are (with one scale per tile an edge 2^-20 of a tile-mate kept
// ~18 bits), and the scale costs less than the wave-wide maximum: the two halves of the wave hold the two channel
// halves of the same 32 edges, so one v_permlane32_swap pairs them up. (The weight-gradient products sum over the
// EDGE index and need one scale per tile: the backward kernels keep pvs_tile_scale.)
__device__ __forceinline__ unsigned pvs_pair_halves_max_u32(unsigned v) {
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);   // r[0]: lanes j of both halves, r[1]: lanes 32 + j
    return max(r[0], r[1]);
}

template <int HB>
__device__ __forceinline__ float pvs_edge_scale_blocks(const float (&v)[HB][16], float* inv) {
    float m = pvs_absmax16(v[0]);
#pragma unroll
    for (int b = 1; b < HB; ++b) m = fmaxf(m, pvs_absmax16(v[b]));
    return pvs_f16_scale(pvs_pair_halves_max_u32(__float_as_uint(m)), inv);
}

// largest |W[i]| of an n-element array as fp32 bits, over the whole workgroup (slot: one LDS word, zeroed
// by the caller before a barrier; call from every thread, read *slot after the next barrier)
__device__ __forceinline__ void pvs_block_absmax(const float* __restrict__ W, int n, unsigned* slot) {
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) m = fmaxf(m, fabsf(W[i]));
    const unsigned wm = pvs_wave_max_u32(__float_as_uint(m));
    if ((threadIdx.x & 63) == 0) atomicMax(slot, wm);
}

template <typename K>
int set_lds(K kernel, size_t lds) {
    if (lds > 48 * 1024)
        PVS_CHECK_HIP(hipFuncSetAttribute((const void*)kernel,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return 0;
}

// Edges a wave gets before the grid grows by another workgroup. 512 until round 5 (sixteen tiles amortise a workgroup's
// weight staging): right for BASELINE-size batches, whose grids are capped by the CU count anyway, and wrong for small ones
// - at the reference's default shape (32 graphs of 500 atoms, r = 4 A: 176k edges) the backward ran on 43 of 256 CUs.
// Two tiles per wave: edge forward 0.62 -> 0.20 ms, edge backward 0.81 -> 0.28 ms per 6-layer step there (32: 0.19 / 0.28;
// profiles/r05_ab_small_batch_grid.txt). PVS_EDGES_PER_WAVE overrides it (A/B).
inline int pvs_edges_per_wave() {
    static const int v = [] { const char* e = getenv("PVS_EDGES_PER_WAVE"); const int x = e ? atoi(e) : 0; return x > 0 ? x : 64; }();
    return v;
}

// Edges per chunk above which a wave's share is cut into several chunks (PVS_CHUNK_EDGES overrides it: A/B only).
// Chunk ends are row-aligned, so a wave's share is uneven by up to a row per chunk end (157 edges at cfg2) and the launch
// waits for the largest share: FEWER, larger chunks per wave balance better (round 6, H = 32 backward at cfg2: two chunks
// of 2.5 k edges per wave -> one of 5 k: -2.5 % per launch; perfectly equal shares - timing-only, -DPVS_ABL_UNALIGNED_CHUNKS
// - would give -3.3 %: profiles/r06_ab_chunk_balance.txt).
inline long long pvs_chunk_edges(long long dflt = 4096) {
    static const long long v = [] { const char* e = getenv("PVS_CHUNK_EDGES"); return e ? atoll(e) : 0ll; }();
    return v > 0 ? v : dflt;
}

void pick_grid(int E, int* blocks, int* n_chunks, int nw = kWaves, int max_blocks = 1024) {
    // fill the chip first: a wave gets >= pvs_edges_per_wave() edges where the range allows; chunks of <= ~4096 edges; every wave gets the same number of chunks
    const long long per = pvs_edges_per_wave();
    long long b = ((long long)E + (long long)nw * per - 1) / ((long long)nw * per);
    if (b < 1) b = 1;
    if (b > max_blocks) b = max_blocks;
    const long long waves = b * nw;
    const long long ce = pvs_chunk_edges();
    long long per_wave = ((long long)E + waves * ce - 1) / (waves * ce);
    if (per_wave < 1) per_wave = 1;
    *blocks = (int)b;
    *n_chunks = (int)(waves * per_wave);
}


}  // namespace
