// TIMING-ONLY PROBE, never in the shipped library (built by tools/variant_obj.sh with -DPVS_PAIR_PROBE; VERDICT r05 item 1,
// profiles/r06_ab_channel_split_pair_rejected.txt): the H = 32 edge backward with a tile's CHANNELS split across a wave
// pair. Wave p of a pair owns output channels [16 p, 16 p + 16) of every tensor of the pair's 32-edge tile (8 values per
// lane and tensor instead of 16; 20 kernel-lifetime weight-gradient accumulator registers instead of 48), the pair
// shares ONE set of fp16 images (a1, m, gradient: 4 KB each), every product is v_mfma_f32_16x16x32_f16 (K = all 32
// channels for the chain products - the partner's half comes from the shared image - and K = the tile's 32 edges for the
// weight gradients), workgroups of 4 waves = 2 pairs so that s_barrier is the pair synchronisation, three workgroups per
// CU = three waves per SIMD (49.7 KB of LDS per workgroup).
//
// What the probe keeps faithful: the instruction stream of one tile per wave - gathers of the wave's 4-channel quads
// for its two edge slots, the per-edge scalars (which BOTH waves of a pair need, i.e. are computed twice), three
// SiLU + SiLU' blocks on 8 values, four scale checks and splits, image writes, the barriers a shared image needs, the
// B / A fragment reads, 40 MFMAs, the two cross-wave channel sums (coordinate scalar, g_rho) through LDS, the edge-major
// tile, row sums and the streamed stores of g_z1 (64 of a row's 128 bytes per wave) and the 16-byte records.
// What it does NOT do: produce right numbers (fragment lane maps, the lazy scales' bookkeeping between the two waves,
// partial tiles at graph boundaries and the slab reduction are not worked out - the time they would add is not in the
// probe, which therefore flatters the design).
#pragma once

namespace pairprobe {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kH = 32;
constexpr int kImg = 32 * 32;                      // shorts of one fp16 part image [32 rows][32 columns]
constexpr int kPairBytes = 3 * 2 * kImg * 2 + 2 * 2 * 32 * 4;          // a1, m, gradient images (hi + lo) + two exchange rows per wave
constexpr int kWaveBytes = 32 * 20 * 4 + 32 * 16 + 32 * 4;             // g_z1 tile [32][16 + 4], records, row ids
constexpr int kSharedBytes = 2 * 2 * kImg * 2 + (5 + PVS_MAX_EDGE_ATTR) * kH * 4 + 2 * 64 * 16;
constexpr int kLds = kSharedBytes + 2 * kPairBytes + 4 * kWaveBytes;

// [row][column] fp16 image, 16-byte chunks of a row permuted by two row bits (row fragments of 16 rows: conflict-free)
__device__ __forceinline__ int ioff(int r, int c) { return r * 32 + 8 * ((c >> 3) ^ ((r >> 2) & 3)) + (c & 7); }

__device__ __forceinline__ f16x8 frag_row(const unsigned short* img, int r, int kg) {
    return __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(img + ioff(r, 8 * kg)));
}
// 8 consecutive ROWS of one column (k = the row index): two transposing reads
__device__ __forceinline__ f16x8 frag_col(const unsigned short* img, int lane, int col0, int r0) {
    const int li = lane & 15, q = li >> 2, p4 = li & 3;
    typedef pvs_v4s __attribute__((address_space(3))) * lds_v4s;
    const pvs_v4s ta = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(img + ioff(r0 + q, col0 + 4 * p4)));
    const pvs_v4s tb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(img + ioff(r0 + 4 + q, col0 + 4 * p4)));
    const uint2 a = __builtin_bit_cast(uint2, ta), b = __builtin_bit_cast(uint2, tb);
    return __builtin_bit_cast(f16x8, make_uint4(a.x, a.y, b.x, b.y));
}

struct Parts8 { unsigned h[4], l[4]; };           // 8 values = two edge slots x four channels: hi / lo words

__device__ __forceinline__ void split8(const float (&v)[8], float s, Parts8& o) {
#pragma unroll
    for (int q = 0; q < 4; ++q) pvs_f16_split2<true>(v[2 * q], v[2 * q + 1], s, o.h[q], o.l[q]);
}
// the lane's two 8-byte chunks (edge slots na and 16 + na, channels c0 .. c0 + 3) of both parts
__device__ __forceinline__ void write8(unsigned short* img, int na, int c0, const Parts8& o) {
    *reinterpret_cast<uint2*>(img + ioff(na, c0)) = make_uint2(o.h[0], o.h[1]);
    *reinterpret_cast<uint2*>(img + ioff(16 + na, c0)) = make_uint2(o.h[2], o.h[3]);
    *reinterpret_cast<uint2*>(img + kImg + ioff(na, c0)) = make_uint2(o.l[0], o.l[1]);
    *reinterpret_cast<uint2*>(img + kImg + ioff(16 + na, c0)) = make_uint2(o.l[2], o.l[3]);
}
// Z[16 own channels][32 edges] = W[own rows][all 32 channels] V: A = weight rows (hi, lo), B = the shared image
template <bool TRANSPOSE>
__device__ __forceinline__ void chain16(const unsigned short* wimg, const unsigned short* vimg, int lane, int p,
                                        f32x4 (&acc)[2]) {
    const int n = lane & 15, kg = lane >> 4;
    // (W^T v: the weight's columns through the transposing read of the same image, as the shipped kernel does)
    const f16x8 ah = TRANSPOSE ? frag_col(wimg, lane, 16 * p, 8 * kg) : frag_row(wimg, 16 * p + n, kg);
    const f16x8 al = TRANSPOSE ? frag_col(wimg + kImg, lane, 16 * p, 8 * kg) : frag_row(wimg + kImg, 16 * p + n, kg);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const f16x8 bh = frag_row(vimg, 16 * b + n, kg), bl = frag_row(vimg + kImg, 16 * b + n, kg);
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c, 0, 0, 0);
        acc[b] = c;
    }
}
// gW[16 own gradient channels][32 activation channels] += G^T Act over the tile's 32 edges; gB += G^T ones
__device__ __forceinline__ void wgrad16(const unsigned short* gimg, const unsigned short* aimg, const unsigned* ones,
                                        int lane, int p, f32x4 (&gW)[2], f32x4& gB) {
    const int kg = lane >> 4;
    const f16x8 gh = frag_col(gimg, lane, 16 * p, 8 * kg), gl = frag_col(gimg + kImg, lane, 16 * p, 8 * kg);
    const f16x8 one = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(ones + lane * 4));
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const f16x8 ah = frag_col(aimg, lane, 16 * b, 8 * kg), al = frag_col(aimg + kImg, lane, 16 * b, 8 * kg);
        gW[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gl, ah, gW[b], 0, 0, 0);
        gW[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gh, al, gW[b], 0, 0, 0);
        gW[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gh, ah, gW[b], 0, 0, 0);
    }
    gB = __builtin_amdgcn_mfma_f32_16x16x32_f16(gl, one, gB, 0, 0, 0);
    gB = __builtin_amdgcn_mfma_f32_16x16x32_f16(gh, one, gB, 0, 0, 0);
}

__device__ __forceinline__ float absmax8(const float (&v)[8]) {
    float m = fmaxf(fabsf(v[0]), fabsf(v[1]));
#pragma unroll
    for (int t = 2; t < 8; t += 2) m = fmaxf(fmaxf(m, fabsf(v[t])), fabsf(v[t + 1]));
    return m;
}
// the wave's own lazy decision (the probe does not reconcile it with the partner's)
__device__ __forceinline__ float lazy8(const float (&v)[8], int& e, float* inv) {
    LazyExp st{e ? e : -1};
    const unsigned m = __float_as_uint(absmax8(v));
    const unsigned hi = (unsigned)(st.e + 1) << 23;
    const unsigned lo = st.e - kLazyWindow <= 16 ? 0u : (unsigned)(st.e - kLazyWindow) << 23;
    float sc;
    if (st.e < 0 || __ballot(m >= hi) != 0ull || __ballot(m >= lo) == 0ull) {
        sc = pvs_lazy_scale_from_max(pvs_wave_max_u32(m), st, inv);
    } else {
        *inv = __uint_as_float((unsigned)(st.e - 13) << 23);
        sc = __uint_as_float((unsigned)(267 - st.e) << 23);
    }
    e = st.e;
    return sc;
}
// sum over the four channel groups of a lane's edge slot (lanes n, n + 16, n + 32, n + 48)
__device__ __forceinline__ float groups_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

#ifndef PVS_PAIR_PROBE_OCC
#define PVS_PAIR_PROBE_OCC 3          // workgroups of 4 waves per CU = waves per SIMD
#endif
__global__ void __launch_bounds__(256, PVS_PAIR_PROBE_OCC)
k_edge_bwd_pair_probe(PvsGraph g, PvsEdgeW w, uint32_t flags, PvsEdgeBwdIO io, int n_chunks, int e_lo, int e_hi) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned short* W2i = reinterpret_cast<unsigned short*>(smem);          // W2, Wc1: hi + lo each
    unsigned short* Wc1i = W2i + 2 * kImg;
    float* b2t = reinterpret_cast<float*>(Wc1i + 2 * kImg);
    float* bc1t = b2t + kH;
    float* wc2t = bc1t + kH;
    float* wrhot = wc2t + kH;
    float* spare = wrhot + kH;
    float* attrt = spare + kH;
    unsigned* ones0 = reinterpret_cast<unsigned*>(attrt + PVS_MAX_EDGE_ATTR * kH);
    unsigned* ones1 = ones0 + 64 * 4;
    char* pair_base = reinterpret_cast<char*>(ones1 + 64 * 4);
    const bool upd = (flags & PVS_UPDATE_COORDS) && io.gxagg != nullptr;

    for (int i = threadIdx.x; i < kH * kH; i += 256) {        // (unscaled staging: timing only)
        const int r = i >> 5, c = i & 31;
        const float a = w.w2[i] * 1024.f, b = upd ? w.wc1[i] * 1024.f : 0.f;
        unsigned h, l;
        pvs_f16_split2<false>(a, b, 1.f, h, l);
        W2i[ioff(r, c)] = (unsigned short)(h & 0xffffu); W2i[kImg + ioff(r, c)] = (unsigned short)(l & 0xffffu);
        Wc1i[ioff(r, c)] = (unsigned short)(h >> 16); Wc1i[kImg + ioff(r, c)] = (unsigned short)(l >> 16);
    }
    for (int c = threadIdx.x; c < kH; c += 256) {
        b2t[c] = w.b2[c];
        bc1t[c] = upd ? w.bc1[c] : 0.f;
        wc2t[c] = upd ? w.wc2[c] : 0.f;
        wrhot[c] = w.w1[c * w.ld1 + w.off_rho];
        for (int t = 0; t < PVS_MAX_EDGE_ATTR; ++t)
            attrt[t * kH + c] = t < w.n_attr ? w.w1[c * w.ld1 + w.off_rho + 1 + t] : 0.f;
    }
    for (int i = threadIdx.x; i < 64 * 4; i += 256) {
        const int col = (i >> 2) & 15;
        ones0[i] = col == 0 ? 0x3c003c00u : 0u;
        ones1[i] = col == 1 ? 0x3c003c00u : 0u;
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, pair = wv >> 1, p = wv & 1;
    const int na = lane & 15, cg = lane >> 4, c0 = 16 * p + 4 * cg;
    unsigned short* A1I = reinterpret_cast<unsigned short*>(pair_base + pair * kPairBytes);
    unsigned short* MI = A1I + 2 * kImg;
    unsigned short* GI = MI + 2 * kImg;
    float* xch = reinterpret_cast<float*>(GI + 2 * kImg);                 // [2 exchanges][2 waves][32 edges]
    char* wave_base = pair_base + 2 * kPairBytes + wv * kWaveBytes;
    float* T1 = reinterpret_cast<float*>(wave_base);                       // [32 edges][16 + 4]
    float* tx = T1 + 32 * 20;
    int* rowbuf = reinterpret_cast<int*>(tx + 32 * 4);

    f32x4 gW2[2], gWc1[2], gB = {0.f, 0.f, 0.f, 0.f};
    float g_wc2x[8];
#pragma unroll
    for (int b = 0; b < 2; ++b) { gW2[b] = gB; gWc1[b] = gB; }
#pragma unroll
    for (int r = 0; r < 8; ++r) g_wc2x[r] = 0.f;
    int e_a1 = 0, e_m = 0, e_g = 0, e_g2 = 0;
    const float inv_sw = 1.f / 1024.f;

    const int total_pairs = gridDim.x * 2;
    for (int chunk = pvs_xcd_block(blockIdx.x, gridDim.x) * 2 + pair; chunk < n_chunks; chunk += total_pairs) {
        const int e_begin = __builtin_amdgcn_readfirstlane(chunk_begin(g, chunk, n_chunks, e_lo, e_hi));
        const int e_end = __builtin_amdgcn_readfirstlane(chunk_begin(g, chunk + 1, n_chunks, e_lo, e_hi));
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        float accx0 = 0.f, accx1 = 0.f, accx2 = 0.f;
        int cur_row = -1;
        // NOTE both waves of a pair (and, through s_barrier, both pairs of a workgroup) walk their chunks tile by tile
        // together: the trip count is made workgroup-uniform below (a real kernel would need the same)
        const int my_tiles = (e_end - e_begin + 31) / 32;
        __shared__ int max_tiles[2];
        if (lane == 0 && p == 0) max_tiles[pair] = my_tiles;
        __syncthreads();
        const int n_tiles = max(max_tiles[0], max_tiles[1]);
        __syncthreads();
        for (int t = 0; t < n_tiles; ++t) {
            const int e0 = min(e_begin + 32 * t, max(e_end - 1, 0));
            const int t_end = min(e0 + 32, e_end);
            int ee[2], ri[2], cj[2], ty[2];
            bool valid[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int e = e0 + 16 * b + na;
                valid[b] = e < t_end;
                ee[b] = min(max(valid[b] ? e : t_end - 1, 0), g.n_edges - 1);
                ri[b] = g.row[ee[b]];
                cj[b] = g.col[ee[b]];
                ty[b] = w.n_attr ? (int)g.etype[ee[b]] : 0;
            }
            const int prev_row = (ee[0] == e_begin || ee[0] == 0) ? -1 : g.row[ee[0] - 1];
            const unsigned bmask = (unsigned)__ballot(valid[0] && cg == 0 && ri[0] != prev_row) |
                                   ((unsigned)__ballot(valid[1] && cg == 0 && ri[1] != ri[0]) << 16);
            // ---- gather: the wave's channel quad of P_i and Q_j, coordinates, for both edge slots ----
            float z1[8], d[2][3], rho[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const float4 P = *reinterpret_cast<const float4*>(io.PQ + (size_t)ri[b] * 2 * kH + c0);
                const float4 Q = *reinterpret_cast<const float4*>(io.PQ + (size_t)cj[b] * 2 * kH + kH + c0);
                d[b][0] = io.x[3 * ri[b]] - io.x[3 * cj[b]];
                d[b][1] = io.x[3 * ri[b] + 1] - io.x[3 * cj[b] + 1];
                d[b][2] = io.x[3 * ri[b] + 2] - io.x[3 * cj[b] + 2];
                rho[b] = d[b][0] * d[b][0] + d[b][1] * d[b][1] + d[b][2] * d[b][2];
                const float4 wr = *reinterpret_cast<const float4*>(wrhot + c0);
                const float4 at = *reinterpret_cast<const float4*>(attrt + ty[b] * kH + c0);
                z1[4 * b] = fmaf(wr.x, rho[b], P.x + Q.x) + at.x;
                z1[4 * b + 1] = fmaf(wr.y, rho[b], P.y + Q.y) + at.y;
                z1[4 * b + 2] = fmaf(wr.z, rho[b], P.z + Q.z) + at.z;
                z1[4 * b + 3] = fmaf(wr.w, rho[b], P.w + Q.w) + at.w;
            }
            float a1[8], d1[8];
#pragma unroll
            for (int r = 0; r < 8; r += 2) {
                pvs_f2 av, dv;
                pvs_silu_grad2(pvs_f2{z1[r], z1[r + 1]}, av, dv);
                a1[r] = av.x; a1[r + 1] = av.y; d1[r] = dv.x; d1[r + 1] = dv.y;
            }
            Parts8 pb;
            float inv_sa1, inv_sm = 1.f;
            const float sa1 = lazy8(a1, e_a1, &inv_sa1);
            split8(a1, sa1, pb);
            write8(A1I, na, c0, pb);
            __syncthreads();                                   // (1) the pair's a1 image is complete
            f32x4 acc2[2];
            chain16<false>(W2i, A1I, lane, p, acc2);
            float z2[8], m[8], dz2[8];
            {
                const float4 bias = *reinterpret_cast<const float4*>(b2t + c0);
                const float k2 = inv_sa1 * inv_sw;
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    z2[4 * b] = fmaf(acc2[b][0], k2, bias.x); z2[4 * b + 1] = fmaf(acc2[b][1], k2, bias.y);
                    z2[4 * b + 2] = fmaf(acc2[b][2], k2, bias.z); z2[4 * b + 3] = fmaf(acc2[b][3], k2, bias.w);
                }
            }
#pragma unroll
            for (int r = 0; r < 8; r += 2) {
                pvs_f2 mv, dv;
                pvs_silu_grad2(pvs_f2{z2[r], z2[r + 1]}, mv, dv);
                m[r] = mv.x; m[r + 1] = mv.y; dz2[r] = dv.x; dz2[r + 1] = dv.y;
            }
            float gm[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) gm[r] = 0.f;
            float s_coord[2] = {0.f, 0.f}, nrm[2] = {1.f, 1.f}, gT[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
            if (upd) {
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    gT[b][0] = io.gxagg[3 * ri[b]]; gT[b][1] = io.gxagg[3 * ri[b] + 1]; gT[b][2] = io.gxagg[3 * ri[b] + 2];
                }
                const float sm = lazy8(m, e_m, &inv_sm);
                split8(m, sm, pb);
                write8(MI, na, c0, pb);
                __syncthreads();                               // (2) the m image
                f32x4 accc[2];
                chain16<false>(Wc1i, MI, lane, p, accc);
                const float4 bias2 = *reinterpret_cast<const float4*>(bc1t + c0);
                const float4 wc2 = *reinterpret_cast<const float4*>(wc2t + c0);
                const float wc2v[4] = {wc2.x, wc2.y, wc2.z, wc2.w}, b2v[4] = {bias2.x, bias2.y, bias2.z, bias2.w};
                const float kc = inv_sm * inv_sw;
                float q[8], dq[8], sp[2] = {0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 8; r += 2) {
                    const int b = r >> 2;
                    const pvs_f2 zc = pvs_fma2(pvs_f2{accc[b][r & 3], accc[b][(r & 3) + 1]}, pvs_f2{kc, kc},
                                               pvs_f2{b2v[r & 3], b2v[(r & 3) + 1]});
                    pvs_f2 qv, dv;
                    pvs_silu_grad2(zc, qv, dv);
                    q[r] = qv.x; q[r + 1] = qv.y; dq[r] = dv.x; dq[r + 1] = dv.y;
                    sp[b] = fmaf(wc2v[r & 3], q[r], sp[b]);
                    sp[b] = fmaf(wc2v[(r & 3) + 1], q[r + 1], sp[b]);
                }
                // the coordinate scalar sums over ALL 32 channels: over the lane's four groups, then the partner's half
                sp[0] = groups_sum(sp[0]);
                sp[1] = groups_sum(sp[1]);
                if (cg == 0) { xch[p * 32 + na] = sp[0]; xch[p * 32 + 16 + na] = sp[1]; }
                __syncthreads();                               // (3) both halves of the channel sum
                float g_s[2];
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    float s = sp[b] + xch[(p ^ 1) * 32 + 16 * b + na];
                    float dact = 1.f;
                    if (flags & PVS_TANH) { s = pvs_tanh(s); dact = 1.f - s * s; }
                    if (flags & PVS_NORMALIZE) nrm[b] = 1.f / (sqrtf(rho[b]) + 1e-8f);
                    s_coord[b] = s;
                    g_s[b] = (d[b][0] * gT[b][0] + d[b][1] * gT[b][1] + d[b][2] * gT[b][2]) * nrm[b] * dact * (valid[b] ? 1.f : 0.f);
                }
                float g_zc[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    g_zc[r] = g_s[r >> 2] * wc2v[r & 3] * dq[r];
                    g_wc2x[r] = fmaf(g_s[r >> 2], q[r], g_wc2x[r]);
                }
                float inv_sg;
                const float sg_ = lazy8(g_zc, e_g, &inv_sg);
                split8(g_zc, sg_, pb);
                write8(GI, na, c0, pb);
                __syncthreads();                               // (4) the g_zc image
                f32x4 accg[2];
                chain16<true>(Wc1i, GI, lane, p, accg);
                const float kg = inv_sg * inv_sw;
#pragma unroll
                for (int r = 0; r < 8; ++r) gm[r] = accg[r >> 2][r & 3] * kg;
                wgrad16(GI, MI, ones0, lane, p, gWc1, gB);
            }
            {   // row terms: the wave's quad of g_M[row] for both edge slots
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const float4 gM = *reinterpret_cast<const float4*>(io.gM + (size_t)ri[b] * kH + c0);
                    const float vm = valid[b] ? 1.f : 0.f;
                    gm[4 * b] = fmaf(vm, gM.x, gm[4 * b]); gm[4 * b + 1] = fmaf(vm, gM.y, gm[4 * b + 1]);
                    gm[4 * b + 2] = fmaf(vm, gM.z, gm[4 * b + 2]); gm[4 * b + 3] = fmaf(vm, gM.w, gm[4 * b + 3]);
                }
            }
            float g_z2[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) g_z2[r] = gm[r] * dz2[r];
            float inv_sg2;
            const float sg2 = lazy8(g_z2, e_g2, &inv_sg2);
            split8(g_z2, sg2, pb);
            __syncthreads();                                   // (5) the partner has read the g_zc image
            write8(GI, na, c0, pb);
            __syncthreads();                                   // (6) the g_z2 image
            f32x4 ga1[2];
            chain16<true>(W2i, GI, lane, p, ga1);
            wgrad16(GI, A1I, ones1, lane, p, gW2, gB);
            float g_z1[8], gr[2] = {0.f, 0.f};
            const float k1g = inv_sg2 * inv_sw;
            {
                const float4 wr = *reinterpret_cast<const float4*>(wrhot + c0);
                const float wrv[4] = {wr.x, wr.y, wr.z, wr.w};
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    g_z1[r] = ga1[r >> 2][r & 3] * (d1[r] * k1g);
                    gr[r >> 2] = fmaf(wrv[r & 3], g_z1[r], gr[r >> 2]);
                }
            }
            gr[0] = groups_sum(gr[0]);
            gr[1] = groups_sum(gr[1]);
            if (cg == 0) { xch[64 + p * 32 + na] = gr[0]; xch[64 + p * 32 + 16 + na] = gr[1]; }
            __syncthreads();                                   // (7) both halves of g_rho
            // per-edge records: both waves form all 32 (the row sums of g_x and the row ids are per wave), wave p streams
            // the records of edge slot block p
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const float g_rho = gr[b] + xch[64 + (p ^ 1) * 32 + 16 * b + na];
                const float k1 = s_coord[b] * nrm[b] * (valid[b] ? 1.f : 0.f);
                const float gd0 = fmaf(k1, gT[b][0], 2.f * d[b][0] * g_rho);
                const float gd1 = fmaf(k1, gT[b][1], 2.f * d[b][1] * g_rho);
                const float gd2 = fmaf(k1, gT[b][2], 2.f * d[b][2] * g_rho);
                if (cg == 0) {
                    *reinterpret_cast<float4*>(tx + (16 * b + na) * 4) = make_float4(gd0, gd1, gd2, 0.f);
                    rowbuf[16 * b + na] = ri[b];
                    if (b == p && valid[b])
                        pvs_store_nt(io.gd + (size_t)(e0 + 16 * b + na) * 4, make_float4(gd0, gd1, gd2, pvs_pack_rho_type(rho[b], ty[b])));
                }
            }
            // ---- g_z1 of the wave's 16 channels edge-major; 64 of each row's 128 bytes to HBM + the row-side sums ----
            *reinterpret_cast<float4*>(T1 + na * 20 + 4 * cg) = make_float4(g_z1[0], g_z1[1], g_z1[2], g_z1[3]);
            *reinterpret_cast<float4*>(T1 + (16 + na) * 20 + 4 * cg) = make_float4(g_z1[4], g_z1[5], g_z1[6], g_z1[7]);
            pvs_wave_lds_sync();
            {
                const int rs = lane >> 2, qd = lane & 3;
#pragma unroll
                for (int pass = 0; pass < 2; ++pass) {
                    const int rl = 16 * pass + rs;
                    const float4 v = *reinterpret_cast<const float4*>(T1 + rl * 20 + 4 * qd);
                    if (e0 + rl < t_end) pvs_store_nt(io.gz1 + (size_t)(e0 + rl) * kH + 16 * p + 4 * qd, v);
                    const bool starts = (bmask >> rl) & 1u;
                    if (__ballot(starts) != 0ull) {            // a row starts inside these 16 slots: close the open row
                        float4 tot = acc;
#pragma unroll
                        for (int o = 4; o < 64; o <<= 1) {
                            tot.x += __shfl_xor(tot.x, o, 64); tot.y += __shfl_xor(tot.y, o, 64);
                            tot.z += __shfl_xor(tot.z, o, 64); tot.w += __shfl_xor(tot.w, o, 64);
                        }
                        if (cur_row >= 0 && rs == 0)
                            *reinterpret_cast<float4*>(io.gPQ + (size_t)cur_row * 2 * kH + 16 * p + 4 * qd) = tot;
                        if (cur_row >= 0 && lane == 0 && p == 0) {
                            io.gx_row[3 * cur_row] = accx0; io.gx_row[3 * cur_row + 1] = accx1; io.gx_row[3 * cur_row + 2] = accx2;
                        }
                        acc = make_float4(0.f, 0.f, 0.f, 0.f);
                        accx0 = accx1 = accx2 = 0.f;
                        cur_row = rowbuf[rl & 31];
                    }
                    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                    if (p == 0) {
                        const float4 t4 = *reinterpret_cast<const float4*>(tx + rl * 4);
                        accx0 += t4.x; accx1 += t4.y; accx2 += t4.z;
                    }
                }
            }
            __syncthreads();                                   // (8) the images may be overwritten by the next tile
        }
        if (cur_row >= 0 && (lane >> 2) == 0)
            *reinterpret_cast<float4*>(io.gPQ + (size_t)cur_row * 2 * kH + 16 * p + 4 * (lane & 3)) = acc;
    }
    // keep every accumulator alive: one slab row per wave (wrong numbers; the real reduction is not part of the probe)
    const int slab_total = pvs_slab_layout(kH).total;
    float* dst = io.slabs + (size_t)blockIdx.x * slab_total;
    for (int i = threadIdx.x; i < slab_total; i += 256) dst[i] = 0.f;     // (finite weight gradients for the steps that follow)
    float keep = gB[0] + gB[1] + gB[2] + gB[3];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) keep += gW2[b][r] + gWc1[b][r] + g_wc2x[4 * b + r];
    if (keep == 12345.678f) dst[threadIdx.x] = keep;           // (never true for finite sums of this size: no store traffic)
}

}  // namespace pairprobe
