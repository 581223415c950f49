#include "dense_ops.h"
#include "mfma_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kRowsPerBlock = 512;   // rows one block reduces in tsgemm / colreduce
constexpr int kMaxBlocks = 512;

__global__ void __launch_bounds__(kThreads)
k_linear(float* __restrict__ y, int ldy, const float* __restrict__ x, int ldx,
         const float* __restrict__ W, int swc, int swk, const float* __restrict__ b,
         const float* __restrict__ x2, int ldx2, const float* __restrict__ W2, int swc2, int swk2,
         int N, int K, int K2, int C, int accumulate, int NB) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int KK = K + K2;
    float* Wt = smem;             // [KK][C]  transposed: lanes over c read consecutive words
    float* xs = smem + KK * C;    // [NB][KK]
    const int tid = threadIdx.x;
    for (int i = tid; i < KK * C; i += kThreads) {
        int k = i / C, c = i % C;
        Wt[i] = k < K ? W[(size_t)c * swc + (size_t)k * swk]
                      : W2[(size_t)c * swc2 + (size_t)(k - K) * swk2];
    }
    for (int n0 = blockIdx.x * NB; n0 < N; n0 += gridDim.x * NB) {
        __syncthreads();
        for (int i = tid; i < NB * KK; i += kThreads) {
            int nn = i / KK, k = i % KK, n = n0 + nn;
            float v = 0.f;
            if (n < N) v = k < K ? x[(size_t)n * ldx + k] : x2[(size_t)n * ldx2 + (k - K)];
            xs[i] = v;
        }
        __syncthreads();
        for (int o = tid; o < NB * C; o += kThreads) {
            int nn = o / C, c = o % C, n = n0 + nn;
            if (n >= N) continue;
            float acc = b ? b[c] : 0.f;
            const float* xr = xs + nn * KK;
            for (int k = 0; k < KK; ++k) acc = fmaf(xr[k], Wt[k * C + c], acc);
            float* dst = y + (size_t)n * ldy + c;
            *dst = accumulate ? *dst + acc : acc;
        }
    }
}

// Each block owns a contiguous row range and every output (c,k); outputs are dealt to threads in
// chunks of 256; partial results go to slabs[block][C*K] and are summed in block order afterwards.
template <int MAXCH>
__global__ void __launch_bounds__(kThreads)
k_tsgemm_tn(float* __restrict__ slabs, const float* __restrict__ A, int lda,
            const float* __restrict__ B, int ldb, int N, int C, int K, int rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NT = 16;
    float* As = smem;            // [NT][C]
    float* Bs = smem + NT * C;   // [NT][K]
    const int tid = threadIdx.x;
    const int CK = C * K;
    float acc[MAXCH];
    int oc[MAXCH], ok[MAXCH];
#pragma unroll
    for (int j = 0; j < MAXCH; ++j) {
        acc[j] = 0.f;
        int o = j * kThreads + tid;
        oc[j] = o < CK ? o / K : 0;
        ok[j] = o < CK ? o % K : 0;
    }
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(N, r0 + rows_per_block);
    for (int n0 = r0; n0 < r1; n0 += NT) {
        __syncthreads();
        for (int i = tid; i < NT * C; i += kThreads) {
            int nn = i / C, c = i % C, n = n0 + nn;
            As[i] = n < r1 ? A[(size_t)n * lda + c] : 0.f;
        }
        for (int i = tid; i < NT * K; i += kThreads) {
            int nn = i / K, k = i % K, n = n0 + nn;
            Bs[i] = n < r1 ? B[(size_t)n * ldb + k] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < MAXCH; ++j) {
            if (j * kThreads < CK) {
                float a = acc[j];
#pragma unroll
                for (int nn = 0; nn < NT; ++nn) a = fmaf(As[nn * C + oc[j]], Bs[nn * K + ok[j]], a);
                acc[j] = a;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < MAXCH; ++j) {
        int o = j * kThreads + tid;
        if (o < CK) slabs[(size_t)blockIdx.x * CK + o] = acc[j];
    }
}

__global__ void __launch_bounds__(kThreads)
k_colreduce(int mode, float* __restrict__ slabs, const float* __restrict__ A, int lda,
            const float* __restrict__ B, int ldb, const float* __restrict__ shift, int N, int C,
            int rows_per_block) {
    __shared__ float part[kThreads];
    const int tid = threadIdx.x;
    const int R = kThreads / C;          // row lanes per column (C <= 256)
    const int c = tid % C, r = tid / C;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(N, r0 + rows_per_block);
    float acc = 0.f;
    if (r < R) {
        const float sh = (mode == PVS_COL_SUMSQ_SHIFT) ? shift[c] : 0.f;
        for (int n = r0 + r; n < r1; n += R) {
            float a = A[(size_t)n * lda + c];
            if (mode == PVS_COL_SUM_A) acc += a;
            else if (mode == PVS_COL_SUM_AB) acc = fmaf(a, B[(size_t)n * ldb + c], acc);
            else { float d = a - sh; acc = fmaf(d, d, acc); }
        }
    }
    part[tid] = acc;
    __syncthreads();
    if (tid < C) {
        float s = 0.f;
        for (int rr = 0; rr < R; ++rr) s += part[rr * C + tid];
        slabs[(size_t)blockIdx.x * C + tid] = s;
    }
}

// The same with 16-byte loads (C % 4 == 0, 16-byte aligned rows): lane = (row lane, quad of columns),
// four rows in flight per lane.
__global__ void __launch_bounds__(kThreads)
k_colreduce4(int mode, float* __restrict__ slabs, const float* __restrict__ A, int lda,
             const float* __restrict__ B, int ldb, const float* __restrict__ shift, int N, int C,
             int rows_per_block) {
    __shared__ float4 part[kThreads];
    const int tid = threadIdx.x;
    const int QC = C / 4, R = kThreads / QC;
    const int q = tid % QC, r = tid / QC;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(N, r0 + rows_per_block);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < R) {
        const float4 sh = (mode == PVS_COL_SUMSQ_SHIFT) ? *reinterpret_cast<const float4*>(shift + 4 * q)
                                                        : make_float4(0.f, 0.f, 0.f, 0.f);
        constexpr int UN = 4;
        for (int n0 = r0 + r; n0 < r1; n0 += UN * R) {
            float4 a[UN], b[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int n = n0 + u * R;
                const bool ok = n < r1;
                a[u] = ok ? *reinterpret_cast<const float4*>(A + (size_t)n * lda + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
                if (mode == PVS_COL_SUM_AB)
                    b[u] = ok ? *reinterpret_cast<const float4*>(B + (size_t)n * ldb + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
                else if (mode == PVS_COL_SUMSQ_SHIFT)
                    a[u] = ok ? make_float4(a[u].x - sh.x, a[u].y - sh.y, a[u].z - sh.z, a[u].w - sh.w) : a[u];
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                if (mode == PVS_COL_SUM_A) { acc.x += a[u].x; acc.y += a[u].y; acc.z += a[u].z; acc.w += a[u].w; }
                else if (mode == PVS_COL_SUM_AB) {
                    acc.x = fmaf(a[u].x, b[u].x, acc.x); acc.y = fmaf(a[u].y, b[u].y, acc.y);
                    acc.z = fmaf(a[u].z, b[u].z, acc.z); acc.w = fmaf(a[u].w, b[u].w, acc.w);
                } else {
                    acc.x = fmaf(a[u].x, a[u].x, acc.x); acc.y = fmaf(a[u].y, a[u].y, acc.y);
                    acc.z = fmaf(a[u].z, a[u].z, acc.z); acc.w = fmaf(a[u].w, a[u].w, acc.w);
                }
            }
        }
    }
    part[tid] = acc;
    __syncthreads();
    if (tid < QC) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int rr = 0; rr < R; ++rr) {
            const float4 v = part[rr * QC + tid];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        *reinterpret_cast<float4*>(slabs + (size_t)blockIdx.x * C + 4 * tid) = s;
    }
}

// out[map(o)] (=|+=) scale * sum_g slabs[g][o].  One block per 32 outputs: 8 slab lanes x 32
// outputs; each slab lane sums its slabs g = lane, lane+8, ... in order, then the 8 partials are
// added in lane order: a fixed summation tree, so the result is reproducible.
__global__ void __launch_bounds__(kThreads)
k_reduce_slabs(float* __restrict__ out, int ldo, int inner, const float* __restrict__ slabs,
               int n_slabs, int width, float scale, int accumulate, int inner_valid = 1 << 30,
               float* __restrict__ extra_col = nullptr) {
    // extra_col: entry (row, inner_valid) of every row goes to extra_col[row] (the column sums of a ones column)
    __shared__ float part[8][33];
    const int ol = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int o = blockIdx.x * 32 + ol;
    float s = 0.f;
    if (o < width)
    {   // four independent partial sums per lane (loads in flight), combined in a fixed order
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int gidx = sl;
        for (; gidx + 24 < n_slabs; gidx += 32) {
            a0 += slabs[(size_t)gidx * width + o];
            a1 += slabs[(size_t)(gidx + 8) * width + o];
            a2 += slabs[(size_t)(gidx + 16) * width + o];
            a3 += slabs[(size_t)(gidx + 24) * width + o];
        }
        for (; gidx < n_slabs; gidx += 8) a0 += slabs[(size_t)gidx * width + o];
        s = (a0 + a1) + (a2 + a3);
    }
    part[sl][ol] = s;
    __syncthreads();
    if (sl == 0 && o < width && ((o % inner) < inner_valid || (extra_col && (o % inner) == inner_valid))) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += part[k][ol];
        t *= scale;
        float* dst = (o % inner) < inner_valid ? out + (size_t)(o / inner) * ldo + (o % inner) : extra_col + o / inner;
        *dst = (accumulate && (o % inner) < inner_valid) ? *dst + t : t;
    }
}

// Two slab reductions in one launch (the edge kernel's slabs and the column gather's slabs of one
// layer backward): blocks [0, blocks_a) do A, the rest B; A leaves [skip_lo, skip_hi) to B.
__global__ void __launch_bounds__(kThreads)
k_reduce_slabs2(PvsReduce2Args args) {
    __shared__ float part[8][33];
    pvs_reduce2_block(args, (int)blockIdx.x, part);
}

// The folded per-row launches (PvsLinearExt side jobs) for the 32 rows of a tile, by one wave: every store instruction
// covers whole rows' worth of consecutive bytes (a lane-per-row loop writes 16-byte pieces 128+ bytes apart and costs
// more than the launch it replaces).
__device__ __forceinline__ void pvs_tile_side_jobs(const PvsLinearExt& ext, int row0, int N, int lane) {
    if (ext.zero_rows) {
        const int zq = ext.zero_w >> 2;
        for (int idx = lane; idx < 32 * zq; idx += 64) {
            const int row = idx / zq, q = idx - row * zq;
            if (row0 + row < N)
                *reinterpret_cast<float4*>(ext.zero_rows + (size_t)(row0 + row) * ext.zero_ld + 4 * q) =
                    make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    if (ext.zero3 || ext.copy3_dst || ext.scale3_dst) {
        for (int idx = lane; idx < 96; idx += 64) {
            const int n = row0 + idx / 3;
            if (n < N) {
                const size_t o = (size_t)3 * row0 + idx;
                if (ext.zero3) ext.zero3[o] = 0.f;
                if (ext.copy3_dst) ext.copy3_dst[o] = ext.copy3_src[o];
                if (ext.scale3_dst) ext.scale3_dst[o] = ext.scale3_src[o] * ext.scale3_by[n];
            }
        }
    }
}

// ---- the node MLP of a layer as ONE launch each way (H = 32, 64; no GraphNorm, node gate or gated residual) ---------
// Every node-level product is row-local, and a product's accumulator is already the B operand of the next one (X
// layout), so the chain  y1 = [h | M] Wn1^T + b1 -> u = SiLU(y1) -> o = u Wn2^T + b2 -> h_out = (h +) o  runs per
// 32-row tile without leaving the registers; y1, u, o are written for the backward as before. What these kernels save is
// launches: a node-level launch at N = 64000 rows costs ~5 us before it moves a byte (profiles/r03_ab_small_launch_folding.txt).
template <int HB>
__global__ void __launch_bounds__(kThreads)
k_node_mlp_fwd(const float* __restrict__ h, const float* __restrict__ Magg, const float* __restrict__ W1,
               const float* __restrict__ b1, const float* __restrict__ W2, const float* __restrict__ b2, int N,
               int residual, const float* __restrict__ natt_w, const float* __restrict__ natt_b, int att_act,
               float* __restrict__ y1, float* __restrict__ u, float* __restrict__ o,
               float* __restrict__ h_out, float* __restrict__ natt_out) {
    constexpr int H = 32 * HB, LD1 = 2 * H + 1, LD2 = H + 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* W1s = smem;                    // [H][2H + 1]
    float* W2s = W1s + H * LD1;           // [H][H + 1]
    float* bs = W2s + H * LD2;            // b1 | b2 | node attention weight
    for (int i = threadIdx.x; i < H * 2 * H; i += kThreads) W1s[(i / (2 * H)) * LD1 + i % (2 * H)] = W1[i];
    for (int i = threadIdx.x; i < H * H; i += kThreads) W2s[(i / H) * LD2 + i % H] = W2[i];
    for (int i = threadIdx.x; i < 3 * H; i += kThreads)
        bs[i] = i < H ? b1[i] : (i < 2 * H ? b2[i - H] : (natt_w ? natt_w[i - 2 * H] : 0.f));
    __syncthreads();
    const float bna = natt_w ? natt_b[0] : 0.f;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int j = lane & 31, hh = lane >> 5;
    const int n_tiles = (N + 31) / 32;
    for (int tile = blockIdx.x * (kThreads / 64) + wv; tile < n_tiles; tile += gridDim.x * (kThreads / 64)) {
        const int n = tile * 32 + j;
        const bool valid = n < N;
        const size_t row = (size_t)(valid ? n : N - 1) * H;
        float v[2 * HB][16];
#pragma unroll
        for (int bb = 0; bb < 2 * HB; ++bb) {
            const float* src = (bb < HB ? h + row + 32 * bb : Magg + row + 32 * (bb - HB));
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 q = *reinterpret_cast<const float4*>(src + 8 * g + 4 * hh);
                v[bb][4 * g] = q.x; v[bb][4 * g + 1] = q.y; v[bb][4 * g + 2] = q.z; v[bb][4 * g + 3] = q.w;
            }
        }
        f32x16 acc[HB];
#pragma unroll
        for (int cb = 0; cb < HB; ++cb)
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[cb][t] = bs[32 * cb + xch(t, hh)];
        mfma_chain_rect<HB, 2 * HB, false>(W1s, lane, v, acc);
        float uv[HB][16];
#pragma unroll
        for (int cb = 0; cb < HB; ++cb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int off = 32 * cb + 8 * g + 4 * hh;
                const float4 yq = make_float4(acc[cb][4 * g], acc[cb][4 * g + 1], acc[cb][4 * g + 2], acc[cb][4 * g + 3]);
                const float4 uq = make_float4(pvs_silu(yq.x), pvs_silu(yq.y), pvs_silu(yq.z), pvs_silu(yq.w));
                uv[cb][4 * g] = uq.x; uv[cb][4 * g + 1] = uq.y; uv[cb][4 * g + 2] = uq.z; uv[cb][4 * g + 3] = uq.w;
                if (valid) {
                    *reinterpret_cast<float4*>(y1 + row + off) = yq;
                    *reinterpret_cast<float4*>(u + row + off) = uq;
                }
            }
        f32x16 acc2[HB];
#pragma unroll
        for (int cb = 0; cb < HB; ++cb)
#pragma unroll
            for (int t = 0; t < 16; ++t) acc2[cb][t] = bs[H + 32 * cb + xch(t, hh)];
        mfma_chain_rect<HB, HB, false>(W2s, lane, uv, acc2);
        float a = 1.f;
        if (natt_w) {        // node gate: o * act(w . o + b); the row's channels sit in this lane and in lane ^ 32
            float part = 0.f;
#pragma unroll
            for (int cb = 0; cb < HB; ++cb)
#pragma unroll
                for (int t = 0; t < 16; ++t) part = fmaf(bs[2 * H + 32 * cb + xch(t, hh)], acc2[cb][t], part);
            part += __shfl_xor(part, 32, 64);
            a = pvs_att_act(att_act, part + bna);
            if (valid && hh == 0 && natt_out) natt_out[n] = a;
        }
        if (valid) {
#pragma unroll
            for (int cb = 0; cb < HB; ++cb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int off = 32 * cb + 8 * g + 4 * hh;
                    float4 r = make_float4(acc2[cb][4 * g], acc2[cb][4 * g + 1], acc2[cb][4 * g + 2], acc2[cb][4 * g + 3]);
                    *reinterpret_cast<float4*>(o + row + off) = r;
                    r.x *= a; r.y *= a; r.z *= a; r.w *= a;
                    if (residual) { r.x += v[cb][4 * g]; r.y += v[cb][4 * g + 1]; r.z += v[cb][4 * g + 2]; r.w += v[cb][4 * g + 3]; }
                    *reinterpret_cast<float4*>(h_out + row + off) = r;
                }
        }
    }
}

// backward of the chain: g_y1 = (g_o Wn2) * SiLU'(y1) -> g_h (+)= g_y1 Wn1[:, :H], gM = g_y1 Wn1[:, H:], with the
// per-row preparation of the edge backward as side jobs (PvsLinearExt: zero_rows / zero3 / scale3)
template <int HB>
__global__ void __launch_bounds__(kThreads)
k_node_mlp_bwd(const float* __restrict__ g_hout, const float* __restrict__ o, const float* __restrict__ y1,
               const float* __restrict__ W1, const float* __restrict__ W2, int N, int residual,
               const float* __restrict__ natt_w, const float* __restrict__ natt_b, int att_act,
               float* __restrict__ g_o, float* __restrict__ t1, float* __restrict__ gl, float* __restrict__ g_y1,
               float* __restrict__ g_h, float* __restrict__ gM, PvsLinearExt ext) {
    // In front of the chain, the output stage's backward (node_ops.hip: k_node_out_bwd for plain / residual / node-gated
    // layers): with a node gate a = act(l), l = w . o + b:  g_o = g_hout a + g_l w,  g_l = act'(l) (g_hout . o), and
    // t1 = g_l o, gl = g_l for the gate's weight gradients; the residual's share of g_h is g_hout itself.
    constexpr int H = 32 * HB, LD = H + 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* W2t = smem;                    // [H][H + 1]:   W2t[j][c] = Wn2[c][j]
    float* W1t = W2t + H * LD;            // [2H][H + 1]:  W1t[j][c] = Wn1[c][j]
    float* wna = W1t + 2 * H * LD;        // [H] node attention weight
    for (int i = threadIdx.x; i < H * H; i += kThreads) { const int c = i / H, jj = i % H; W2t[jj * LD + c] = W2[i]; }
    for (int i = threadIdx.x; i < H * 2 * H; i += kThreads) { const int c = i / (2 * H), jj = i % (2 * H); W1t[jj * LD + c] = W1[i]; }
    for (int i = threadIdx.x; i < H; i += kThreads) wna[i] = natt_w ? natt_w[i] : 0.f;
    __syncthreads();
    const float bna = natt_w ? natt_b[0] : 0.f;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int j = lane & 31, hh = lane >> 5;
    const bool side = ext.zero_rows || ext.zero3 || ext.scale3_dst;
    const int n_tiles = (N + 31) / 32;
    for (int tile = blockIdx.x * (kThreads / 64) + wv; tile < n_tiles; tile += gridDim.x * (kThreads / 64)) {
        const int n = tile * 32 + j;
        const bool valid = n < N;
        const size_t row = (size_t)(valid ? n : N - 1) * H;
        if (side) pvs_tile_side_jobs(ext, tile * 32, N, lane);
        float v[HB][16], gres[HB][16];
        load_x<HB>(g_hout + row, hh, v);
#pragma unroll
        for (int cb = 0; cb < HB; ++cb)
#pragma unroll
            for (int t = 0; t < 16; ++t) gres[cb][t] = residual ? v[cb][t] : 0.f;
        if (natt_w) {
            float ov[HB][16];
            load_x<HB>(o + row, hh, ov);
            float lp = 0.f, dp = 0.f;
#pragma unroll
            for (int cb = 0; cb < HB; ++cb)
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    lp = fmaf(wna[32 * cb + xch(t, hh)], ov[cb][t], lp);
                    dp = fmaf(v[cb][t], ov[cb][t], dp);
                }
            lp += __shfl_xor(lp, 32, 64);
            dp += __shfl_xor(dp, 32, 64);
            const float l = lp + bna;
            const float a = pvs_att_act(att_act, l);
            const float g_l = pvs_att_act_grad(att_act, l, a) * dp;
#pragma unroll
            for (int cb = 0; cb < HB; ++cb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int off = 32 * cb + 8 * g + 4 * hh;
                    float4 gq, tq;
                    float* gp = &gq.x; float* tp = &tq.x;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int t = 4 * g + i;
                        const float gv = fmaf(g_l, wna[off + i], v[cb][t] * a);
                        tp[i] = g_l * ov[cb][t];
                        v[cb][t] = gv;
                        gp[i] = gv;
                    }
                    if (valid) {
                        *reinterpret_cast<float4*>(g_o + row + off) = gq;
                        *reinterpret_cast<float4*>(t1 + row + off) = tq;
                    }
                }
            if (valid && hh == 0) gl[n] = g_l;
        }
        f32x16 acc[HB];
#pragma unroll
        for (int cb = 0; cb < HB; ++cb)
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[cb][t] = 0.f;
        mfma_chain_rect<HB, HB, false>(W2t, lane, v, acc);
        float gy[HB][16];
#pragma unroll
        for (int cb = 0; cb < HB; ++cb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int off = 32 * cb + 8 * g + 4 * hh;
                const float4 z = *reinterpret_cast<const float4*>(y1 + row + off);
                float4 r = make_float4(acc[cb][4 * g], acc[cb][4 * g + 1], acc[cb][4 * g + 2], acc[cb][4 * g + 3]);
                r.x *= pvs_silu_grad(z.x, pvs_sigmoid(z.x)); r.y *= pvs_silu_grad(z.y, pvs_sigmoid(z.y));
                r.z *= pvs_silu_grad(z.z, pvs_sigmoid(z.z)); r.w *= pvs_silu_grad(z.w, pvs_sigmoid(z.w));
                gy[cb][4 * g] = r.x; gy[cb][4 * g + 1] = r.y; gy[cb][4 * g + 2] = r.z; gy[cb][4 * g + 3] = r.w;
                if (valid) *reinterpret_cast<float4*>(g_y1 + row + off) = r;
            }
        f32x16 acc2[2 * HB];
#pragma unroll
        for (int cb = 0; cb < 2 * HB; ++cb)
#pragma unroll
            for (int t = 0; t < 16; ++t) acc2[cb][t] = cb < HB ? gres[cb < HB ? cb : 0][t] : 0.f;
        mfma_chain_rect<2 * HB, HB, false>(W1t, lane, gy, acc2);
        if (valid) {
#pragma unroll
            for (int cb = 0; cb < 2 * HB; ++cb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float* dst = (cb < HB ? g_h + row + 32 * cb : gM + row + 32 * (cb - HB)) + 8 * g + 4 * hh;
                    *reinterpret_cast<float4*>(dst) =
                        make_float4(acc2[cb][4 * g], acc2[cb][4 * g + 1], acc2[cb][4 * g + 2], acc2[cb][4 * g + 3]);
                }
        }
    }
}

constexpr int kPoolThreads = 1024;
// One workgroup per graph (and per chunk of 1024 channels); thread = (row slot, channel), 4 independent partial sums
// per thread so that 4 * (1024 / width) rows are in flight; fixed summation order.
__global__ void __launch_bounds__(kPoolThreads)
k_mean_pool_fwd(const float* __restrict__ h, const int32_t* __restrict__ gptr,
                float* __restrict__ pooled, int ld) {
    __shared__ float part[kPoolThreads];
    const int g = blockIdx.x, tid = threadIdx.x;
    const int c0 = blockIdx.y * kPoolThreads;
    const int width = min(kPoolThreads, ld - c0);
    h += c0;
    const int n0 = gptr[g], n1 = gptr[g + 1];
    const int R = kPoolThreads / width;
    const int c = tid % width, r = tid / width;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (r < R) {
        int n = n0 + r;
        for (; n + 3 * R < n1; n += 4 * R) {
            a0 += h[(size_t)n * ld + c];
            a1 += h[(size_t)(n + R) * ld + c];
            a2 += h[(size_t)(n + 2 * R) * ld + c];
            a3 += h[(size_t)(n + 3 * R) * ld + c];
        }
        for (; n < n1; n += R) a0 += h[(size_t)n * ld + c];
    }
    part[tid] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (tid < width) {
        float s = 0.f;
        for (int rr = 0; rr < R; ++rr) s += part[rr * width + tid];
        int cnt = n1 - n0;
        pooled[(size_t)g * ld + c0 + tid] = s / (float)(cnt > 1 ? cnt : 1);
    }
}

// global_mean_pool + the first Linear of the head in one launch (one workgroup per graph): the pooled row stays in LDS
// for the product; y[g, c] = b[c] + sum_k W[c, k] pooled[g, k], k ascending (the order of k_linear).
__global__ void __launch_bounds__(kPoolThreads)
k_pool_head_fwd(const float* __restrict__ h, const int32_t* __restrict__ gptr, const float* __restrict__ W,
                const float* __restrict__ b, float* __restrict__ pooled, float* __restrict__ y, int width, int C) {
    __shared__ float part[kPoolThreads];
    __shared__ float pl[kPoolThreads];
    const int g = blockIdx.x, tid = threadIdx.x;
    const int n0 = gptr[g], n1 = gptr[g + 1];
    const int R = kPoolThreads / width;
    const int c = tid % width, r = tid / width;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (r < R) {
        int n = n0 + r;
        for (; n + 3 * R < n1; n += 4 * R) {
            a0 += h[(size_t)n * width + c];
            a1 += h[(size_t)(n + R) * width + c];
            a2 += h[(size_t)(n + 2 * R) * width + c];
            a3 += h[(size_t)(n + 3 * R) * width + c];
        }
        for (; n < n1; n += R) a0 += h[(size_t)n * width + c];
    }
    part[tid] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (tid < width) {
        float s = 0.f;
        for (int rr = 0; rr < R; ++rr) s += part[rr * width + tid];
        const int cnt = n1 - n0;
        s = s / (float)(cnt > 1 ? cnt : 1);
        pl[tid] = s;
        pooled[(size_t)g * width + tid] = s;
    }
    __syncthreads();
    for (int o = tid; o < C; o += kPoolThreads) {
        float acc = b ? b[o] : 0.f;
        const float* wr = W + (size_t)o * width;
        for (int k = 0; k < width; ++k) acc = fmaf(pl[k], wr[k], acc);
        y[(size_t)g * C + o] = acc;
    }
}

// backward of the pair: blocks [0, node_blocks) write g_h[n, k] = (sum_c g_y[g(n), c] W[c, k]) / count(g(n)); the blocks
// behind them the parameter gradients g_W[c, k] = sum_g g_y[g, c] pooled[g, k], g_b[c] = sum_g g_y[g, c] (g ascending)
__global__ void __launch_bounds__(256)
k_pool_head_bwd(const float* __restrict__ gy, const float* __restrict__ pooled, const float* __restrict__ W,
                const int32_t* __restrict__ gptr, float* __restrict__ gh, float* __restrict__ gW,
                float* __restrict__ gb, int B, int N, int width, int C, int node_blocks) {
    if ((int)blockIdx.x >= node_blocks) {
        const int o = ((int)blockIdx.x - node_blocks) * 256 + threadIdx.x;
        if (o >= C * width) return;
        const int c = o / width, k = o - c * width;
        float s = 0.f, t = 0.f;
        for (int g = 0; g < B; ++g) {
            const float v = gy[(size_t)g * C + c];
            s = fmaf(v, pooled[(size_t)g * width + k], s);
            t += v;
        }
        gW[o] = s;
        if (k == 0 && gb) gb[c] = t;
        return;
    }
    if (!gh) return;
    auto graph_of = [&](int n) {
        int lo = 0, hi = B;  // largest g with gptr[g] <= n
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (gptr[mid] <= n) lo = mid; else hi = mid;
        }
        return lo;
    };
    if ((width & 3) == 0 && ((uintptr_t)gh & 15) == 0) {      // one thread per 16 bytes of a row: one search per quad
        const int qpr = width >> 2;
        const long long total = (long long)N * qpr;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)node_blocks * 256) {
            const int n = (int)(i / qpr), k = 4 * (int)(i % qpr);
            const int g = graph_of(n);
            const int cnt = gptr[g + 1] - gptr[g];
            float4 gp = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int c = 0; c < C; ++c) {
                const float v = gy[(size_t)g * C + c];
                const float* wr = W + (size_t)c * width + k;
                gp.x = fmaf(v, wr[0], gp.x); gp.y = fmaf(v, wr[1], gp.y);
                gp.z = fmaf(v, wr[2], gp.z); gp.w = fmaf(v, wr[3], gp.w);
            }
            const float d = (float)(cnt > 1 ? cnt : 1);
            *reinterpret_cast<float4*>(gh + (size_t)n * width + k) = make_float4(gp.x / d, gp.y / d, gp.z / d, gp.w / d);
        }
        return;
    }
    const long long total = (long long)N * width;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)node_blocks * 256) {
        const int n = (int)(i / width), k = (int)(i % width);
        const int g = graph_of(n);
        const int cnt = gptr[g + 1] - gptr[g];
        float gp = 0.f;
        for (int c = 0; c < C; ++c) gp = fmaf(gy[(size_t)g * C + c], W[(size_t)c * width + k], gp);
        gh[i] = gp / (float)(cnt > 1 ? cnt : 1);
    }
}

// nn.BCEWithLogitsLoss() (mean reduction; point_neural_network_base.py:74, :365) in one launch: the loss and, kept for
// the backward, d loss / d x = (sigmoid(x) - t) / n. One workgroup, fixed summation order.
__global__ void __launch_bounds__(256)
k_bce_logits_fwd(const float* __restrict__ x, const float* __restrict__ t, int n, float* __restrict__ loss,
                 float* __restrict__ grad) {
    __shared__ float part[256];
    const int tid = threadIdx.x;
    const float inv_n = 1.0f / (float)n;
    float acc = 0.f;
    for (int i = tid; i < n; i += 256) {
        const float xi = x[i], ti = t[i];
        acc += fmaxf(xi, 0.f) - xi * ti + log1pf(expf(-fabsf(xi)));
        grad[i] = (1.0f / (1.0f + expf(-xi)) - ti) * inv_n;
    }
    part[tid] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) part[tid] += part[tid + o];
        __syncthreads();
    }
    if (tid == 0) loss[0] = part[0] * inv_n;
}

__global__ void k_scale_by_device_scalar(const float* __restrict__ a, const float* __restrict__ scalar, int n,
                                         float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] * scalar[0];
}

__global__ void k_mean_pool_bwd(const float* __restrict__ gp, const int32_t* __restrict__ gptr,
                                float* __restrict__ gh, int B, int N, int width) {
    long long total = (long long)N * width;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        int n = (int)(i / width), c = (int)(i % width);
        int lo = 0, hi = B;  // largest g with gptr[g] <= n
        while (hi - lo > 1) {
            int mid = (lo + hi) >> 1;
            if (gptr[mid] <= n) lo = mid; else hi = mid;
        }
        int cnt = gptr[lo + 1] - gptr[lo];
        gh[i] = gp[(size_t)lo * width + c] / (float)(cnt > 1 ? cnt : 1);
    }
}

// MFMA linear for 32-aligned shapes: y[n][c] (=|+=) b[c] + sum_k x[n][k] W(c,k) (+ second input).
// Rows (nodes) sit on the lanes: the X-layout operand of the products is read straight from the
// row-major input (16-byte pieces), the weights are staged once per block in LDS (natural, padded
// rows), the accumulator is stored back as rows. 32 rows per wave-tile.
// The product as a workgroup-level routine (workgroup `block` of `n_blocks`, its weights staged in `smem`:
// 32 CB (32 KB + 1) + 32 CB floats): k_linear_mfma is this routine alone; the weight-gradient pass of a layer's backward
// carries it as one of its workgroup roles (independent work in one launch).
template <int KB, int CB>
__device__ __forceinline__ void linear_mfma_block(float* __restrict__ smem, int block, int n_blocks,
              float* __restrict__ y, int ldy, const float* __restrict__ x, int ldx, int kb1,
              const float* __restrict__ x2, int ldx2, const float* __restrict__ W, int swc, int swk,
              const float* __restrict__ W2, int swc2, int swk2, const float* __restrict__ b, int N,
              int accumulate, int epi, const float* __restrict__ aux_in, int ld_in, float* __restrict__ aux_out,
              int ld_out, const PvsLinearExt& ext) {
    // epi (elementwise epilogue on the accumulator, saves a pass over [N,C]):
    //   1: aux_out = SiLU(y)          2: aux_out = aux_in + y          3: y *= SiLU'(aux_in)
    //   4: aux_out = y
    constexpr int K = 32 * KB, C = 32 * CB, LD = K + 1;
    float* Wn = smem;
    float* bias = smem + C * LD;
    const int k1 = 32 * kb1;
    // blockIdx.y = group of CB column blocks (ext.groups > 1: a product with 32 CB groups outputs as `groups` sets of
    // workgroups that run side by side - P | Q at H = 64)
    if (ext.side_group && (int)blockIdx.y == ext.groups) {      // the group that only does the per-row side jobs
        const int n_tiles_s = (N + 31) / 32;
        for (int tile = block * (kThreads / 64) + (threadIdx.x >> 6); tile < n_tiles_s; tile += n_blocks * (kThreads / 64))
            pvs_tile_side_jobs(ext, tile * 32, N, threadIdx.x & 63);
        return;
    }
    const int c0 = (int)blockIdx.y * C;
    y += c0;
    if (aux_in) aux_in += c0;
    if (aux_out) aux_out += c0;
    for (int i = threadIdx.x; i < C * K; i += kThreads) {
        const int c = i / K, k = i % K;
        const long long sh = (c0 + c >= 32 * ext.shift_block) ? ext.w_shift1 : 0;
        Wn[c * LD + k] = k < k1 ? W[(long long)(c0 + c) * swc + (long long)k * swk + sh]
                                : W2[(size_t)(c0 + c) * swc2 + (size_t)(k - k1) * swk2];
    }
    for (int c = threadIdx.x; c < C; c += kThreads) bias[c] = (b && c0 + c < 32 * ext.bias_blocks) ? b[c0 + c] : 0.f;
    __syncthreads();
    const bool side = blockIdx.y == 0 && !ext.side_group && (ext.zero_rows || ext.zero3 || ext.copy3_dst || ext.scale3_dst);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int j = lane & 31, hh = lane >> 5;
    const int n_tiles = (N + 31) / 32;
    for (int tile = block * (kThreads / 64) + wv; tile < n_tiles; tile += n_blocks * (kThreads / 64)) {
        const int n = tile * 32 + j;
        const bool valid = n < N;
        const int nn = valid ? n : N - 1;
        float v[KB][16];
#pragma unroll
        for (int bb = 0; bb < KB; ++bb) {
            const float* src = bb < kb1 ? x + (size_t)nn * ldx + 32 * bb
                                        : x2 + (size_t)nn * ldx2 + 32 * (bb - kb1);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 q = *reinterpret_cast<const float4*>(src + 8 * g + 4 * hh);
                v[bb][4 * g] = q.x; v[bb][4 * g + 1] = q.y; v[bb][4 * g + 2] = q.z; v[bb][4 * g + 3] = q.w;
            }
        }
        if (side) pvs_tile_side_jobs(ext, tile * 32, N, lane);      // the folded per-row launches
        f32x16 acc[CB];
        float* dst0 = y + (size_t)nn * ldy;
        float* dst1 = (CB > 1 && ext.y1) ? ext.y1 + (size_t)nn * ext.ldy1 - 32 : dst0;
        const int acc_flag1 = ext.acc1 < 0 ? accumulate : ext.acc1;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bq = *reinterpret_cast<const float4*>(bias + 32 * cb + 8 * g + 4 * hh);
                float4 init = bq;
                float* dst = cb == 1 ? dst1 : dst0;
                if (cb == 1 ? acc_flag1 : accumulate) {
                    const float4 old = *reinterpret_cast<const float4*>(dst + 32 * cb + 8 * g + 4 * hh);
                    init.x += old.x; init.y += old.y; init.z += old.z; init.w += old.w;
                }
                acc[cb][4 * g] = init.x; acc[cb][4 * g + 1] = init.y;
                acc[cb][4 * g + 2] = init.z; acc[cb][4 * g + 3] = init.w;
            }
        mfma_chain_rect<CB, KB, false>(Wn, lane, v, acc);
        if (valid) {
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int off = 32 * cb + 8 * g + 4 * hh;
                    float* dst = cb == 1 ? dst1 : dst0;
                    float4 r = make_float4(acc[cb][4 * g], acc[cb][4 * g + 1], acc[cb][4 * g + 2], acc[cb][4 * g + 3]);
                    if (epi == 3) {
                        const float4 z = *reinterpret_cast<const float4*>(aux_in + (size_t)nn * ld_in + off);
                        r.x *= pvs_silu_grad(z.x, pvs_sigmoid(z.x)); r.y *= pvs_silu_grad(z.y, pvs_sigmoid(z.y));
                        r.z *= pvs_silu_grad(z.z, pvs_sigmoid(z.z)); r.w *= pvs_silu_grad(z.w, pvs_sigmoid(z.w));
                    }
                    *reinterpret_cast<float4*>(dst + off) = r;
                    if (epi == 1) {
                        *reinterpret_cast<float4*>(aux_out + (size_t)nn * ld_out + off) =
                            make_float4(pvs_silu(r.x), pvs_silu(r.y), pvs_silu(r.z), pvs_silu(r.w));
                    } else if (epi == 2) {
                        const float4 a = *reinterpret_cast<const float4*>(aux_in + (size_t)nn * ld_in + off);
                        *reinterpret_cast<float4*>(aux_out + (size_t)nn * ld_out + off) =
                            make_float4(a.x + r.x, a.y + r.y, a.z + r.z, a.w + r.w);
                    } else if (epi == 4) {
                        *reinterpret_cast<float4*>(aux_out + (size_t)nn * ld_out + off) = r;
                    }
                }
        }
    }
}

template <int KB, int CB>
__global__ void __launch_bounds__(kThreads)
k_linear_mfma(float* __restrict__ y, int ldy, const float* __restrict__ x, int ldx, int kb1,
              const float* __restrict__ x2, int ldx2, const float* __restrict__ W, int swc, int swk,
              const float* __restrict__ W2, int swc2, int swk2, const float* __restrict__ b, int N,
              int accumulate, int epi, const float* __restrict__ aux_in, int ld_in, float* __restrict__ aux_out,
              int ld_out, PvsLinearExt ext) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    linear_mfma_block<KB, CB>(smem, (int)blockIdx.x, (int)gridDim.x, y, ldy, x, ldx, kb1, x2, ldx2, W, swc, swk, W2, swc2,
                              swk2, b, N, accumulate, epi, aux_in, ld_in, aux_out, ld_out, ext);
}

// MFMA weight-gradient product for 32-aligned shapes: out[c][k] = sum_n A[n][c] * B[n][k].
// The sum runs over rows, so both operands are read straight from the row-major arrays in
// A/B-operand order (lane l: row n + (l>>5), column 32*blk + (l&31): two coalesced 128-B runs per
// instruction) - no LDS staging. Each wave walks its rows two at a time; the 4 waves of a block
// are added in wave order through LDS; block partials go to slabs.
typedef f32x16 f32x16_t;

template <int CB, int KB>
__global__ void __launch_bounds__(kThreads)
k_tsgemm_mfma(float* __restrict__ slabs, const float* __restrict__ A, int lda,
              const float* __restrict__ B, int ldb, int N, int rows_per_block, int kvalid, int ones_col = 0) {
    // kvalid: number of real columns of B (< 32*KB: the rest of the block is zero; slab stays padded).
    // ones_col: column kvalid of the padded B is 1 for every row, so column kvalid of the product is the column sums of A
    // (a bias gradient out of the same pass)
    __shared__ float red[CB * KB * 1024];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int j = lane & 31, hh = lane >> 5;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(N, r0 + rows_per_block);
    f32x16_t acc[CB][KB];
#pragma unroll
    for (int a = 0; a < CB; ++a)
#pragma unroll
        for (int b = 0; b < KB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    constexpr int UN = 4;
    for (int n0 = r0 + wv * 2 * UN; n0 < r1; n0 += kThreads / 64 * 2 * UN) {
        float av[UN][CB], bv[UN][KB];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int n = n0 + 2 * u + hh;
            const bool ok = n < r1;
#pragma unroll
            for (int a = 0; a < CB; ++a) av[u][a] = ok ? A[(size_t)n * lda + 32 * a + j] : 0.f;
#pragma unroll
            for (int b = 0; b < KB; ++b)
                bv[u][b] = (ok && 32 * b + j < kvalid) ? B[(size_t)n * ldb + 32 * b + j]
                                                       : ((ok && ones_col && 32 * b + j == kvalid) ? 1.f : 0.f);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int a = 0; a < CB; ++a)
#pragma unroll
                for (int b = 0; b < KB; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][a], bv[u][b], acc[a][b], 0, 0, 0);
    }
    // D layout: register r of lane l = out[c = 32a + (r&3) + 8(r>>2) + 4hh][k = 32b + j]
    for (int i = threadIdx.x; i < CB * KB * 1024; i += kThreads) red[i] = 0.f;
    __syncthreads();
    for (int turn = 0; turn < kThreads / 64; ++turn) {
        if (wv == turn) {
#pragma unroll
            for (int a = 0; a < CB; ++a)
#pragma unroll
                for (int b = 0; b < KB; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int c = (r & 3) + 8 * (r >> 2) + 4 * hh;
                        red[((a * KB + b) * 32 + c) * 32 + j] += acc[a][b][r];
                    }
        }
        __syncthreads();
    }
    // slab layout = row-major [C][K] like the generic kernel
    constexpr int K = 32 * KB;
    float* dst = slabs + (size_t)blockIdx.x * (CB * KB * 1024);
    for (int i = threadIdx.x; i < CB * KB * 1024; i += kThreads) {
        const int jj = i & 31, c = (i >> 5) & 31, blk = i >> 10;
        const int a = blk / KB, b = blk % KB;
        dst[(32 * a + c) * K + 32 * b + jj] = red[i];
    }
}

// ---- fused node-level weight gradients (see dense_ops.h) ----
// Products (A, B): (g_o, u) (g_y1, h) (g_y1, Magg) (gP, h) (gQ, h); column sums of g_o, g_y1, gP.
// blockIdx.y = 32x32 sub-block (bo, bi) of every product; per wave 5 accumulators; rows as the MFMA
// k index, operands straight from row-major HBM (as k_tsgemm_mfma).
constexpr int kWgProducts = PVS_WG_PRODUCTS, kWgBias = PVS_WG_BIAS;
constexpr int kWgSlab = PVS_WG_SLAB;   // floats per (row block, sub-block)
constexpr int kWgRowsPerBlock = 256;

// GH: the g_h product role is compiled in (H = 32). At H = 64 that role needs 208 registers - two waves per SIMD for
// the WHOLE kernel, half of the weight-gradient row blocks waiting for a slot - so there it is a launch of its own.
template <int HB, bool GH>
__global__ void __launch_bounds__(kThreads)
k_node_wgrads(float* __restrict__ slabs, PvsNodeWgradIn in, int N, int row_blocks, PvsReduce2Args extra,
              PvsGhJob gh) {
    constexpr int H = 32 * HB;
    // (sized for the product role's weights: 32 HB (64 HB + 1) + 32 HB floats at HB = 2)
    constexpr int kRedWords = (!GH || kWgSlab > H * (2 * H + 1) + H) ? kWgSlab : H * (2 * H + 1) + H;
    __shared__ __attribute__((aligned(16))) float red[kRedWords];
    // The first workgroups carry the two other jobs of this point of a layer's backward, both independent of the weight
    // gradients: the edge-slab reduction, and g_h += [gP | gQ] W1 (the product that the next layer's backward waits
    // for). FIRST in dispatch order (round 4): behind the row blocks they only started when those had drained - the
    // launch took the SUM of the three jobs' times (cfg3: 81 + 14 + 31 us = 132 us per launch).
    const int n_extra = extra.blocks() + (GH ? gh.blocks : 0);
    if ((int)blockIdx.x < n_extra) {
        if (blockIdx.y != 0) return;
        const int b = (int)blockIdx.x;
        if (b < extra.blocks()) { pvs_reduce2_block(extra, b, reinterpret_cast<float(*)[33]>(red)); return; }
        if constexpr (GH)
            linear_mfma_block<2 * HB, HB>(red, b - extra.blocks(), gh.blocks, gh.g_h, H, gh.gPQ, 2 * H, HB, gh.gPQ + H,
                                          2 * H, gh.W1, 1, gh.ld1, gh.W1 + gh.off_q, 1, gh.ld1, nullptr, N, 1, 0,
                                          nullptr, 0, nullptr, 0, PvsLinearExt{});
        return;
    }
    const int rbi = (int)blockIdx.x - n_extra;          // row block
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int j = lane & 31, hh = lane >> 5;
    const int bo = blockIdx.y / HB, bi = blockIdx.y % HB;
    const int r0 = rbi * kWgRowsPerBlock;
    const int r1 = min(N, r0 + kWgRowsPerBlock);
    const float* A0 = in.g_o + 32 * bo + j;
    const float* A1 = in.g_y1 + 32 * bo + j;
    const float* A2 = in.gPQ + 32 * bo + j;
    const float* A3 = in.gPQ + H + 32 * bo + j;
    const float* B0 = in.u + 32 * bi + j;
    const float* B1 = in.h + 32 * bi + j;
    const float* B2 = in.Magg + 32 * bi + j;
    f32x16_t acc[kWgProducts];
#pragma unroll
    for (int p = 0; p < kWgProducts; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
    float bs0 = 0.f, bs1 = 0.f, bs2 = 0.f, bs3 = 0.f, bs4 = 0.f;
    const bool gate_sums = in.t1 != nullptr && bi == 0;      // node gate: column sums of t1, sum of gl
    const float* T1 = in.t1 + 32 * bo + j;
    constexpr int UN = 4;
    // Two operand buffers: the rows of step k + 1 are in flight while the products of step k issue (round 4). A wave
    // used to run load - wait - 20 MFMAs eight times over, the whole HBM / L2 latency exposed every time: 81 us per
    // launch at H = 64 for 2.6 GFLOP and 115 MB.
    struct Rows { float a0[UN], a1[UN], a2[UN], a3[UN], b0[UN], b1[UN], b2[UN], t1[UN], gl[UN]; };
    auto fetch = [&](int n0, Rows& R) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int n = n0 + 2 * u + hh;
            const bool ok = n < r1;
            const size_t nh = (size_t)n * H, n2h = (size_t)n * 2 * H;
            R.a0[u] = ok ? A0[nh] : 0.f;
            R.a1[u] = ok ? A1[nh] : 0.f;
            R.a2[u] = ok ? A2[n2h] : 0.f;
            R.a3[u] = ok ? A3[n2h] : 0.f;
            R.b0[u] = ok ? B0[nh] : 0.f;
            R.b1[u] = ok ? B1[nh] : 0.f;
            R.b2[u] = ok ? B2[nh] : 0.f;
            R.t1[u] = (gate_sums && ok) ? T1[nh] : 0.f;
            R.gl[u] = (gate_sums && ok && bo == 0 && j == 0) ? in.gl[n] : 0.f;
        }
    };
    auto products = [&](const Rows& R) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(R.a0[u], R.b0[u], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(R.a1[u], R.b1[u], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(R.a1[u], R.b2[u], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(R.a2[u], R.b1[u], acc[3], 0, 0, 0);
            acc[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(R.a3[u], R.b1[u], acc[4], 0, 0, 0);
            bs0 += R.a0[u]; bs1 += R.a1[u]; bs2 += R.a2[u]; bs3 += R.t1[u]; bs4 += R.gl[u];
        }
    };
    constexpr int kStep = kThreads / 64 * 2 * UN;        // rows per step of the workgroup
    Rows Ra, Rb;
    fetch(r0 + wv * 2 * UN, Ra);
    for (int n0 = r0 + wv * 2 * UN; n0 < r1; n0 += 2 * kStep) {      // (same row order as the single-buffer loop)
        fetch(n0 + kStep, Rb);
        products(Ra);
        fetch(n0 + 2 * kStep, Ra);
        products(Rb);
    }
    bs0 += __shfl_xor(bs0, 32, 64); bs1 += __shfl_xor(bs1, 32, 64); bs2 += __shfl_xor(bs2, 32, 64);
    bs3 += __shfl_xor(bs3, 32, 64); bs4 += __shfl_xor(bs4, 32, 64);
    for (int i = threadIdx.x; i < kWgSlab; i += kThreads) red[i] = 0.f;
    __syncthreads();
    for (int turn = 0; turn < kThreads / 64; ++turn) {   // fixed wave order
        if (wv == turn) {
#pragma unroll
            for (int p = 0; p < kWgProducts; ++p)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int c = (r & 3) + 8 * (r >> 2) + 4 * hh;
                    red[(p * 32 + c) * 32 + j] += acc[p][r];
                }
            if (hh == 0) {
                red[kWgProducts * 1024 + j] += bs0;
                red[kWgProducts * 1024 + 32 + j] += bs1;
                red[kWgProducts * 1024 + 64 + j] += bs2;
                red[kWgProducts * 1024 + 96 + j] += bs3;
                red[kWgProducts * 1024 + 128 + j] += bs4;
            }
        }
        __syncthreads();
    }
    float* dst = slabs + ((size_t)rbi * gridDim.y + blockIdx.y) * kWgSlab;
    for (int i = threadIdx.x; i < kWgSlab; i += kThreads) dst[i] = red[i];
}

__global__ void k_node_wgrads_scatter(const float* __restrict__ gsum, PvsNodeWgradOut out, int H) {
    pvs_node_wgrads_scatter(gsum, out, H, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

int grid_for(long long work_items, int per_block) {
    long long b = (work_items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;
    return (int)b;
}

}  // namespace

int pvs_reduce_blocks(int N) {
    int b = (N + kRowsPerBlock - 1) / kRowsPerBlock;
    if (b < 1) b = 1;
    if (b > kMaxBlocks) b = kMaxBlocks;
    return b;
}

// column reductions read each row once: more, smaller blocks than the products (at most 4x
// pvs_reduce_blocks, which is what the callers' slab buffers are sized for)
int pvs_colreduce_blocks(int N) {
    int b = (N + 127) / 128;
    if (b < 1) b = 1;
    if (b > kMaxBlocks) b = kMaxBlocks;
    return b;
}

static int rows_per_block_for(int N, int blocks) { return (N + blocks - 1) / blocks; }

bool pvs_linear_epilogue_supported(int ldy, int ldx, int ldx2, int K, int K2, int C, const void* y,
                                   const void* x, const void* x2) {
    const int KK = K + K2;
    const bool aligned16 = ((ldx | ldy | ldx2) & 3) == 0 && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)x2) & 15) == 0;
    return aligned16 && K % 32 == 0 && K2 % 32 == 0 && C % 32 == 0 && C <= 64 && (KK == 32 || KK == 64 || KK == 128);
}

int pvs_launch_linear(hipStream_t s, float* y, int ldy, const float* x, int ldx, const float* W,
                      int swc, int swk, const float* b, const float* x2, int ldx2, const float* W2,
                      int swc2, int swk2, int N, int K, int K2, int C, bool accumulate, int epi,
                      const float* aux_in, int ld_in, float* aux_out, int ld_out, const PvsLinearExt* ext) {
    PVS_REQUIRE(C >= 1, "linear: n_out %d unsupported", C);
    if (ext && ext->groups > 1) {
        // `groups` sets of workgroups, each the product for its own C / groups output columns (one launch, grid.y)
        const int G = ext->groups, Cg = C / G;
        PVS_REQUIRE(C % G == 0 && epi == 0 && !ext->y1 && !x2 && K2 == 0 &&
                    ext->zero_w % 4 == 0 && (ext->zero_ld & 3) == 0 && ((uintptr_t)ext->zero_rows & 15) == 0 &&
                    pvs_linear_epilogue_supported(ldy, ldx, 0, K, 0, Cg, y, x, nullptr),
                    "linear: grouped launch of %d x %d outputs unsupported", G, Cg);
        if (N <= 0) return 0;
        const int rows_g = N >= 32768 ? 256 : 128;
        int blocks_g = (N + rows_g - 1) / rows_g;
        if (blocks_g > 1024) blocks_g = 1024;
        const size_t lds_g = (size_t)(Cg * (K + 1) + Cg) * sizeof(float);
        const PvsLinearExt e = *ext;
#define PVS_LIN_G(KBV, CBV)                                                                                        \
        k_linear_mfma<KBV, CBV><<<dim3(blocks_g, G + (e.side_group ? 1 : 0)), kThreads, lds_g, s>>>(y, ldy, x, ldx, K / 32, nullptr, 0, W, swc, swk, \
                                                                           nullptr, 0, 0, b, N, accumulate ? 1 : 0, 0,  \
                                                                           nullptr, 0, nullptr, 0, e)
        const int kbg = K / 32, cbg = Cg / 32;
        if (kbg == 1 && cbg == 1) PVS_LIN_G(1, 1);
        else if (kbg == 2 && cbg == 1) PVS_LIN_G(2, 1);
        else if (kbg == 4 && cbg == 1) PVS_LIN_G(4, 1);
        else if (kbg == 1 && cbg == 2) PVS_LIN_G(1, 2);
        else if (kbg == 2 && cbg == 2) PVS_LIN_G(2, 2);
        else PVS_LIN_G(4, 2);
#undef PVS_LIN_G
        PVS_CHECK_LAUNCH();
        return 0;
    }
    PVS_REQUIRE(!ext || (pvs_linear_epilogue_supported(ldy, ldx, x2 ? ldx2 : 0, K, K2, C, y, x, x2) &&
                         ext->zero_w % 8 == 0 && (ext->zero_ld & 3) == 0 && ((uintptr_t)ext->zero_rows & 15) == 0 &&
                         (!ext->y1 || (C == 64 && (ext->ldy1 & 3) == 0 && ((uintptr_t)ext->y1 & 15) == 0))),
                "linear: the extras need the MFMA path");
    const PvsLinearExt ext_v = ext ? *ext : PvsLinearExt{};
    if (C > kThreads) {
        // more output channels than one pass holds (one per thread): chunks of the output dimension (the input
        // gradient of a layer whose input is wider than 256: edge_mlp.0 at hidden size 128 on the decomposed path)
        PVS_REQUIRE(epi == 0 && !x2, "linear: n_out %d > %d needs the plain form", C, kThreads);
        for (int c0 = 0; c0 < C; c0 += kThreads) {
            const int rc = pvs_launch_linear(s, y + c0, ldy, x, ldx, W + (size_t)c0 * swc, swc, swk, b ? b + c0 : nullptr,
                                             nullptr, 0, nullptr, 0, 0, N, K, 0, C - c0 < kThreads ? C - c0 : kThreads,
                                             accumulate, 0, nullptr, 0, nullptr, 0);
            if (rc) return rc;
        }
        return 0;
    }
    PVS_REQUIRE(epi == 0 || pvs_linear_epilogue_supported(ldy, ldx, x2 ? ldx2 : 0, K, K2, C, y, x, x2),
                "linear: the epilogue needs the MFMA path");
    if (N <= 0) return 0;
    const int KK = K + K2;
    const bool aligned16 = ((ldx | ldy | (x2 ? ldx2 : 0)) & 3) == 0 &&
                           (((uintptr_t)x | (uintptr_t)y | (uintptr_t)x2) & 15) == 0;
    // the wide layer (hidden 128): the MFMA kernel is built for <= 64 outputs and <= 128 inputs, so 128 (or 256)
    // outputs go in chunks of 64 and a second input as a second, accumulating pass
    auto mfma_k = [](int k) { return k == 32 || k == 64 || k == 128; };
    if (aligned16 && epi == 0 && C > 64 && C % 64 == 0 && mfma_k(K) && (K2 == 0 || mfma_k(K2)) &&
        (KK > 128 || C > 64)) {
        for (int c0 = 0; c0 < C; c0 += 64) {
            int rc = pvs_launch_linear(s, y + c0, ldy, x, ldx, W + (size_t)c0 * swc, swc, swk, b ? b + c0 : nullptr,
                                       nullptr, 0, nullptr, 0, 0, N, K, 0, 64, accumulate, 0, nullptr, 0, nullptr, 0);
            if (rc) return rc;
            if (K2 > 0) {
                rc = pvs_launch_linear(s, y + c0, ldy, x2, ldx2, W2 + (size_t)c0 * swc2, swc2, swk2, nullptr, nullptr, 0,
                                       nullptr, 0, 0, N, K2, 0, 64, true, 0, nullptr, 0, nullptr, 0);
                if (rc) return rc;
            }
        }
        return 0;
    }
    if (aligned16 && K % 32 == 0 && K2 % 32 == 0 && C % 32 == 0 && C <= 64 && (KK == 32 || KK == 64 || KK == 128)) {
        const int kb = KK / 32, cb = C / 32;
        const size_t lds_m = (size_t)(C * (KK + 1) + C) * sizeof(float);
#ifndef PVS_LIN_ROWS
#define PVS_LIN_ROWS 256   // rows per block: 2 tiles per wave amortise the weight staging (cfg3 step -0.4 %)
#endif
        // (few rows - 4-graph batches of the wide layers, 8000 nodes: 32 blocks of 256 rows leave 7/8 of the chip idle
        // behind the weight staging - one 32-row tile per wave)
        const int rows_m = N >= 32768 ? PVS_LIN_ROWS : 128;
        int blocks_m = (N + rows_m - 1) / rows_m;
        if (blocks_m > 1024) blocks_m = 1024;
#define PVS_LIN(KBV, CBV)                                                                          \
    k_linear_mfma<KBV, CBV><<<blocks_m, kThreads, lds_m, s>>>(y, ldy, x, ldx, K / 32, x2, ldx2, W, swc, \
                                                              swk, W2, swc2, swk2, b, N, accumulate ? 1 : 0, \
                                                              epi, aux_in, ld_in, aux_out, ld_out, ext_v)
        if (kb == 1 && cb == 1) PVS_LIN(1, 1);
        else if (kb == 2 && cb == 1) PVS_LIN(2, 1);
        else if (kb == 4 && cb == 1) PVS_LIN(4, 1);
        else if (kb == 1 && cb == 2) PVS_LIN(1, 2);
        else if (kb == 2 && cb == 2) PVS_LIN(2, 2);
        else PVS_LIN(4, 2);
#undef PVS_LIN
        PVS_CHECK_LAUNCH();
        return 0;
    }
    int NB = kThreads / C > 0 ? kThreads / C : 1;      // rows per pass: one output per thread ...
    const int fit = (160 * 1024 / (int)sizeof(float) - KK * C) / KK;
    if (NB > fit) NB = fit;                            // ... or as many input rows as fit beside the weights
    PVS_REQUIRE(NB >= 1, "linear: %d x %d weights do not fit LDS", KK, C);
    size_t lds = (size_t)(KK * C + NB * KK) * sizeof(float);
    if (lds > 48 * 1024)
        PVS_CHECK_HIP(hipFuncSetAttribute((const void*)k_linear,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int blocks = grid_for(N, NB * 4);
    k_linear<<<blocks, kThreads, lds, s>>>(y, ldy, x, ldx, W, swc, swk, b, x2, ldx2, W2, swc2, swk2,
                                           N, K, K2, C, accumulate ? 1 : 0, NB);
    PVS_CHECK_LAUNCH();
    return 0;
}

bool pvs_node_mlp_fused_supported(int H, const void* a, const void* b, const void* c, const void* d) {
    return (H == 32 || H == 64) && (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d) & 15) == 0;
}

static int node_mlp_blocks(int N) {
    const int rows_m = N >= 32768 ? 256 : 128;
    int blocks = (N + rows_m - 1) / rows_m;
    return blocks > 1024 ? 1024 : (blocks < 1 ? 1 : blocks);
}

int pvs_launch_node_mlp_fwd(hipStream_t s, int H, int N, const float* h, const float* Magg, const float* W1,
                            const float* b1, const float* W2, const float* b2, bool residual, const float* natt_w,
                            const float* natt_b, int att_act, float* y1, float* u, float* o, float* h_out,
                            float* natt_out) {
    PVS_REQUIRE(pvs_node_mlp_fused_supported(H, h, Magg, y1, h_out) && (((uintptr_t)u | (uintptr_t)o) & 15) == 0,
                "node_mlp_fwd: H = %d or the alignment is unsupported", H);
    PVS_REQUIRE(!natt_w || natt_b, "node_mlp_fwd: node attention bias missing");
    if (N <= 0) return 0;
    const size_t lds = (size_t)(H * (2 * H + 1) + H * (H + 1) + 3 * H) * sizeof(float);
    if (H == 32) {
        k_node_mlp_fwd<1><<<node_mlp_blocks(N), kThreads, lds, s>>>(h, Magg, W1, b1, W2, b2, N, residual ? 1 : 0, natt_w,
                                                                    natt_b, att_act, y1, u, o, h_out, natt_out);
    } else {
        static bool raised = false;      // (more than 48 KB of dynamic LDS: once per process)
        if (!raised) {
            PVS_CHECK_HIP(hipFuncSetAttribute((const void*)k_node_mlp_fwd<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            raised = true;
        }
        k_node_mlp_fwd<2><<<node_mlp_blocks(N), kThreads, lds, s>>>(h, Magg, W1, b1, W2, b2, N, residual ? 1 : 0, natt_w,
                                                                    natt_b, att_act, y1, u, o, h_out, natt_out);
    }
    PVS_CHECK_LAUNCH();
    return 0;
}

int pvs_launch_node_mlp_bwd(hipStream_t s, int H, int N, const float* g_hout, const float* o, const float* y1,
                            const float* W1, const float* W2, bool residual, const float* natt_w, const float* natt_b,
                            int att_act, float* g_o, float* t1, float* gl, float* g_y1, float* g_h, float* gM,
                            const PvsLinearExt* ext) {
    PVS_REQUIRE(pvs_node_mlp_fused_supported(H, g_hout, y1, g_y1, g_h) && ((uintptr_t)gM & 15) == 0,
                "node_mlp_bwd: H = %d or the alignment is unsupported", H);
    PVS_REQUIRE(!natt_w || (natt_b && o && g_o && t1 && gl && (((uintptr_t)o | (uintptr_t)g_o | (uintptr_t)t1) & 15) == 0),
                "node_mlp_bwd: node attention needs o, g_o, t1, gl");
    PVS_REQUIRE(!ext || (ext->zero_w % 8 == 0 && (ext->zero_ld & 3) == 0 && ((uintptr_t)ext->zero_rows & 15) == 0),
                "node_mlp_bwd: bad side job");
    if (N <= 0) return 0;
    const PvsLinearExt e = ext ? *ext : PvsLinearExt{};
    const size_t lds = (size_t)(3 * H * (H + 1) + H) * sizeof(float);
    if (H == 32) {
        k_node_mlp_bwd<1><<<node_mlp_blocks(N), kThreads, lds, s>>>(g_hout, o, y1, W1, W2, N, residual ? 1 : 0, natt_w, natt_b,
                                                                    att_act, g_o, t1, gl, g_y1, g_h, gM, e);
    } else {
        static bool raised = false;
        if (!raised) {
            PVS_CHECK_HIP(hipFuncSetAttribute((const void*)k_node_mlp_bwd<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            raised = true;
        }
        k_node_mlp_bwd<2><<<node_mlp_blocks(N), kThreads, lds, s>>>(g_hout, o, y1, W1, W2, N, residual ? 1 : 0, natt_w, natt_b,
                                                                    att_act, g_o, t1, gl, g_y1, g_h, gM, e);
    }
    PVS_CHECK_LAUNCH();
    return 0;
}

int pvs_launch_reduce_slabs(hipStream_t s, float* out, int ldo, int inner, const float* slabs,
                            int n_slabs, int width, bool accumulate) {
    k_reduce_slabs<<<(width + 31) / 32, kThreads, 0, s>>>(out, ldo, inner, slabs, n_slabs, width, 1.0f,
                                                          accumulate ? 1 : 0);
    PVS_CHECK_LAUNCH();
    return 0;
}

int pvs_launch_reduce_slabs2(hipStream_t s, float* out_a, const float* slabs_a, int n_a, int width_a, int skip_lo,
                             int skip_hi, float* out_b, const float* slabs_b, int n_b, int width_b) {
    PvsReduce2Args a;
    a.out_a = out_a; a.slabs_a = slabs_a; a.n_a = n_a; a.width_a = width_a; a.skip_lo = skip_lo; a.skip_hi = skip_hi;
    a.out_b = out_b; a.slabs_b = slabs_b; a.n_b = n_b; a.width_b = width_b;
    k_reduce_slabs2<<<a.blocks(), kThreads, 0, s>>>(a);
    PVS_CHECK_LAUNCH();
    return 0;
}

bool pvs_tsgemm_colsum_supported(int N, int C, int K) { return C % 32 == 0 && C <= 64 && K < 32 && N >= 1024; }

int pvs_launch_tsgemm_tn(hipStream_t s, float* out, int ldo, const float* A, int lda, const float* B,
                         int ldb, int N, int C, int K, float* slabs, bool accumulate, float* colsum_out) {
    PVS_REQUIRE(!colsum_out || pvs_tsgemm_colsum_supported(N, C, K), "tsgemm: column sums need the narrow MFMA path");
    const int CK = C * K;
    PVS_REQUIRE(C <= 32 * kThreads, "tsgemm: %d x %d outputs unsupported", C, K);
    if (C % 64 == 0 && K % 64 == 0 && (C > 64 || K > 64)) {
        // the wide layer (hidden 128): 64 x 64 blocks of the output on the MFMA kernel, one product each. The caller's
        // slab buffer holds pvs_reduce_blocks(N) slabs of the WHOLE product, i.e. (C K / 4096) times as many of a
        // 64 x 64 block: with few rows (4-graph batches: 8000 nodes = 16 row blocks) the rows are cut finer
        int blocks = pvs_reduce_blocks(N) * (CK / 4096);
        const int by_rows = (N + 127) / 128;
        if (blocks > by_rows) blocks = by_rows;
        if (blocks > kMaxBlocks) blocks = kMaxBlocks;
        if (blocks < 1) blocks = 1;
        const int rpb = rows_per_block_for(N, blocks);
        for (int c0 = 0; c0 < C; c0 += 64)
            for (int k0 = 0; k0 < K; k0 += 64) {
                k_tsgemm_mfma<2, 2><<<blocks, kThreads, 0, s>>>(slabs, A + c0, lda, B + k0, ldb, N, rpb, 64);
                PVS_CHECK_LAUNCH();
                const int rc = pvs_launch_reduce_slabs(s, out + (size_t)c0 * ldo + k0, ldo, 64, slabs, blocks, 4096, accumulate);
                if (rc) return rc;
            }
        return 0;
    }
    if (CK > 32 * kThreads) {
        // wider than one pass holds (32 outputs per thread): column chunks of the right operand, each a
        // product of its own (hidden sizes above 64 on the decomposed layer path; A is re-read per chunk)
        const int kc = (32 * kThreads / C) & ~3;
        PVS_REQUIRE(kc >= 4, "tsgemm: %d x %d outputs unsupported", C, K);
        for (int k0 = 0; k0 < K; k0 += kc) {
            const int rc = pvs_launch_tsgemm_tn(s, out + k0, ldo, A, lda, B + k0, ldb, N, C, K - k0 < kc ? K - k0 : kc,
                                                slabs, accumulate);
            if (rc) return rc;
        }
        return 0;
    }
    const int blocks = pvs_reduce_blocks(N);
    const int rpb = rows_per_block_for(N, blocks);
    if (C % 32 == 0 && K % 32 == 0 && C <= 64 && K <= 64) {
        const int cb = C / 32, kb = K / 32;
        if (cb == 1 && kb == 1) k_tsgemm_mfma<1, 1><<<blocks, kThreads, 0, s>>>(slabs, A, lda, B, ldb, N, rpb, K);
        else if (cb == 1 && kb == 2) k_tsgemm_mfma<1, 2><<<blocks, kThreads, 0, s>>>(slabs, A, lda, B, ldb, N, rpb, K);
        else if (cb == 2 && kb == 1) k_tsgemm_mfma<2, 1><<<blocks, kThreads, 0, s>>>(slabs, A, lda, B, ldb, N, rpb, K);
        else k_tsgemm_mfma<2, 2><<<blocks, kThreads, 0, s>>>(slabs, A, lda, B, ldb, N, rpb, K);
        PVS_CHECK_LAUNCH();
        return pvs_launch_reduce_slabs(s, out, ldo, K, slabs, blocks, CK, accumulate);
    }
    if (C % 32 == 0 && C <= 64 && K < 32 && N >= 1024) {
        // narrow right operand (the input embedding: K = 12 atom features): zero-padded to one
        // 32-column MFMA block; the slabs keep the padded [C][32] layout, the reduction skips the pad
        const int ones = colsum_out ? 1 : 0;      // (column K of the padded right operand = 1: column sums of A for free)
        if (C == 32) k_tsgemm_mfma<1, 1><<<blocks, kThreads, 0, s>>>(slabs, A, lda, B, ldb, N, rpb, K, ones);
        else k_tsgemm_mfma<2, 1><<<blocks, kThreads, 0, s>>>(slabs, A, lda, B, ldb, N, rpb, K, ones);
        PVS_CHECK_LAUNCH();
        k_reduce_slabs<<<(C * 32 + 31) / 32, kThreads, 0, s>>>(out, ldo, 32, slabs, blocks, C * 32, 1.0f,
                                                              accumulate ? 1 : 0, K, colsum_out);
        PVS_CHECK_LAUNCH();
        return 0;
    }
    size_t lds = (size_t)16 * (C + K) * sizeof(float);
    if (CK <= 8 * kThreads)
        k_tsgemm_tn<8><<<blocks, kThreads, lds, s>>>(slabs, A, lda, B, ldb, N, C, K, rpb);
    else
        k_tsgemm_tn<32><<<blocks, kThreads, lds, s>>>(slabs, A, lda, B, ldb, N, C, K, rpb);
    PVS_CHECK_LAUNCH();
    return pvs_launch_reduce_slabs(s, out, ldo, K, slabs, blocks, CK, accumulate);
}

int pvs_node_wgrads_supported(int H) { return H == 32 || H == 64; }

#define PVS_TRY_RC(call) do { int rc_ = (call); if (rc_) return rc_; } while (0)

static int wg_row_blocks(int N) { return (N + kWgRowsPerBlock - 1) / kWgRowsPerBlock; }

size_t pvs_node_wgrads_slab_floats(int N, int H) {
    const int hb = H / 32 > 0 ? H / 32 : 1;
    // per-row-block partials + the reduced sums behind them
    return ((size_t)wg_row_blocks(N) + 1) * hb * hb * kWgSlab;
}

int pvs_launch_node_wgrads(hipStream_t s, int H, int N, const PvsNodeWgradIn& in, const PvsNodeWgradOut& out,
                           float* slabs, bool scatter, const float** gsum_out, const PvsReduce2Args* extra,
                           PvsNodeWgradSlabs* slabs_out, const PvsGhJob* gh_job) {
    PVS_REQUIRE(pvs_node_wgrads_supported(H), "node_wgrads: H = %d unsupported", H);
    PVS_REQUIRE(out.node_w2 && out.node_w1 && out.edge_w1, "node_wgrads: NULL weight gradient");
    const int hb = H / 32, rb = wg_row_blocks(N);
    const int width = hb * hb * kWgSlab;
    float* gsum = slabs + (size_t)rb * width;
    const PvsReduce2Args ex = extra ? *extra : PvsReduce2Args{};
    PvsGhJob gh = gh_job ? *gh_job : PvsGhJob{};
    if (gh_job) {
        PVS_REQUIRE(((uintptr_t)gh.g_h | (uintptr_t)gh.gPQ) % 16 == 0, "node_wgrads: the g_h job needs 16-byte aligned rows");
        const int rows_m = N >= 32768 ? 256 : 128;
        gh.blocks = (N + rows_m - 1) / rows_m;
        if (gh.blocks > 1024) gh.blocks = 1024;
    }
    if (hb == 1) {
        k_node_wgrads<1, true><<<dim3(rb + ex.blocks() + gh.blocks, 1), kThreads, 0, s>>>(slabs, in, N, rb, ex, gh);
    } else {
        if (gh_job)     // (H = 64: the product as its own launch, see k_node_wgrads)
            PVS_TRY_RC(pvs_launch_linear(s, gh.g_h, H, gh.gPQ, 2 * H, gh.W1, 1, gh.ld1, nullptr, gh.gPQ + H, 2 * H,
                                         gh.W1 + gh.off_q, 1, gh.ld1, N, H, H, H, true));
        k_node_wgrads<2, false><<<dim3(rb + ex.blocks(), 4), kThreads, 0, s>>>(slabs, in, N, rb, ex, gh);
    }
    PVS_CHECK_LAUNCH();
    if (slabs_out) {
        slabs_out->slabs = slabs; slabs_out->n_slabs = rb; slabs_out->width = width;
        return 0;
    }
    k_reduce_slabs<<<(width + 31) / 32, kThreads, 0, s>>>(gsum, width, width, slabs, rb, width, 1.0f, 0);
    PVS_CHECK_LAUNCH();
    if (gsum_out) *gsum_out = gsum;
    if (!scatter) return 0;
    k_node_wgrads_scatter<<<hb == 1 ? 4 : 16, kThreads, 0, s>>>(gsum, out, H);
    PVS_CHECK_LAUNCH();
    return 0;
}

int pvs_launch_colreduce(hipStream_t s, int mode, float* out, const float* A, int lda, const float* B,
                         int ldb, const float* shift, int N, int C, float scale, float* slabs,
                         bool accumulate) {
    PVS_REQUIRE(C >= 1, "colreduce: width %d unsupported", C);
    if (C > kThreads) {      // one thread per column: wider inputs in chunks of 256 columns (the slabs are reused in stream order)
        for (int c0 = 0; c0 < C; c0 += kThreads) {
            const int rc = pvs_launch_colreduce(s, mode, out + c0, A + c0, lda, B ? B + c0 : nullptr, ldb,
                                                shift ? shift + c0 : nullptr, N, C - c0 < kThreads ? C - c0 : kThreads,
                                                scale, slabs, accumulate);
            if (rc) return rc;
        }
        return 0;
    }
    const int blocks = pvs_colreduce_blocks(N);
    const int rpb = rows_per_block_for(N, blocks);
    const bool vec4 = C % 4 == 0 && lda % 4 == 0 && (B == nullptr || ldb % 4 == 0) &&
                      (((uintptr_t)A | (uintptr_t)B | (uintptr_t)shift | (uintptr_t)slabs) & 15) == 0;
    if (vec4) k_colreduce4<<<blocks, kThreads, 0, s>>>(mode, slabs, A, lda, B, ldb, shift, N, C, rpb);
    else k_colreduce<<<blocks, kThreads, 0, s>>>(mode, slabs, A, lda, B, ldb, shift, N, C, rpb);
    PVS_CHECK_LAUNCH();
    k_reduce_slabs<<<(C + 31) / 32, kThreads, 0, s>>>(out, C, C, slabs, blocks, C, scale,
                                                      accumulate ? 1 : 0);
    PVS_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
extern "C" int pvs_linear_fwd(const float* x, const float* w, const float* b, float* y, int32_t N,
                              int32_t K, int32_t C, pvs_stream_t stream) {
    return pvs_launch_linear((hipStream_t)stream, y, C, x, K, w, K, 1, b, nullptr, 0, nullptr, 0, 0,
                             N, K, 0, C, false);
}

extern "C" size_t pvs_linear_bwd_workspace_bytes(int32_t N, int32_t K, int32_t C) {
    const size_t kpad = K < 32 ? 32 : K;     // narrow K runs zero-padded to one MFMA block
    return (size_t)pvs_reduce_blocks(N) * (size_t)C * kpad * sizeof(float) + 256;
}

extern "C" int pvs_linear_bwd(const float* x, const float* w, const float* g_y, float* g_x,
                              float* g_w, float* g_b, int32_t N, int32_t K, int32_t C,
                              void* workspace, size_t workspace_bytes, pvs_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    PVS_REQUIRE(workspace_bytes >= pvs_linear_bwd_workspace_bytes(N, K, C) - 256,
                "pvs_linear_bwd: workspace too small");
    float* slabs = (float*)workspace;
    int rc;
    if (g_x) {  // g_x[n,k] = sum_c g_y[n,c] W[c,k]
        rc = pvs_launch_linear(s, g_x, K, g_y, C, w, 1, K, nullptr, nullptr, 0, nullptr, 0, 0, N, C,
                               0, K, false);
        if (rc) return rc;
    }
    const bool bias_in_product = g_w && g_b && pvs_tsgemm_colsum_supported(N, C, K);
    if (g_w) {
        rc = pvs_launch_tsgemm_tn(s, g_w, K, g_y, C, x, K, N, C, K, slabs, false, bias_in_product ? g_b : nullptr);
        if (rc) return rc;
    }
    if (g_b && !bias_in_product) {
        rc = pvs_launch_colreduce(s, PVS_COL_SUM_A, g_b, g_y, C, nullptr, 0, nullptr, N, C, 1.0f,
                                  slabs, false);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int pvs_mean_pool_fwd(const float* h, const int32_t* graph_ptr, float* pooled,
                                 int32_t B, int32_t width, pvs_stream_t stream) {
    PVS_REQUIRE(width >= 1, "mean_pool: width %d unsupported", width);
    if (B <= 0) return 0;
    k_mean_pool_fwd<<<dim3(B, (width + kPoolThreads - 1) / kPoolThreads), kPoolThreads, 0, (hipStream_t)stream>>>(
        h, graph_ptr, pooled, width);
    PVS_CHECK_LAUNCH();
    return 0;
}

extern "C" int pvs_mean_pool_bwd(const float* g_pooled, const int32_t* graph_ptr, float* g_h,
                                 int32_t B, int32_t N, int32_t width, pvs_stream_t stream) {
    if (N <= 0) return 0;
    int blocks = grid_for((long long)N * width, 256);
    k_mean_pool_bwd<<<blocks, 256, 0, (hipStream_t)stream>>>(g_pooled, graph_ptr, g_h, B, N, width);
    PVS_CHECK_LAUNCH();
    return 0;
}

extern "C" int pvs_pool_head_fwd(const float* h, const int32_t* graph_ptr, const float* w, const float* b, float* pooled,
                                 float* y, int32_t B, int32_t width, int32_t n_out, pvs_stream_t stream) {
    PVS_REQUIRE(width >= 1 && width <= kPoolThreads && n_out >= 1, "pool_head: width %d / n_out %d unsupported", width,
                n_out);
    if (B <= 0) return 0;
    k_pool_head_fwd<<<B, kPoolThreads, 0, (hipStream_t)stream>>>(h, graph_ptr, w, b, pooled, y, width, n_out);
    PVS_CHECK_LAUNCH();
    return 0;
}

extern "C" int pvs_pool_head_bwd(const float* g_y, const float* pooled, const float* w, const int32_t* graph_ptr,
                                 float* g_h, float* g_w, float* g_b, int32_t B, int32_t N, int32_t width, int32_t n_out,
                                 pvs_stream_t stream) {
    PVS_REQUIRE(width >= 1 && n_out >= 1 && g_w, "pool_head_bwd: bad arguments");
    const int node_blocks = (g_h && N > 0) ? grid_for((long long)N * ((width & 3) == 0 ? width / 4 : width), 256) : 0;
    const int w_blocks = (n_out * width + 255) / 256;
    k_pool_head_bwd<<<node_blocks + w_blocks, 256, 0, (hipStream_t)stream>>>(g_y, pooled, w, graph_ptr, g_h, g_w, g_b, B,
                                                                           N, width, n_out, node_blocks);
    PVS_CHECK_LAUNCH();
    return 0;
}

extern "C" int pvs_bce_logits_fwd(const float* x, const float* target, int32_t n, float* loss, float* grad,
                                  pvs_stream_t stream) {
    PVS_REQUIRE(n >= 1 && x && target && loss && grad, "bce_logits: bad arguments");
    k_bce_logits_fwd<<<1, 256, 0, (hipStream_t)stream>>>(x, target, n, loss, grad);
    PVS_CHECK_LAUNCH();
    return 0;
}

extern "C" int pvs_scale_by_device_scalar(const float* a, const float* scalar, int32_t n, float* out,
                                          pvs_stream_t stream) {
    if (n <= 0) return 0;
    k_scale_by_device_scalar<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(a, scalar, n, out);
    PVS_CHECK_LAUNCH();
    return 0;
}
