// Shared device helpers of the fp32 MFMA kernels (v_mfma_f32_32x32x2_f32, "X layout").
//
// X layout = the accumulator layout of the instruction: lane l = (column j = l&31, half hh = l>>5),
// register t of 32-channel block b holds channel 32b + (t&3) + 8(t>>2) + 4hh. A tensor in X layout
// is directly the B operand of the next product over channels (cdna_hip_programming.md §3;
// validated lane-by-lane in tools/mfma_layout_check.py).
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// timing-only ablation (tools/variant_obj.sh dense_ops.hip "-DPVS_ABL_NODE_MFMA_STEP=3"): every third fp32 MFMA of the
// node-level products only - what a faster product would buy (results are wrong in such a build)
#ifndef PVS_ABL_NODE_MFMA_STEP
#define PVS_ABL_NODE_MFMA_STEP 1
#endif

__device__ __forceinline__ int xch(int t, int hh) { return (t & 3) + 8 * (t >> 2) + 4 * hh; }

// rows of a row-major [rows][ld] array <-> X layout: lane reads/writes 4 floats at 32b + 8g + 4hh
template <int HB>
__device__ __forceinline__ void load_x(const float* __restrict__ base, int hh, float (&out)[HB][16]) {
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

// Natural row-major staging Wn[c*(K+1) + k] (row stride padded by one word): ONE copy serves both
// Z = W V (lanes vary the row: stride K+1 -> distinct banks) and Z = W^T V (lanes vary the column:
// consecutive words).
template <int CB, int KB, bool TRANSPOSE>
__device__ __forceinline__ void mfma_chain_rect(const float* __restrict__ Wn, int lane,
                                                const float (&v)[TRANSPOSE ? CB : KB][16],
                                                f32x16 (&acc)[TRANSPOSE ? KB : CB]) {
    // Wn is [32*CB rows][32*KB + 1]; !TRANSPOSE: acc[c-block] += W[c][k] v[k-block]
    //                                 TRANSPOSE: acc[k-block] += W[c][k] v[c-block]
    constexpr int LD = 32 * KB + 1;
    const int j = lane & 31, hh = lane >> 5;
    if (!TRANSPOSE) {
        const float* base = Wn + j * LD + 4 * hh;
#pragma unroll
        for (int bo = 0; bo < CB; ++bo)
#pragma unroll
            for (int bi = 0; bi < KB; ++bi)
#pragma unroll
                for (int t = 0; t < 16; t += PVS_ABL_NODE_MFMA_STEP) {
                    const float a = base[(32 * bo) * LD + 32 * bi + (t & 3) + 8 * (t >> 2)];
                    acc[bo] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, v[bi][t], acc[bo], 0, 0, 0);
                }
    } else {
        const float* base = Wn + (4 * hh) * LD + j;
#pragma unroll
        for (int bo = 0; bo < KB; ++bo)
#pragma unroll
            for (int bi = 0; bi < CB; ++bi)
#pragma unroll
                for (int t = 0; t < 16; t += PVS_ABL_NODE_MFMA_STEP) {
                    const float a = base[(32 * bi + (t & 3) + 8 * (t >> 2)) * LD + 32 * bo];
                    acc[bo] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, v[bi][t], acc[bo], 0, 0, 0);
                }
    }
}
