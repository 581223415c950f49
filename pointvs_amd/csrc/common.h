// Shared host/device helpers for libpvs_egnn.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/pvs_egnn.h"

#define PVS_WAVE 64

void pvs_set_error(const char* fmt, ...);

#define PVS_CHECK_HIP(expr)                                                              \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            pvs_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                          __LINE__);                                                     \
            return -2;                                                                   \
        }                                                                                \
    } while (0)

#define PVS_CHECK_LAUNCH() PVS_CHECK_HIP(hipGetLastError())

#define PVS_REQUIRE(cond, ...)                \
    do {                                      \
        if (!(cond)) {                        \
            pvs_set_error(__VA_ARGS__);       \
            return -1;                        \
        }                                     \
    } while (0)

static inline size_t pvs_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// 16-byte store / load of data that is written once and read once by a later kernel (per-edge
// gradients): non-temporal, so the stream does not displace the gathered node rows in L2
// (measured: cfg2 backward + column gather -0.16 ms/step).
typedef float pvs_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void pvs_store_nt(float* p, const float4& v) {
    __builtin_nontemporal_store(pvs_f4{v.x, v.y, v.z, v.w}, reinterpret_cast<pvs_f4*>(p));
}
__device__ __forceinline__ float4 pvs_load_nt(const float* p) {
    const pvs_f4 v = __builtin_nontemporal_load(reinterpret_cast<const pvs_f4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}

// Bump allocator over the caller's workspace (256-B aligned pieces).
struct PvsArena {
    char* base;
    size_t cap;
    size_t off;
    PvsArena(void* p, size_t bytes) : base((char*)p), cap(bytes), off(0) {}
    template <typename T>
    T* take(size_t count) {
        off = pvs_align_up(off, 256);
        T* r = (T*)(base ? base + off : nullptr);
        off += count * sizeof(T);
        return r;
    }
    bool ok() const { return off <= cap; }
};

// ---- device math: SiLU / sigmoid and their derivatives ----
// One v_exp_f32 + one v_rcp_f32 (1 ulp each) per sigmoid: an IEEE division costs ~10 VALU
// instructions and the edge kernels evaluate ~100 sigmoids per edge.
__device__ __forceinline__ float pvs_rcp(float v) { return __builtin_amdgcn_rcpf(v); }
__device__ __forceinline__ float pvs_exp(float v) { return __builtin_amdgcn_exp2f(v * 1.4426950408889634f); }
#ifdef PVS_ABL_NO_SILU   // timing-only build: no transcendental in the activations
__device__ __forceinline__ float pvs_sigmoid(float v) { return fmaf(0.1f, v, 0.5f); }
#else
__device__ __forceinline__ float pvs_sigmoid(float v) { return pvs_rcp(1.0f + pvs_exp(-v)); }
#endif
__device__ __forceinline__ float pvs_silu(float v) { return v * pvs_sigmoid(v); }
// The same on PAIRS of values in adjacent registers (round 5). A wave issues at most one instruction per ~5 cycles
// whatever it is (profiles/r03_micro_valu_issue.txt: "a wave alone"), and the two-waves-per-SIMD kernels are bound by
// that, not by the ALU: a v_pk_*_f32 produces two results in one issue slot. The compiler's SLP pass packs some of
// this by itself but leaves the multiply by a literal (VOP3P takes no literal) and the add behind the scalar
// v_exp_f32 results unpacked; written on two-element vectors they are packed. Same operations, same roundings.
typedef float pvs_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pvs_f2 pvs_fma2(pvs_f2 a, pvs_f2 b, pvs_f2 c) { return __builtin_elementwise_fma(a, b, c); }
#ifdef PVS_ABL_NO_SILU
__device__ __forceinline__ pvs_f2 pvs_sigmoid2(pvs_f2 v) { return pvs_fma2(v, pvs_f2{0.1f, 0.1f}, pvs_f2{0.5f, 0.5f}); }
#else
__device__ __forceinline__ pvs_f2 pvs_sigmoid2(pvs_f2 v) {
    pvs_f2 t = v * -1.4426950408889634f;
    t = pvs_f2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + 1.0f;
    return pvs_f2{pvs_rcp(t.x), pvs_rcp(t.y)};
}
#endif
__device__ __forceinline__ pvs_f2 pvs_silu2(pvs_f2 z) { return z * pvs_sigmoid2(z); }
// a = SiLU(z), d = SiLU'(z) = s + z s (1 - s)
__device__ __forceinline__ void pvs_silu_grad2(pvs_f2 z, pvs_f2& a, pvs_f2& d) {
    const pvs_f2 sg = pvs_sigmoid2(z);
    a = z * sg;
    d = pvs_fma2(a, 1.0f - sg, sg);
}
// d/dv [v*sigmoid(v)] given s = sigmoid(v)
__device__ __forceinline__ float pvs_silu_grad(float v, float s) { return s * (1.0f + v * (1.0f - s)); }
__device__ __forceinline__ float pvs_tanh(float v) {
    // |v| < 0.35: odd series (relative accuracy for tiny arguments: coord_mlp's last layer is
    // initialised with gain 1e-3); else 1 - 2/(exp(2|v|)+1) with the sign restored.
    const float a = fabsf(v);
    if (a < 0.35f) {
        const float v2 = v * v;
        return v * (1.0f + v2 * (-0.33333334f + v2 * (0.13333334f + v2 * (-0.053968254f + v2 * 0.021869488f))));
    }
    const float e = pvs_exp(2.0f * a);
    return copysignf(1.0f - 2.0f * pvs_rcp(e + 1.0f), v);
}

// attention activation (PVS_ACT_*) and derivative wrt its logit, given logit l and value a
__device__ __forceinline__ float pvs_att_act(int act, float l) {
    switch (act) {
        case PVS_ACT_SIGMOID: return pvs_sigmoid(l);
        case PVS_ACT_TANH: return pvs_tanh(l);
        case PVS_ACT_RELU: return l > 0.f ? l : 0.f;
        case PVS_ACT_SILU: return pvs_silu(l);
        default: return l;
    }
}
__device__ __forceinline__ float pvs_att_act_grad(int act, float l, float a) {
    switch (act) {
        case PVS_ACT_SIGMOID: return a * (1.0f - a);
        case PVS_ACT_TANH: return 1.0f - a * a;
        case PVS_ACT_RELU: return l > 0.f ? 1.0f : 0.f;
        case PVS_ACT_SILU: { float s = pvs_sigmoid(l); return pvs_silu_grad(l, s); }
        default: return 1.0f;
    }
}

// rho (>= 0) and the edge class (< 4) share one float of the per-edge backward record: rho is
// rounded to 22 mantissa bits (relative 1.2e-7, it only weights the w_rho gradient), the class
// sits in the two low bits, so the column gather needs no separate 1-byte type gather.
__device__ __forceinline__ float pvs_pack_rho_type(float rho, int ty) {
    return __uint_as_float(((__float_as_uint(rho) + 2u) & ~3u) | (unsigned)(ty & 3));
}
__device__ __forceinline__ float pvs_unpack_rho(float packed, int* ty) {
    const unsigned u = __float_as_uint(packed);
    *ty = (int)(u & 3u);
    return __uint_as_float(u & ~3u);
}

// XCD-aware block order: consecutive hardware block ids are dealt round-robin over the 8 XCDs
// (MI355X_MICROARCH.md, workgroup dispatch), so blocks b and b+8 share an L2. Give each XCD a
// CONTIGUOUS range of work items (graphs): the node rows an XCD gathers then fit its 4 MB L2 instead
// of every L2 seeing every graph. Speed only - any placement is correct.
__device__ __forceinline__ int pvs_xcd_block(int b, int nb) {
    return (nb & 7) ? b : (b & 7) * (nb >> 3) + (b >> 3);
}

// Same-wave LDS hand-off: order this wave's LDS writes before its later LDS reads. LDS
// instructions of one wave execute in issue order, so only the COMPILER must be kept from
// reordering them: wavefront-scope fences emit no instruction. (A workgroup-scope release would
// also drain every outstanding global load/store of the wave - s_waitcnt vmcnt(0) - at each
// hand-off, serialising the gathers behind the reductions.)
__device__ __forceinline__ void pvs_wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Sum over aligned groups of W consecutive lanes (W power of two <= 64); every lane gets the total.
template <int W>
__device__ __forceinline__ float pvs_group_sum(float v) {
#pragma unroll
    for (int o = W / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
