// Microbenchmark: what one vector instruction costs a SIMD on gfx950, by instruction and by waves per SIMD
// (profiles/r03_micro_valu_issue.txt). It prices the VALU stream of the edge kernels: the floor of a kernel that
// is bound by vector issue is the sum of these costs over its instructions, not "N instructions x 4 cycles".
// One workgroup per CU (100 KB of LDS), W waves per SIMD, every wave runs 8 independent dependency chains of
// the instruction. Reported: SIMD cycles per wave-instruction = wall time x in-kernel clock / instructions per
// SIMD, with the clock from s_memtime / s_memrealtime (100 MHz) around the loop of wave 0 (median over blocks).
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/valu_issue_bench.hip -o tools/micro/valu_issue_bench.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum {
    FMA, FMAC, ADD, MUL, SUB, MAXF, MAX3, AND, MOV, CNDMASK, LDEXP, PKFMA, PKMUL, PKADD, EXP, RCP, MIXLO, MIXF32,
    CVTPK, CVTPKRTZ, CVTF32F16, CVTF32F16HI, CVTPKBF16, PERM, MOVDPP, PKFMA16, FMA2SRC, CMPCND, ADDU32, LSHLADD, LSHLADD64, MADU64,
    SILU, SPLIT_MIX, SPLIT_CVT, SPLIT_NEW, SIGPAIR_2RCP, SIGPAIR_1RCP,
    FMA_MFMA, PKFMA_MFMA, ADD_MFMA, EXP_MFMA, NMODES
};
const char* kNames[NMODES] = {
    "v_fma_f32", "v_fmac_f32_e32", "v_add_f32_e32", "v_mul_f32_e32", "v_sub_f32_e32", "v_max_f32_e32", "v_max3_f32",
    "v_and_b32_e32", "v_mov_b32", "v_cndmask_b32_e32", "v_ldexp_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32",
    "v_exp_f32", "v_rcp_f32", "v_fma_mixlo_f16", "v_fma_mix_f32", "v_cvt_pk_f16_f32", "v_cvt_pkrtz_f16_f32",
    "v_cvt_f32_f16", "v_cvt_f32_f16 sdwa hi", "v_cvt_pk_bf16_f32", "v_perm_b32", "v_mov_b32 dpp", "v_pk_fma_f16",
    "v_fma_f32 (two distinct sources)", "v_cmp_lt_f32 + v_cndmask_b32 (vcc) /2", "v_add_u32_e32", "v_lshl_add_u32", "v_lshl_add_u64", "v_mad_u64_u32",
    "SiLU (mul exp add rcp mul) /5", "split pair, 4 fma_mix /4", "split pair, mul mul cvt_pk cvt cvt sub sub cvt_pk /8",
    "split pair, mul mul cvt_pk mix_f32 mix_f32 cvt_pk /6",
    "two sigmoids, 2 x (mul exp add rcp): cycles per PAIR", "two sigmoids, ONE rcp: r = 1/(ta tb), sa = r tb, sb = r ta: per PAIR",
    "8 v_fma_f32 + 1 mfma /8", "8 v_pk_fma_f32 + 1 mfma /8", "8 v_add_f32 + 1 mfma /8", "8 v_exp_f32 + 1 mfma /8"};
// instructions per chain step (the sequences count all their instructions)
const int kPerStep[NMODES] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                               1, 2, 1, 1, 1, 1, 5, 4, 8, 6, 1, 1, 1, 1, 1, 1};

template <int MODE, int NT>
__global__ void __launch_bounds__(NT) k(long long* cyc, float* out, int iters) {
    extern __shared__ float pad[];
    if (threadIdx.x == 0) pad[0] = 0.f;
    float a[8], b[8];
    f32x2 p[8];
    unsigned h[8], l[8];
    unsigned long long q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = 1.0f + threadIdx.x * 1e-3f + i;
        b[i] = 0.5f + i;
        p[i] = f32x2{a[i], a[i] + 1.f};
        h[i] = threadIdx.x + i;
        l[i] = i;
        q[i] = threadIdx.x * 8ull + i;
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    f16x8 fa, fb;
#pragma unroll
    for (int r = 0; r < 8; ++r) { fa[r] = (_Float16)(threadIdx.x & 7); fb[r] = (_Float16)1; }
    const float c0 = 0.999f, c1 = 1e-4f;
    const int ci = 1;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    const long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE >= FMA_MFMA) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == FMA || MODE == FMA_MFMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c0), "v"(c1));
                if (MODE == FMAC) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "v"(c0), "v"(c1));
                if (MODE == ADD || MODE == ADD_MFMA) asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
                if (MODE == MUL) asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(c0));
                if (MODE == SUB) asm volatile("v_sub_f32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
                if (MODE == MAXF) asm volatile("v_max_f32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
                if (MODE == MAX3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c0), "v"(c1));
                if (MODE == AND) asm volatile("v_and_b32_e32 %0, %0, %1" : "+v"(h[i]) : "v"(0xffffe000u));
                if (MODE == MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(h[i]) : "v"(l[i]));
                if (MODE == CNDMASK) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(h[i]) : "v"(l[i]));
                if (MODE == LDEXP) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(a[i]) : "v"(ci));
                if (MODE == PKFMA || MODE == PKFMA_MFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
                if (MODE == PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
                if (MODE == PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
                if (MODE == EXP || MODE == EXP_MFMA) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                if (MODE == RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
                if (MODE == MIXLO) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(h[i]) : "v"(a[i]), "v"(c0));
                if (MODE == MIXF32) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(h[i]), "v"(c0));
                if (MODE == CVTPK) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h[i]) : "v"(a[i]), "v"(b[i]));
                if (MODE == CVTPKRTZ) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(h[i]) : "v"(a[i]), "v"(b[i]));
                if (MODE == CVTF32F16) asm volatile("v_cvt_f32_f16_e32 %0, %1" : "=v"(a[i]) : "v"(h[i]));
                if (MODE == CVTF32F16HI) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(a[i]) : "v"(h[i]));
                if (MODE == CVTPKBF16) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h[i]) : "v"(a[i]), "v"(b[i]));
                if (MODE == PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(h[i]) : "v"(h[(i + 1) & 7]), "v"(0x07060302u));
                if (MODE == MOVDPP) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(h[i]));
                if (MODE == PKFMA16) asm volatile("v_pk_fma_f16 %0, %0, %1, %1" : "+v"(h[i]) : "v"(l[i]));
                if (MODE == FMA2SRC) asm volatile("v_fma_f32 %0, %0, %0, %1" : "+v"(a[i]) : "v"(c1));
                if (MODE == CMPCND) {
                    asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" : : "v"(a[i]), "v"(c0) : "vcc");
                    asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(h[i]) : "v"(l[i]) : "vcc");
                }
                if (MODE == ADDU32) asm volatile("v_add_u32_e32 %0, %0, %1" : "+v"(h[i]) : "v"(l[i]));
                if (MODE == LSHLADD) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(h[i]) : "v"(l[i]));
                if (MODE == LSHLADD64) asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(q[i]) : "v"(q[(i + 1) & 7]));
                if (MODE == MADU64) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(h[i]), "v"(l[i]) : "vcc");
                if (MODE == SPLIT_NEW) {
                    float x0, x1, w0, w1;
                    asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(x0) : "v"(c0), "v"(a[i]));
                    asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(x1) : "v"(c0), "v"(b[i]));
                    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h[i]) : "v"(x0), "v"(x1));
                    asm volatile("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(w0) : "v"(x0), "v"(h[i]));
                    asm volatile("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(w1) : "v"(x1), "v"(h[i]));
                    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(l[i]) : "v"(w0), "v"(w1));
                }
                if (MODE == SIGPAIR_2RCP || MODE == SIGPAIR_1RCP) {   // (a, b) <- (sigmoid(a), sigmoid(b))
                    float ta, tb;
                    asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(ta) : "v"(-1.4426950408889634f), "v"(a[i]));
                    asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(tb) : "v"(-1.4426950408889634f), "v"(b[i]));
                    asm volatile("v_exp_f32 %0, %0" : "+v"(ta));
                    asm volatile("v_exp_f32 %0, %0" : "+v"(tb));
                    asm volatile("v_add_f32_e32 %0, 1.0, %0" : "+v"(ta));
                    asm volatile("v_add_f32_e32 %0, 1.0, %0" : "+v"(tb));
                    if (MODE == SIGPAIR_2RCP) {
                        asm volatile("v_rcp_f32 %0, %1" : "=v"(a[i]) : "v"(ta));
                        asm volatile("v_rcp_f32 %0, %1" : "=v"(b[i]) : "v"(tb));
                    } else {
                        float r;
                        asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(r) : "v"(ta), "v"(tb));
                        asm volatile("v_rcp_f32 %0, %0" : "+v"(r));
                        asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(a[i]) : "v"(r), "v"(tb));
                        asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(b[i]) : "v"(r), "v"(ta));
                    }
                }
                if (MODE == SILU) {   // a <- a * sigmoid(a): mul, exp, add, rcp, mul
                    float t;
                    asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(t) : "v"(-1.4426950408889634f), "v"(a[i]));
                    asm volatile("v_exp_f32 %0, %0" : "+v"(t));
                    asm volatile("v_add_f32_e32 %0, 1.0, %0" : "+v"(t));
                    asm volatile("v_rcp_f32 %0, %0" : "+v"(t));
                    asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(t));
                }
                if (MODE == SPLIT_MIX) {   // the shipped split of a pair (edge_mfma_common.h: pvs_f16_hi2 / pvs_f16_lo2)
                    asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h[i]) : "v"(a[i]), "v"(c0));
                    asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h[i]) : "v"(b[i]), "v"(c0));
                    asm volatile("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l[i]) : "v"(a[i]), "v"(c0), "v"(h[i]));
                    asm volatile("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l[i]) : "v"(b[i]), "v"(c0), "v"(h[i]));
                }
                if (MODE == SPLIT_CVT) {   // candidate: scale, RNE pack, widen, subtract, RNE pack
                    float x0, x1, w0, w1;
                    asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(x0) : "v"(c0), "v"(a[i]));
                    asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(x1) : "v"(c0), "v"(b[i]));
                    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h[i]) : "v"(x0), "v"(x1));
                    asm volatile("v_cvt_f32_f16_e32 %0, %1" : "=v"(w0) : "v"(h[i]));
                    asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(w1) : "v"(h[i]));
                    asm volatile("v_sub_f32_e32 %0, %0, %1" : "+v"(x0) : "v"(w0));
                    asm volatile("v_sub_f32_e32 %0, %0, %1" : "+v"(x1) : "v"(w1));
                    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(l[i]) : "v"(x0), "v"(x1));
                }
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    const long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + b[i] + p[i][0] + p[i][1] + (float)h[i] + (float)l[i] + (float)q[i];
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * NT + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int MODE, int NT>
void run(long long* cyc, float* out) {
    const int iters = 3000;
    (void)hipFuncSetAttribute((const void*)k<MODE, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    k<MODE, NT><<<256, NT, 100 * 1024>>>(cyc, out, 50);
    (void)hipDeviceSynchronize();
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a);
    k<MODE, NT><<<256, NT, 100 * 1024>>>(cyc, out, iters);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    std::vector<long long> c(512);
    (void)hipMemcpy(c.data(), cyc, 512 * sizeof(long long), hipMemcpyDeviceToHost);
    std::vector<double> clk;
    for (int i = 0; i < 256; ++i) clk.push_back((double)c[2 * i] / (double)c[2 * i + 1] * 0.1);   // GHz
    std::sort(clk.begin(), clk.end());
    const double ghz = clk[128];
    const int w = NT / 256;                                              // waves per SIMD
    const double instr_per_simd = (double)w * iters * 32.0 * kPerStep[MODE];
    printf("%-58s %d wave(s)/SIMD: %6.2f SIMD cycles per instruction   (%.3f ms, %.2f GHz)\n", kNames[MODE], w,
           ms * 1e6 * ghz / instr_per_simd, ms, ghz);
}

template <int MODE>
void all_w(long long* cyc, float* out) {
    run<MODE, 256>(cyc, out);
    run<MODE, 512>(cyc, out);
    run<MODE, 1024>(cyc, out);
}

template <int MODE>
struct Sweep {
    static void go(long long* cyc, float* out) {
        all_w<MODE>(cyc, out);
        Sweep<MODE + 1>::go(cyc, out);
    }
};
template <>
struct Sweep<NMODES> {
    static void go(long long*, float*) {}
};

int main(int argc, char** argv) {
    long long* cyc; float* out;
    (void)hipMalloc(&cyc, 512 * sizeof(long long));
    (void)hipMalloc(&out, 256 * 1024 * sizeof(float));
    if (argc > 1 && argv[1][0] == 's') {      // "sigmoid": only the rows a sigmoid is made of
        all_w<EXP>(cyc, out); all_w<RCP>(cyc, out); all_w<SILU>(cyc, out);
        all_w<SIGPAIR_2RCP>(cyc, out); all_w<SIGPAIR_1RCP>(cyc, out);
        return 0;
    }
    Sweep<0>::go(cyc, out);
    return 0;
}
