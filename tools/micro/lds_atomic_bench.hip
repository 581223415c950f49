// Microbenchmark: LDS atomic-add throughput on gfx950 (one 512-thread block per CU).
// Address pattern = the tiled backward's: lane l adds to row[(l>>5)] * 36 + (l & 31), rows random.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int MODE>
__global__ void __launch_bounds__(512) k(float* out, int iters, const int* rows) {
    __shared__ float acc[128 * 36];
    __shared__ int rb[4096];
    for (int i = threadIdx.x; i < 128 * 36; i += 512) acc[i] = 0.f;
    for (int i = threadIdx.x; i < 4096; i += 512) rb[i] = rows[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int j = lane & 31, hh = lane >> 5;
    float v = 1.0f + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int r = rb[(it * 32 + 2 * s + hh + wv * 512) & 4095];
            float* p = acc + r * 36 + j;
            if (MODE == 0) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (MODE == 1) __hip_atomic_fetch_add((int*)p, (int)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (MODE == 2) { *p = *p + v; }                       // racy RMW: rate reference only
            else if (MODE == 3) {                                      // 16 lanes x float4? no: plain store
                *p = v;
            } else if (MODE == 4) {                                    // one row per instruction (32 lanes active)
                if (hh == 0) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else if (MODE == 5) {                                    // distinct addresses, all 64 lanes, no row indirection
                __hip_atomic_fetch_add(acc + ((s * 64 + lane + it) & 4095), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    __syncthreads();
    float t = 0.f;
    for (int i = threadIdx.x; i < 128 * 36; i += 512) t += acc[i];
    out[blockIdx.x * 512 + threadIdx.x] = t;
}

template <int MODE>
void run(const char* name, float* out, const int* rows) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2000;
    k<MODE><<<256, 512>>>(out, 10, rows);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<MODE><<<256, 512>>>(out, iters, rows);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    // per CU: 8 waves x iters x 16 instructions
    const double instr = 8.0 * iters * 16;
    printf("%-28s %.3f ms  -> %.1f ns per wave-instruction per CU (%.1f cycles @2.4GHz)\n", name, ms,
           ms * 1e6 / instr, ms * 1e6 / instr * 2.4);
}

int main() {
    float* out; int* rows;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&rows, 4096 * 4);
    int h[4096];
    srand(1);
    for (int i = 0; i < 4096; ++i) h[i] = rand() % 128;
    hipMemcpy(rows, h, sizeof(h), hipMemcpyHostToDevice);
    run<0>("ds_add_f32 2rows/instr", out, rows);
    run<1>("ds_add_u32 2rows/instr", out, rows);
    run<2>("plain read+add+write", out, rows);
    run<3>("plain write", out, rows);
    run<4>("ds_add_f32 1row/instr", out, rows);
    run<5>("ds_add_f32 distinct addrs", out, rows);
    return 0;
}
