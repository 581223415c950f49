// Where does `global_load_lds_dwordx4 v, off offset:N` put its data? (MI355X) The LDS destination of an LDS-DMA is
// M0 + lane * size; this checks whether the instruction's immediate offset moves the LDS address as well as the global
// one. Build: hipcc -O2 --offload-arch=gfx950 tools/micro/glds_offset_test.hip -o /tmp/glds_offset_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k(const float* src, float* out) {
    extern __shared__ float smem[];
    for (int i = threadIdx.x; i < 4096; i += 64) smem[i] = -1.f;
    __syncthreads();
    const float* g = src + threadIdx.x * 64;              // lane l: floats 64 l ...
    const unsigned base = (unsigned)(unsigned long long)smem;
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"             // floats 64 l + 0..3   -> LDS base + 16 l
        "s_add_i32 m0, %2, 4096\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off offset:32\n\t"   // floats 64 l + 8..11  -> LDS base + 4096 (+ 32 ?) + 16 l
        "s_add_i32 m0, %2, 8192\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx3 %1, off offset:64\n\t"   // floats 64 l + 16..18 -> LDS base + 8192 (+ 64 ?) + 12 l
        "s_mov_b32 m0, %0\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&s"(keep) : "v"(g), "s"(base) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += 64) out[i] = smem[i];
}

int main() {
    std::vector<float> h(64 * 64);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)i;
    float *src, *out;
    hipMalloc(&src, h.size() * 4);
    hipMalloc(&out, 4096 * 4);
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    k<<<1, 64, 4096 * 4>>>(src, out);
    std::vector<float> o(4096);
    hipMemcpy(o.data(), out, 4096 * 4, hipMemcpyDeviceToHost);
    auto find = [&](float v) { for (int i = 0; i < 4096; ++i) if (o[i] == v) return i * 4; return -1; };
    printf("x4 offset:0   lane 0 float 0   at LDS byte %d (expect 0),  lane 1 float 64 at %d (expect 16)\n", find(0.f), find(64.f));
    printf("x4 offset:32  lane 0 float 8   at LDS byte %d (M0 = 4096: 4096 if the offset is global-only, 4128 if it moves LDS too), lane 1 float 72 at %d\n", find(8.f), find(72.f));
    printf("x3 offset:64  lane 0 float 16  at LDS byte %d (M0 = 8192), lane 1 float 80 at %d (lane stride 12?), lane 0 float 18 at %d, float 19 (must be absent) at %d\n",
           find(16.f), find(80.f), find(18.f), find(19.f));
    return 0;
}
