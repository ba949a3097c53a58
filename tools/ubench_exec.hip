// tools/ubench_exec.hip -- does a wave64 VALU instruction cost less when part of the wave is switched off?
// v_fma_f32 / v_mul_lo_u32 / v_sqrt_f32 / v_add_u32 streams (8 independent chains) under eight EXEC patterns.
// Not part of the library.
//   hipcc -O3 --offload-arch=gfx950 -o ab/ubench_exec tools/ubench_exec.hip && ab/ubench_exec
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int UNROLL = 8;

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed, unsigned long long mask)
{
    uint32_t v[UNROLL];
    float f[UNROLL];
    for (int i = 0; i < UNROLL; i++) { v[i] = seed + threadIdx.x * 7919u + i * 104729u; f[i] = 1.0f + 1e-3f * (float)(v[i] & 1023); }
    if ((mask >> (threadIdx.x & 63u)) & 1ull) {
        for (int it = 0; it < ITERS; it++) {
#pragma unroll
            for (int i = 0; i < UNROLL; i++) {
                if (OP == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[i]));
                if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %0" : "+v"(v[i]));
                if (OP == 2) asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[i]));
                if (OP == 3) asm volatile("v_add_u32 %0, %0, %0" : "+v"(v[i]));
            }
        }
    }
    uint32_t acc = 0;
    for (int i = 0; i < UNROLL; i++) acc += v[i] + __float_as_uint(f[i]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int OP>
int run(const char* name, uint32_t* d_out, int waves_per_simd, unsigned long long mask, const char* mname)
{
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 1u, mask);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 2u, mask);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double insts = (double)waves_per_simd * ITERS * UNROLL;
    printf("%-14s waves/SIMD=%d  exec %-16s %.3f ms  %.2f cyc/inst/SIMD @2.4GHz\n", name, waves_per_simd, mname, ms, ms * 1e-3 * 2.4e9 / insts);
    return 0;
}

int main()
{
    uint32_t* d_out; CHECK(hipMalloc(&d_out, 256 * 8 * 256 * sizeof(uint32_t)));
    struct { unsigned long long m; const char* n; } masks[] = {
        {~0ull, "all 64"}, {0xFFFFFFFFull, "lanes 0-31"}, {0xFFFFFFFF00000000ull, "lanes 32-63"}, {0x5555555555555555ull, "even lanes"},
        {0xFFFFull, "lanes 0-15"}, {0x0000FFFF0000FFFFull, "0-15 and 32-47"}, {1ull, "lane 0"}, {0x100000001ull, "lanes 0 and 32"},
    };
    for (int w : {4, 8}) {
        for (auto& m : masks) { run<0>("v_fma_f32", d_out, w, m.m, m.n); }
        for (auto& m : masks) { run<1>("v_mul_lo_u32", d_out, w, m.m, m.n); }
        for (auto& m : masks) { run<2>("v_sqrt_f32", d_out, w, m.m, m.n); }
        for (auto& m : masks) { run<3>("v_add_u32", d_out, w, m.m, m.n); }
    }
    return 0;
}
