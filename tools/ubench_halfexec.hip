// tools/ubench_halfexec.hip -- does a wave64 VALU instruction with EXEC[63:32] == 0 (or EXEC[31:0] == 0)
// issue faster on CDNA4's SIMD32?  Times an FMA chain with different EXEC masks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITERS = 4096, UNROLL = 8;

__global__ __launch_bounds__(256) void k(float* out, unsigned long long mask)
{
    float f[UNROLL];
    for (int i = 0; i < UNROLL; i++) f[i] = 1.0f + 1e-3f * (float)((threadIdx.x * 7 + i) & 1023);
    const bool on = (mask >> (threadIdx.x & 63)) & 1ull;
    if (on) {
        for (int it = 0; it < ITERS; it++) {
#pragma unroll
            for (int i = 0; i < UNROLL; i++) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[i]) : "v"(f[(i + 1) % UNROLL]));
        }
    }
    float acc = 0; for (int i = 0; i < UNROLL; i++) acc += f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main()
{
    float* d; CHECK(hipMalloc(&d, 256 * 8 * 256 * sizeof(float)));
    const unsigned long long masks[] = { ~0ull, 0x00000000FFFFFFFFull, 0xFFFFFFFF00000000ull, 0x5555555555555555ull, 0x000000000000FFFFull, 0x0000FFFF0000FFFFull };
    const char* names[] = { "all 64", "low 32", "high 32", "even lanes", "low 16", "16+16" };
    for (int w : {2, 4, 8}) for (int m = 0; m < 6; m++) {
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k, dim3(256 * w), dim3(256), 0, 0, d, masks[m]); CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(k, dim3(256 * w), dim3(256), 0, 0, d, masks[m]); CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("waves/SIMD=%d  exec=%-10s %.3f ms  %.2f cyc/inst/SIMD @2.4GHz\n", w, names[m], ms, ms * 1e-3 * 2.4e9 / ((double)w * ITERS * UNROLL));
    }
    return 0;
}
