// tools/ubench_valu.hip -- issue cost of the VALU / LDS instructions the render kernel leans on,
// measured on the box (cycles per wave64 instruction per SIMD).  Not part of the library.
//   hipcc -O3 --offload-arch=gfx950 -o build/ubench_valu tools/ubench_valu.hip && build/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int UNROLL = 8;   // independent chains

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed)
{
    uint32_t v[UNROLL];
    float f[UNROLL];
    for (int i = 0; i < UNROLL; i++) { v[i] = seed + threadIdx.x * 7919u + i * 104729u; f[i] = 1.0f + 1e-3f * (float)(v[i] & 1023); }
    __shared__ float4 lds[64];
    if (threadIdx.x < 64) lds[threadIdx.x] = make_float4(1.f, 2.f, 3.f, 4.f);
    __syncthreads();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < UNROLL; i++) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[i]) : "v"(f[(i + 1) % UNROLL]));
            if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) % UNROLL]));
            if (OP == 2) { uint64_t t; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "=v"(t) : "v"(v[i]), "v"(v[(i + 1) % UNROLL]) : "vcc"); v[i] = (uint32_t)t; }
            if (OP == 3) asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[i]));
            if (OP == 4) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));
            if (OP == 5) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) % UNROLL]));
            if (OP == 6) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(f[i]) : "v"(v[i]));
            if (OP == 7) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) % UNROLL]));
            if (OP == 8) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) % UNROLL]));
            if (OP == 9) asm volatile("v_rndne_f32 %0, %0" : "+v"(f[i]));
            if (OP == 10) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(v[(i + 1) % UNROLL]) : "vcc");
            if (OP == 11) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(*(double*)&v[i & ~1]) : "v"(*(double*)&f[i & ~1]));
            if (OP == 12) { float4 t; asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(t) : "v"((uint32_t)((it & 63) * 16)) : "memory"); f[i] += t.x; }
            if (OP == 13) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) % UNROLL]));
            if (OP == 14) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(v[i]) : "v"(v[(i + 1) % UNROLL]));
            if (OP == 15) asm volatile("v_exp_f32 %0, %0" : "+v"(f[i]));
        }
    }
    uint32_t acc = 0;
    for (int i = 0; i < UNROLL; i++) acc += v[i] + __float_as_uint(f[i]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int OP>
int run(const char* name, uint32_t* d_out, int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd;   // 256 CUs x (4 waves per block = 1 per SIMD)
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 1u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 2u);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    // instructions per SIMD = waves_per_simd * ITERS * UNROLL ; assume 2.4 GHz
    double insts = (double)waves_per_simd * ITERS * UNROLL;
    printf("%-16s waves/SIMD=%d  %.3f ms  %.2f cyc/inst/SIMD @2.4GHz\n", name, waves_per_simd, ms, ms * 1e-3 * 2.4e9 / insts);
    return 0;
}

int main()
{
    uint32_t* d_out; CHECK(hipMalloc(&d_out, 256 * 8 * 256 * sizeof(uint32_t)));
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32", d_out, w);
        run<1>("v_mul_lo_u32", d_out, w);
        run<2>("v_mad_u64_u32", d_out, w);
        run<13>("v_mul_hi_u32", d_out, w);
        run<7>("v_mul_u32_u24", d_out, w);
        run<3>("v_sqrt_f32", d_out, w);
        run<4>("v_rcp_f32", d_out, w);
        run<15>("v_exp_f32", d_out, w);
        run<5>("v_add_u32", d_out, w);
        run<14>("v_lshl_add_u32", d_out, w);
        run<8>("v_xor_b32", d_out, w);
        run<6>("v_cvt_f32_u32", d_out, w);
        run<9>("v_rndne_f32", d_out, w);
        run<10>("v_cndmask_b32", d_out, w);
        run<11>("v_pk_fma_f32", d_out, w);
        run<12>("ds_read_b128 bc", d_out, w);
    }
    return 0;
}
