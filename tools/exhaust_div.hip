// tools/exhaust_div.hip -- exhaustive check, on the GPU, of short correctly-rounded DIVISION
// sequences against hipcc's own correctly rounded a / b.  For operands whose quotient, residuals
// and reciprocal stay in the normal range the significand of RN(a / b) depends only on the two
// significands, so sweeping every pair of 23-bit significands (2^46 pairs, a and b in [1, 2))
// covers every such division; a second pass samples other binades at random to confirm that the
// exponents do not matter.
//   hipcc -O3 -ffp-contract=off --offload-arch=gfx950 tools/exhaust_div.hip -o tools/exhaust_div.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define DEV __device__ __forceinline__
DEV float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// D6: correctly rounded reciprocal (rcp + one Newton step), quotient, one residual correction
DEV float div_d6(float a, float b)
{
    const float y0 = __builtin_amdgcn_rcpf(b);
    const float y = fma_(fma_(-b, y0, 1.0f), y0, y0);
    const float q = a * y;
    return fma_(fma_(-b, q, a), y, q);
}
// D8: the core of hipcc's expansion without v_div_scale / v_div_fixup
DEV float div_d8(float a, float b)
{
    const float y0 = __builtin_amdgcn_rcpf(b);
    const float y = fma_(fma_(-b, y0, 1.0f), y0, y0);
    float q = a * y;
    q = fma_(fma_(-b, q, a), y, q);
    return fma_(fma_(-b, q, a), y, q);
}

struct Tally { unsigned long long bad6, bad8; uint32_t a6, b6, a8, b8; };

// every b significand in [b_lo, b_hi) against every a significand
__global__ void sweep(uint32_t b_lo, uint32_t b_hi, uint32_t a_exp, uint32_t b_exp, Tally* t)
{
    unsigned long long bad6 = 0, bad8 = 0;
    for (uint32_t mb = b_lo + blockIdx.y; mb < b_hi; mb += gridDim.y) {
        const float b = __builtin_bit_cast(float, (b_exp << 23) | mb);
        for (uint32_t ma = blockIdx.x * blockDim.x + threadIdx.x; ma < (1u << 23); ma += gridDim.x * blockDim.x) {
            const float a = __builtin_bit_cast(float, (a_exp << 23) | ma);
            const uint32_t want = __builtin_bit_cast(uint32_t, a / b);
            if (__builtin_bit_cast(uint32_t, div_d6(a, b)) != want) { if (!bad6++) { t->a6 = ma; t->b6 = mb; } }
            if (__builtin_bit_cast(uint32_t, div_d8(a, b)) != want) { if (!bad8++) { t->a8 = ma; t->b8 = mb; } }
        }
    }
    if (bad6) { atomicAdd(&t->bad6, bad6); }
    if (bad8) { atomicAdd(&t->bad8, bad8); }
}

// random operands: a in [2^-60, 2^60), b in [2^-40, 2^40)
__global__ void sample(uint64_t seed, uint32_t per_thread, Tally* t)
{
    uint64_t s = seed + (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull;
    unsigned long long bad6 = 0, bad8 = 0;
    for (uint32_t i = 0; i < per_thread; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const uint32_t ea = 127u - 60u + (uint32_t)((s >> 46) % 120u), eb = 127u - 40u + (uint32_t)((s >> 54) % 80u);
        const float a = __builtin_bit_cast(float, (ea << 23) | (uint32_t)(s & 0x7FFFFFu));
        const float b = __builtin_bit_cast(float, (eb << 23) | (uint32_t)((s >> 23) & 0x7FFFFFu));
        const uint32_t want = __builtin_bit_cast(uint32_t, a / b);
        if (__builtin_bit_cast(uint32_t, div_d6(a, b)) != want) { bad6++; }
        if (__builtin_bit_cast(uint32_t, div_d8(a, b)) != want) { bad8++; }
    }
    if (bad6) { atomicAdd(&t->bad6, bad6); }
    if (bad8) { atomicAdd(&t->bad8, bad8); }
}

int main(int argc, char** argv)
{
    Tally* d;
    if (hipMalloc(&d, sizeof(Tally)) != hipSuccess) { printf("no device\n"); return 1; }
    (void)hipMemset(d, 0, sizeof(Tally));
    const uint32_t chunk = 1u << 14;                              // b significands per launch
    const uint32_t limit = argc > 1 ? (uint32_t)atoi(argv[1]) : (1u << 23);   // (a smaller sweep for a smoke run)
    Tally h{};
    for (uint32_t b0 = 0; b0 < limit; b0 += chunk) {
        hipLaunchKernelGGL(sweep, dim3(256, 64), dim3(256), 0, 0, b0, b0 + chunk, 127u, 127u, d);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
        if (((b0 / chunk) & 31u) == 31u) {
            (void)hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost);
            printf("b significands [0, %u): D6 mismatches %llu, D8 mismatches %llu\n", b0 + chunk, h.bad6, h.bad8);
            fflush(stdout);
        }
    }
    (void)hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("all %llu significand pairs: D6 (6 VALU) mismatches %llu", (unsigned long long)limit << 23, h.bad6);
    if (h.bad6) { printf(" (first a=0x%06X b=0x%06X)", h.a6, h.b6); }
    printf("; D8 (8 VALU) mismatches %llu", h.bad8);
    if (h.bad8) { printf(" (first a=0x%06X b=0x%06X)", h.a8, h.b8); }
    printf("\n");
    (void)hipMemset(d, 0, sizeof(Tally));
    hipLaunchKernelGGL(sample, dim3(4096), dim3(256), 0, 0, 0x4C554D41ull, 4096u, d);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    (void)hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("4294967296 random pairs, a in [2^-60, 2^60), b in [2^-40, 2^40): D6 mismatches %llu, D8 mismatches %llu\n", h.bad6, h.bad8);
    (void)hipFree(d);
    return 0;
}
