// tools/exhaust_sqrt.hip -- exhaustive check, on the GPU, of short correctly-rounded sqrt and
// reciprocal sequences against hipcc's own correctly rounded sqrtf() / division, over EVERY
// positive normal binary32 input.  The render kernels may only use a sequence that matches on
// every input of the range it is used on (the oracle calls libm sqrtf and IEEE division).
//   hipcc -O3 -ffp-contract=off --offload-arch=gfx950 tools/exhaust_sqrt.hip -o build/exhaust_sqrt && build/exhaust_sqrt
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define DEV __device__ __forceinline__
DEV float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// S1: one Newton step on x * rsq(x), the residual formed exactly by FMA
DEV float sqrt_s1(float x)
{
    const float y = __builtin_amdgcn_rsqf(x);
    const float g = x * y, h = 0.5f * y;
    const float r = fma_(-g, g, x);
    return fma_(r, h, g);
}
// S2: two steps
DEV float sqrt_s2(float x)
{
    const float y = __builtin_amdgcn_rsqf(x);
    const float g = x * y, h = 0.5f * y;
    const float r = fma_(-g, g, x);
    const float s = fma_(r, h, g);
    const float r2 = fma_(-s, s, x);
    return fma_(r2, h, s);
}
// S3: start from v_sqrt_f32 instead of x * rsq(x)
DEV float sqrt_s3(float x)
{
    const float g = __builtin_amdgcn_sqrtf(x);
    const float h = 0.5f * __builtin_amdgcn_rsqf(x);
    const float r = fma_(-g, g, x);
    return fma_(r, h, g);
}
// R1 / R2: reciprocal, one / two Newton steps on v_rcp_f32
DEV float rcp_r1(float s)
{
    const float y = __builtin_amdgcn_rcpf(s);
    const float e = fma_(-s, y, 1.0f);
    return fma_(e, y, y);
}
DEV float rcp_r2(float s)
{
    const float y = __builtin_amdgcn_rcpf(s);
    const float e = fma_(-s, y, 1.0f);
    const float y1 = fma_(e, y, y);
    const float e1 = fma_(-s, y1, 1.0f);
    return fma_(e1, y1, y1);
}
// N1 / N2: 1 / sqrt_rn(x) (two roundings, as normalize does) reusing rsq(x) as the first guess
DEV float rsqrt_n(float x, int steps)
{
    const float y = __builtin_amdgcn_rsqf(x);
    const float g = x * y, h = 0.5f * y;
    const float r = fma_(-g, g, x);
    const float s = fma_(r, h, g);                 // S1
    float i = y;
    for (int k = 0; k < steps; k++) { const float e = fma_(-s, i, 1.0f); i = fma_(e, i, i); }
    return i;
}

struct Tally { unsigned long long bad[8]; uint32_t first[8]; };

__global__ void sweep(uint32_t lo, uint32_t hi, Tally* t)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    unsigned long long bad[8] = {};
    for (uint64_t u = (uint64_t)lo + blockIdx.x * blockDim.x + threadIdx.x; u < hi; u += stride) {
        const float x = __builtin_bit_cast(float, (uint32_t)u);
        const float want_s = __builtin_sqrtf(x);
        const float want_r = 1.0f / x;
        const float want_n = 1.0f / want_s;
        const float got[8] = { sqrt_s1(x), sqrt_s2(x), sqrt_s3(x), rcp_r1(x), rcp_r2(x),
                               rsqrt_n(x, 1), rsqrt_n(x, 2), rsqrt_n(x, 3) };
        const float want[8] = { want_s, want_s, want_s, want_r, want_r, want_n, want_n, want_n };
        for (int k = 0; k < 8; k++) {
            if (__builtin_bit_cast(uint32_t, got[k]) != __builtin_bit_cast(uint32_t, want[k])) {
                bad[k]++;
                atomicMin(&t->first[k], (uint32_t)u);
            }
        }
    }
    for (int k = 0; k < 8; k++) { if (bad[k]) { atomicAdd(&t->bad[k], bad[k]); } }
}

int main()
{
    Tally* d;
    if (hipMalloc(&d, sizeof(Tally)) != hipSuccess) { printf("no device\n"); return 1; }
    const char* names[8] = { "sqrt S1 (rsq, 1 step)", "sqrt S2 (rsq, 2 steps)", "sqrt S3 (v_sqrt + rsq step)",
                             "rcp R1 (1 step)", "rcp R2 (2 steps)", "1/sqrt_rn N1", "1/sqrt_rn N2", "1/sqrt_rn N3" };
    // per binade pair: sqrt behaviour repeats every two binades, rcp every one; sweep ranges of exponents
    struct { uint32_t lo, hi; const char* what; } ranges[] = {
        { 0x00800000u, 0x7F800000u, "all positive normals [2^-126, 2^128)" },
        { 0x0F800000u, 0x6F800000u, "[2^-96, 2^96)" },
        { 0x1F800000u, 0x5F800000u, "[2^-64, 2^64)" },
    };
    for (auto& r : ranges) {
        Tally h{};
        for (int k = 0; k < 8; k++) { h.first[k] = 0xFFFFFFFFu; }
        hipMemcpy(d, &h, sizeof h, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(sweep, dim3(4096), dim3(256), 0, 0, r.lo, r.hi, d);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
        hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("range %s: %llu inputs\n", r.what, (unsigned long long)(r.hi - r.lo));
        for (int k = 0; k < 8; k++) {
            printf("  %-28s mismatches %llu%s", names[k], h.bad[k], h.bad[k] ? "" : "  (exact on every input)\n");
            if (h.bad[k]) { printf("  first at 0x%08X\n", h.first[k]); }
        }
    }
    hipFree(d);
    return 0;
}
