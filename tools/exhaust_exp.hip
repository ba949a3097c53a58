// tools/exhaust_exp.hip -- every float in [-87, 88]: the contract's exp with its last step as e * 2^k (2^k built through the
// exponent field, what the oracle does) against the same with v_ldexp_f32 (what the kernels do).  Prints the number of inputs
// whose results differ in any bit.  Not part of the library.
//   hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -o ab/exhaust_exp tools/exhaust_exp.hip && ab/exhaust_exp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
template <bool LDEXP>
__device__ __forceinline__ float core(float x)
{
    float kf = __builtin_rintf(x * 1.44269504088896341f);
    float r = fma_(kf, -0.693359375f, x);
    r = fma_(kf, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fma_(p, r, 1.3981999507e-3f);
    p = fma_(p, r, 8.3334519073e-3f);
    p = fma_(p, r, 4.1665795894e-2f);
    p = fma_(p, r, 1.6666665459e-1f);
    p = fma_(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float e = fma_(p, r2, r) + 1.0f;
    if (LDEXP) { return __builtin_ldexpf(e, (int)kf); }
    int k = (int)kf;
    float s = __builtin_bit_cast(float, (uint32_t)(k + 127) << 23);
    return e * s;
}

__global__ void check(unsigned long long* diff, unsigned long long* seen, uint32_t* first)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    unsigned long long d = 0, n = 0;
    for (uint64_t b = blockIdx.x * blockDim.x + threadIdx.x; b < (1ull << 32); b += stride) {
        const float x = __builtin_bit_cast(float, (uint32_t)b);
        if (!(x >= -87.0f && x <= 88.0f)) { continue; }
        n++;
        const uint32_t u = __builtin_bit_cast(uint32_t, core<false>(x)), v = __builtin_bit_cast(uint32_t, core<true>(x));
        if (u != v) { d++; atomicMin(first, (uint32_t)b); }
    }
    atomicAdd(diff, d); atomicAdd(seen, n);
}

int main()
{
    unsigned long long *d_diff, *d_seen; uint32_t* d_first;
    hipMalloc(&d_diff, 8); hipMalloc(&d_seen, 8); hipMalloc(&d_first, 4);
    hipMemset(d_diff, 0, 8); hipMemset(d_seen, 0, 8); hipMemset(d_first, 0xFF, 4);
    hipLaunchKernelGGL(check, dim3(256 * 16), dim3(256), 0, 0, d_diff, d_seen, d_first);
    unsigned long long diff = 0, seen = 0; uint32_t first = 0;
    hipMemcpy(&diff, d_diff, 8, hipMemcpyDeviceToHost); hipMemcpy(&seen, d_seen, 8, hipMemcpyDeviceToHost); hipMemcpy(&first, d_first, 4, hipMemcpyDeviceToHost);
    printf("inputs in [-87, 88]: %llu; results that differ between e * 2^k and ldexp(e, k): %llu", seen, diff);
    if (diff) { printf(" (first input bits 0x%08X)", first); }
    printf("\n");
    return diff ? 1 : 0;
}
