// luma_device.hpp -- the device code of the Luma render loop for gfx950 (MI355X / CDNA4): arithmetic of the contract,
// counter RNG, closest-hit scans, and the three render kernel templates.  Included by luma_kernels.hip (launchers, the
// small kernels) and by the luma_inst_*.hip files, each of which instantiates one group of kernels (luma_instances.h),
// so that the groups compile in parallel.
#pragma once
#include "luma_kernels.h"

#include <hip/hip_runtime.h>
#include <float.h>
#include <stddef.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <type_traits>

namespace luma {

#define LUMA_DEV __device__ __forceinline__

struct f3 { float x, y, z; };

LUMA_DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
LUMA_DEV f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
LUMA_DEV f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
LUMA_DEV f3 scale3(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
LUMA_DEV f3 neg3(f3 a) { return mk3(-a.x, -a.y, -a.z); }
LUMA_DEV float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
LUMA_DEV float dot3(f3 a, f3 b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }

// Correctly rounded sqrt and reciprocal, short forms.  The arithmetic contract is libm sqrtf and IEEE
// division, so whatever runs here must be correctly rounded too.  hipcc's sqrtf() expands to 17
// VALU instructions and its division to 11 (input rescaling, class checks, v_div_scale /
// v_div_fixup for the extreme exponents).  For 2^-96 <= x < 2^96 -- every discriminant, squared
// length and length on the path -- two much shorter sequences give the SAME bits:
//   sqrt:  y = v_rsq_f32(x); g = x*y; h = y/2; r = fma(-g, g, x); s = fma(r, h, g)      (5 VALU)
//   1/s:   y = v_rcp_f32(s); e = fma(-s, y, 1); i = fma(e, y, y)                         (3 VALU)
// (one Newton step whose residual the FMA forms exactly).  That is not argued but MEASURED:
// tools/exhaust_sqrt.hip compares both with hipcc's correctly rounded forms on every one of the
// 1,610,612,736 binary32 inputs of the range, on MI355X: 0 mismatches (profiles/r02/exhaust_sqrt.txt;
// outside the range they do differ, so other inputs take the compiler's forms).
// (one integer range test on the bits: 2^-96 <= x < 2^96; negative numbers, zero, subnormals, infinities and NaN fall outside)
LUMA_DEV bool lean_range(float x) { return __builtin_bit_cast(uint32_t, x) - 0x0F800000u < 0x60000000u; }
LUMA_DEV float sqrt_lean(float x)
{
    const float y = __builtin_amdgcn_rsqf(x);
    const float g = x * y, h = 0.5f * y;
    return fma_(fma_(-g, g, x), h, g);
}
LUMA_DEV float rcp_lean(float s)
{
    const float y = __builtin_amdgcn_rcpf(s);
    return fma_(fma_(-s, y, 1.0f), y, y);
}
LUMA_DEV float sqrt_rn(float x)
{
    // the short form for every lane, then the compiler's form over it for the lanes outside the range (none, in practice:
    // a repair branch that is not taken costs less scalar work than an if / else around the two forms)
    float s = sqrt_lean(x);
    if (__builtin_expect(!lean_range(x), 0)) { s = __builtin_sqrtf(x); }
    return s;
}
// fx::normalize: v * (1 / sqrt(dot(v,v))).  dot in [2^-96, 2^96) puts the length in [2^-48, 2^48).
LUMA_DEV f3 normalize3(f3 a)
{
    const float d = dot3(a, a);
    float inv = rcp_lean(sqrt_lean(d));
    if (__builtin_expect(!lean_range(d), 0)) { inv = 1.0f / __builtin_sqrtf(d); }
    return scale3(a, inv);
}
LUMA_DEV f3 fma3(f3 a, float s, f3 c) { return mk3(fma_(a.x, s, c.x), fma_(a.y, s, c.y), fma_(a.z, s, c.z)); }

// exp() of the contract: Cody-Waite by ln2, degree-5 polynomial, 2^k through the exponent
// field.  Stands in for std::exp at Renderer.cpp:61 and :427.
#ifndef LUMA_EXP_LDEXP
#define LUMA_EXP_LDEXP 1
#endif
LUMA_DEV float luma_expf_core(float x);
LUMA_DEV float luma_expf(float x)
{
    if (!(x >= -87.0f)) { return (x != x) ? x : 0.0f; }
    if (x > 88.0f) { return __builtin_inff(); }
    return luma_expf_core(x);
}

// ::tonemap, Renderer.cpp:57-70: 1 - exp(-c) per channel.  One range test covers the three
// channels (radiance is 0..a few units, so the out-of-range branches of luma_expf never run);
// anything outside takes the checked form.
LUMA_DEV float luma_expf_core(float x)                           // luma_expf for -87 <= x <= 88
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
    // e * 2^k: the contract builds 2^k through the exponent field (k in [-126, 127] here) and multiplies; v_ldexp_f32 is the
    // same correctly rounded scaling in one instruction instead of three (denormal results included: both round once)
#if LUMA_EXP_LDEXP
    return __builtin_ldexpf(e, (int)kf);
#else
    int k = (int)kf;
    float s = __builtin_bit_cast(float, (uint32_t)(k + 127) << 23);
    return e * s;
#endif
}
LUMA_DEV f3 tonemap3(f3 c)
{
    // (fmin/fmax drop a NaN operand: a NaN channel may take either form, and is NaN after both)
    const float hi = __builtin_fmaxf(__builtin_fmaxf(c.x, c.y), c.z), lo = __builtin_fminf(__builtin_fminf(c.x, c.y), c.z);
    if (__builtin_expect(hi <= 87.0f && lo >= -88.0f, 1)) {
        return mk3(1.0f - luma_expf_core(-c.x), 1.0f - luma_expf_core(-c.y), 1.0f - luma_expf_core(-c.z));
    }
    return mk3(1.0f - luma_expf(-c.x), 1.0f - luma_expf(-c.y), 1.0f - luma_expf(-c.z));
}

// pcg3d (Jarzynski & Olano 2020) over (pixel, sample, slot) ^ keys: the counter RNG.
struct u3 { uint32_t x, y, z; };
LUMA_DEV uint32_t pcg_first(uint32_t v) { return v * 1664525u + 1013904223u; }
// (x already through pcg_first: the ray-per-lane kernel keeps its pixel in that form)
LUMA_DEV u3 pcg3d_x(uint32_t x, uint32_t y, uint32_t z)
{
    y = y * 1664525u + 1013904223u;
    z = z * 1664525u + 1013904223u;
    x += y * z; y += z * x; z += x * y;
    x ^= x >> 16; y ^= y >> 16; z ^= z >> 16;
    x += y * z; y += z * x; z += x * y;
    u3 r; r.x = x; r.y = y; r.z = z; return r;
}
LUMA_DEV u3 pcg3d(uint32_t x, uint32_t y, uint32_t z) { return pcg3d_x(pcg_first(x), y, z); }

// fx::Random::vec3(lo, hi) at Renderer.cpp:118 and :226.
LUMA_DEV f3 rng_vec3_x(const RenderArgs& A, uint32_t pixel_x, uint32_t sample, uint32_t slot, float lo, float hi)
{
    u3 h = pcg3d_x(pixel_x, sample ^ A.key[1], slot ^ A.key[2]);
    const float w = (hi - lo) * 5.9604644775390625e-8f;          // (hi-lo) * 2^-24
    return mk3(fma_((float)(h.x >> 8), w, lo), fma_((float)(h.y >> 8), w, lo), fma_((float)(h.z >> 8), w, lo));
}
LUMA_DEV f3 rng_vec3(const RenderArgs& A, uint32_t pixel, uint32_t sample, uint32_t slot, float lo, float hi)
{
    u3 h = pcg3d(pixel ^ A.key[0], sample ^ A.key[1], slot ^ A.key[2]);
    const float w = (hi - lo) * 5.9604644775390625e-8f;          // (hi-lo) * 2^-24
    return mk3(fma_((float)(h.x >> 8), w, lo), fma_((float)(h.y >> 8), w, lo), fma_((float)(h.z >> 8), w, lo));
}

// ::lerp, Renderer.cpp:72-83
LUMA_DEV f3 color_lerp(f3 c1, f3 c2, float t)
{
    float t1 = 1.0f - t;
    return mk3(fma_(c2.x, t, c1.x * t1), fma_(c2.y, t, c1.y * t1), fma_(c2.z, t, c1.z * t1));
}

// fx::apply(mat4, vec4): m[col][row], left-to-right fused chain (Camera.cpp:203, :212).
LUMA_DEV void apply4(const float* m, float v0, float v1, float v2, float v3, float out[4])
{
#pragma unroll
    for (int r = 0; r < 4; r++) {
        out[r] = fma_(m[12 + r], v3, fma_(m[8 + r], v2, fma_(m[4 + r], v1, m[r] * v0)));
    }
}

// Local row of the band -> row of the frame (contiguous band or interleaved stripes).
LUMA_DEV uint32_t global_row(const RenderArgs& A, uint32_t yl)
{
    return A.row_begin + (yl / A.stripe) * A.pitch + (yl % A.stripe);
}

// Band-local pixel index -> index in the whole frame.  Recomputed where it is needed (once per pixel and launch, in
// the epilogue) from the one index that is live there: a register held across the render loop for it costs the
// ray-per-lane kernel its eighth wave per SIMD (65 VGPRs).
template <typename ArgsPtr>
LUMA_DEV uint32_t frame_pixel(ArgsPtr A, uint32_t local)
{
    const uint32_t w = A->width, stripe = A->stripe;
    const uint32_t yl = local / w, x = local - yl * w;
    const uint32_t y = A->row_begin + (yl / stripe) * A->pitch + (yl % stripe);
    return y * w + x;
}

// One entry of Camera::recompute_rays, Camera.cpp:193-213, in registers.
template <typename ArgsPtr>
LUMA_DEV f3 camera_ray(ArgsPtr A, uint32_t x, uint32_t y)
{
    float proj_inv[16], view_inv[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { proj_inv[i] = A->proj_inv[i]; view_inv[i] = A->view_inv[i]; }
    float cx = (float)x / (float)A->width;
    float cy = (float)y / (float)A->height;
    cx = cx * 2.f - 1.f;
    cy = cy * 2.f - 1.f;
    float target[4];
    apply4(proj_inv, cx, cy, 1.f, 1.f, target);
    const float scalar = 1.0f / target[3];
    f3 n = normalize3(scale3(mk3(target[0], target[1], target[2]), scalar));
    float ray[4];
    apply4(view_inv, n.x, n.y, n.z, 0.f, ray);
    return mk3(ray[0], ray[1], ray[2]);
}

// Closest-hit linear scan, Renderer::trace_ray (Renderer.cpp:234-287), over records staged
// in LDS.  Sphere index is wave-uniform, so every ds_read_b128 is a broadcast.
//   CAM_ORIGIN: the ray starts at camera.pos (every primary / bounce ray in the reference,
//   Renderer.cpp:303,:416), so diff = camera.pos - centre and c = |diff|^2 - r^2 are
//   per-sphere constants precomputed into `rec` = (diff.xyz, c).
//   otherwise: `rec` = (centre.xyz, radius) and diff, c are formed per test (:244-248).
// Returns the winning index (or -1) and t; hit point / normal are formed by the caller for
// the winner only (the reference normalises every candidate, :264; same values).
// A pointer into read-only global memory through which wave-uniform reads compile to scalar loads.
#if defined(__HIP_DEVICE_COMPILE__)
#define LUMA_SCALAR_PTR const __attribute__((address_space(4))) float4*
#else
#define LUMA_SCALAR_PTR const float4*
#endif
template <bool CAM_ORIGIN, bool UNROLL4, int NFIX = 0, typename RecPtr = const float4*>
LUMA_DEV int scan_spheres(RecPtr rec, uint32_t n, f3 origin, f3 dir, float& t_out)
{
    const float a = dot3(dir, dir);                              // :246 (hoisted: loop invariant)
    // t = num / a with num = -b - sqrt(d).  a > 0 is the same for every sphere, so candidates
    // are ordered by num and the divide is done once, for the winner.  Rounding can make two
    // different num give the same t (the reference then keeps the earlier sphere, :266), so
    // when num is within 2^-22 of the best, or num / a could leave the normal range (flush to 0
    // and fail :258, or overflow past FLT_MAX and fail :266), both quotients are formed and
    // compared exactly as the reference does: with 1e-12 < a < 1e12 and 1e-20 <= num <= 1e25
    // the quotient lies in [1e-32, 1e37], normal and below FLT_MAX.
    const bool odd_a = !(a > 1e-12f && a < 1e12f);
    float best_num = __builtin_inff();
    int best = -1;
    bool any_exact = false;                                      // some candidate of this scan took the exact path

    // b (half of :247) and d (quarter of :251) of one sphere
    auto test = [&](const float4 s, float& b, float& d) {
        float c;
        if (CAM_ORIGIN) {
            b = dot3(mk3(s.x, s.y, s.z), dir);
            c = s.w;
        } else {
            const f3 diff = sub3(origin, mk3(s.x, s.y, s.z));    // :244
            b = dot3(diff, dir);
            c = fma_(-s.w, s.w, dot3(diff, diff));               // :248
        }
        d = fma_(b, b, -(a * c));
    };
    // b >= 0 gives -b - sqrt(d) < 0, i.e. t <= 0 (:258 rejects it): such spheres skip the sqrt.
    auto candidate = [](float b, float d) { return d > 0.0f && b < 0.0f; };   // :253
    // The common case is settled with selects, not branches (round 3: the divergent-if form of this block cost ~24 scalar
    // and branch instructions per sphere for ~22 vector ones): `fast` says the numerator lies in [1e-20, 1e25] (one integer
    // range test on its bits, which also says it is positive) and clearly below the best so far; what is positive and below
    // the best but not `fast` (ties, extreme values, odd a) compares the quotients as the reference does, out of line.
    auto consider = [&](float b, float d, uint32_t i) {
        if (candidate(b, d)) {
            const float num = -b - sqrt_rn(d);                   // :255, numerator
            const bool fast = (__builtin_bit_cast(uint32_t, num) - 0x1E3CE508u <= 0x69045951u - 0x1E3CE508u) & (num < best_num * 0.99999976f) & !odd_a;
            const bool exact = (num > 0.0f) & (num < best_num) & !fast;
            bool take = fast;
            if (__builtin_expect(exact, 0)) {
                any_exact = true;
                const float t = num / a;
                const float tb = (best < 0) ? FLT_MAX : best_num / a;
                take = (t > 0.0f) && (t < tb);                   // :258, :266
            }
            best_num = take ? num : best_num;
            best = take ? (int)i : best;
        }
    };

    // Four records per trip: the four broadcast ds_read_b128 are in flight together (one LDS
    // round trip per four spheres instead of one each), and one wave-level branch skips the
    // candidate handling when none of the four is hit by any lane, the common case in a
    // large scene.  Candidates are still considered in index order (:242, :266).
    // (Used for scenes big enough to run 16-wave workgroups; small scenes keep the plain
    // loop, whose 63 VGPRs allow 8 waves per SIMD.)
    uint32_t i = 0;
    if (NFIX > 0) {
        // the scene size is a compile-time constant (launch_render picks the instance): no loop
        // counter and no address arithmetic
#pragma unroll
        for (int q = 0; q < NFIX; q++) {
            float b, d;
            test(rec[q], b, d);
            consider(b, d, (uint32_t)q);
            // (the reads stay where they are: hoisted ahead of the loop they cost 40 VGPRs and three waves per SIMD)
            asm volatile("" ::: "memory");
        }
        i = n;
    }
    if (UNROLL4) {
        for (; i + 4 <= n; i += 4) {
            const float4 s0 = rec[i], s1 = rec[i + 1], s2 = rec[i + 2], s3 = rec[i + 3];
            float b0, d0, b1, d1, b2, d2, b3, d3;
            test(s0, b0, d0); test(s1, b1, d1); test(s2, b2, d2); test(s3, b3, d3);
            // the largest discriminant of the four is a cheap superset test for "some candidate" (see wave_scan_one)
            const float any = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(d0, d1), d2), d3);
            if (any > 0.0f) {
                consider(b0, d0, i); consider(b1, d1, i + 1); consider(b2, d2, i + 2); consider(b3, d3, i + 3);
            }
        }
    }
    for (; i < n; i++) {
        float b, d;
        test(rec[i], b, d);
        consider(b, d, i);
    }
    // The winner's quotient.  A winner that no candidate contested through the exact path has 1e-12 < a < 1e12 and
    // 1e-20 <= num <= 1e25: quotient, reciprocal and residuals are normal, where the six-instruction division below gives
    // the correctly rounded quotient for every pair of significands (tools/exhaust_div.hip: all 2^46 pairs,
    // profiles/r02/exhaust_div.txt) instead of the compiler's eleven; anything else takes the compiler's.
    float q;
    {
        const float y0 = __builtin_amdgcn_rcpf(a);
        const float y = fma_(fma_(-a, y0, 1.0f), y0, y0);
        const float q0 = best_num * y;
        q = fma_(fma_(-a, q0, best_num), y, q0);
    }
    if (__builtin_expect(any_exact, 0)) { q = best_num / a; }
    t_out = (best >= 0) ? q : FLT_MAX;
    return best;
}


// Debug builds (-DLUMA_DBG_WALK=1|2|3) turn the sphere-test counter of scan_grid<false> / wave_grid_walk into
// 1: hops summed over lanes, 2: hop-loop trips of the wave, 3: test-body trips of the wave (tools/grid_walk_stats.py)
#ifndef LUMA_DBG_WALK
#define LUMA_DBG_WALK 0
#endif
#define LUMA_WAVE_ONCE(CNT) do { const unsigned long long m_ = __builtin_amdgcn_ballot_w64(true); \
                               if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(m_)) { (CNT)++; } } while (0)

struct Hit { f3 pos, normal; };

LUMA_DEV Hit finish_hit(const float4* __restrict__ rec_gen, int idx, f3 origin, f3 dir, float t)
{
    const float4 s = rec_gen[idx];
    Hit h;
    h.pos = fma3(dir, t, origin);                                // :260-261
    h.normal = normalize3(sub3(h.pos, mk3(s.x, s.y, s.z)));      // :263-264
    return h;
}

// Closest hit through the uniform grid (FLAG_GRID): the spheres of the cells the ray crosses, in
// the order it crosses them, plus the `large` list.  Each sphere gets the reference's test
// (:244-258) and the winner is the smallest t, the lowest index among equal t (:266) -- a rule
// that does not depend on the order of the tests, so the result is the linear scan's.  The walk
// stops once the best hit lies before the exit of the current cell by a clear margin; the cell
// lists are built from boxes padded by more than the arithmetic can displace a hit (luma_grid.cpp).
template <bool CAM_ORIGIN>
LUMA_DEV int scan_grid(const GridArgs& G, const uint16_t* __restrict__ lds_cells, const float4* __restrict__ rec_cam,
                       const float4* __restrict__ rec_gen, f3 origin, f3 dir, float& t_out, uint32_t& n_tests)
{
    const float a = dot3(dir, dir);                              // :246
    float best_t = FLT_MAX;
    int best = -1;
    auto test = [&](uint32_t i) {
        float b, c;
        if (CAM_ORIGIN) {
            const float4 s = rec_cam[i];
            b = dot3(mk3(s.x, s.y, s.z), dir);
            c = s.w;
        } else {
            const float4 s = rec_gen[i];
            const f3 diff = sub3(origin, mk3(s.x, s.y, s.z));    // :244
            b = dot3(diff, dir);
            c = fma_(-s.w, s.w, dot3(diff, diff));               // :248
        }
        const float d = fma_(b, b, -(a * c));
        if (LUMA_DBG_WALK == 0 || CAM_ORIGIN) { n_tests++; }
        if (LUMA_DBG_WALK == 3 && !CAM_ORIGIN) { LUMA_WAVE_ONCE(n_tests); }
        if (d > 0.0f && b < 0.0f) {                              // :253
            const float t = (-b - sqrt_rn(d)) / a;               // :255
            if (t > 0.0f && (t < best_t || (t == best_t && (int)i < best))) { best_t = t; best = (int)i; }   // :258, :266
        }
    };

    for (uint32_t j = 0; j < G.n_large; j++) { test(G.large[j]); }

    // the part of the ray inside the grid's box: [t0, t1]  (per-axis scalars: no indexed registers)
    float t0 = 0.0f, t1 = FLT_MAX;
    bool inside = true;
    auto clip = [&](float o, float d, float org, uint32_t dim, float& inv) {
        const float lo = org, hi = org + (float)dim * G.h;
        if (d != 0.0f) {
            inv = 1.0f / d;
            const float ta = (lo - o) * inv, tb = (hi - o) * inv;
            t0 = __builtin_fmaxf(t0, __builtin_fminf(ta, tb));
            t1 = __builtin_fminf(t1, __builtin_fmaxf(ta, tb));
        } else {
            inv = 0.0f;
            inside = inside && (o >= lo && o <= hi);
        }
    };
    float ix, iy, iz;
    clip(origin.x, dir.x, G.org[0], G.dims[0], ix);
    clip(origin.y, dir.y, G.org[1], G.dims[1], iy);
    clip(origin.z, dir.z, G.org[2], G.dims[2], iz);
    if (inside && t0 <= t1 && a > 0.0f) {
        float t_cell = FLT_MAX;                                  // time to cross one cell along the fastest axis
        auto enter = [&](float o, float d, float inv, float org, uint32_t dim, int& cell, float& t_next) {
            const float p = fma_(d, t0, o);
            cell = min(max((int)__builtin_floorf((p - org) * G.inv_h), 0), (int)dim - 1);
            if (d != 0.0f) {
                t_next = (org + (float)(cell + (d > 0.0f ? 1 : 0)) * G.h - o) * inv;
                t_cell = __builtin_fminf(t_cell, G.h * __builtin_fabsf(inv));
            } else {
                t_next = FLT_MAX;
            }
        };
        int cx, cy, cz;
        float tx, ty, tz;
        enter(origin.x, dir.x, ix, G.org[0], G.dims[0], cx, tx);
        enter(origin.y, dir.y, iy, G.org[1], G.dims[1], cy, ty);
        enter(origin.z, dir.z, iz, G.org[2], G.dims[2], cz, tz);
        const float margin = 1e-3f * t_cell;
        const uint32_t most = G.dims[0] + G.dims[1] + G.dims[2] + 3u;    // a straight line crosses no more cells
        for (uint32_t hop = 0; hop < most; hop++) {
            if (LUMA_DBG_WALK == 1 && !CAM_ORIGIN) { n_tests++; }
            if (LUMA_DBG_WALK == 2 && !CAM_ORIGIN) { LUMA_WAVE_ONCE(n_tests); }
            // (at most 4M + 3 x 258^2 cells, luma_grid.cpp: the index fits 32 bits with room to spare)
            const uint32_t id = ((uint32_t)cz * G.dims[1] + (uint32_t)cy) * G.dims[0] + (uint32_t)cx;
            // the walk waits on this lookup at every hop: from the workgroup's LDS copy of the offsets
            // (16 bit, staged by the kernel when the table fits) it costs an LDS round trip, not an L2 one
            uint32_t k0, k1;
            if (lds_cells) { k0 = lds_cells[id]; k1 = lds_cells[id + 1]; }
            else           { k0 = G.cell_start[id]; k1 = G.cell_start[id + 1]; }
            for (uint32_t k = k0; k < k1; k++) { test(G.items[k]); }
            const float t_exit = __builtin_fminf(__builtin_fminf(tx, ty), tz);
            if (best_t < t_exit - margin || t_exit > t1 + margin) { break; }
            // cross the nearest face (when two are due at once, one per hop)
            if (tx <= ty && tx <= tz) {
                cx += dir.x > 0.0f ? 1 : -1;
                if (cx < 0 || cx >= (int)G.dims[0]) { break; }
                tx = (G.org[0] + (float)(cx + (dir.x > 0.0f ? 1 : 0)) * G.h - origin.x) * ix;
            } else if (ty <= tz) {
                cy += dir.y > 0.0f ? 1 : -1;
                if (cy < 0 || cy >= (int)G.dims[1]) { break; }
                ty = (G.org[1] + (float)(cy + (dir.y > 0.0f ? 1 : 0)) * G.h - origin.y) * iy;
            } else {
                cz += dir.z > 0.0f ? 1 : -1;
                if (cz < 0 || cz >= (int)G.dims[2]) { break; }
                tz = (G.org[2] + (float)(cz + (dir.z > 0.0f ? 1 : 0)) * G.h - origin.z) * iz;
            }
        }
    }
    t_out = best_t;
    return best;
}

// The grid arguments of the launch, read where they are used: a pointer into the kernel-argument segment that the
// compiler cannot look through, so the ~30 scalars behind it are loaded (scalar loads, cached) inside the code that
// needs them instead of sitting in SGPRs -- or worse, in spill lanes -- for the whole kernel.  RenderArgs is the
// kernels' only argument, so it starts the segment.
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(4))) GridArgs* KernargGridPtr;
LUMA_DEV KernargGridPtr kernarg_grid()
{
    auto p = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return (KernargGridPtr)(p + offsetof(RenderArgs, grid));
}
typedef const __attribute__((address_space(4))) RenderArgs* KernargArgsPtr;
LUMA_DEV KernargArgsPtr kernarg_args()
{
    auto p = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return (KernargArgsPtr)p;
}
#else
typedef const GridArgs* KernargGridPtr;
LUMA_DEV KernargGridPtr kernarg_grid() { return nullptr; }
typedef const RenderArgs* KernargArgsPtr;
LUMA_DEV KernargArgsPtr kernarg_args() { return nullptr; }
#endif

// Closest hit of a ray FROM camera.pos through the direction grid (FLAG_GRID; luma_grid.h
// HostViewGrid): no walk at all, the ray's direction picks one bin of a cube map around camera.pos
// and the bin lists every sphere such a ray can hit.  Same test, same (t, index) rule as scan_grid.
// FROM_GEN: `rec` holds the general records (centre, radius) and the test derives the camera-origin record
// from them with prepare_records_kernel's own arithmetic (the same bits), for kernels that keep only the
// general records in LDS.
template <bool FROM_GEN = false, typename GridPtr = const GridArgs*>
LUMA_DEV int scan_view(GridPtr G, const float4* __restrict__ rec, f3 campos, f3 dir, float& t_out, uint32_t& n_tests)
{
    const float a = dot3(dir, dir);                              // :246
    float best_t = FLT_MAX;
    int best = -1;
    auto test = [&](uint32_t i) {
        float4 s = rec[i];
        if (FROM_GEN) {
            const f3 diff = sub3(campos, mk3(s.x, s.y, s.z));
            s = make_float4(diff.x, diff.y, diff.z, fma_(-s.w, s.w, dot3(diff, diff)));
        }
        const float b = dot3(mk3(s.x, s.y, s.z), dir);
        const float d = fma_(b, b, -(a * s.w));
        if (LUMA_DBG_WALK == 0) { n_tests++; }
        if (d > 0.0f && b < 0.0f) {                              // :253
            const float t = (-b - sqrt_rn(d)) / a;               // :255
            if (t > 0.0f && (t < best_t || (t == best_t && (int)i < best))) { best_t = t; best = (int)i; }   // :258, :266
        }
    };
    for (uint32_t j = 0; j < G->n_view_large; j++) { test(G->view_large[j]); }

    // face = 2 * major axis + (negative), u and v = the other two components over |major|; the bins
    // are padded by a whole bin on the host, so the approximate reciprocal is accurate enough
    const float ax = __builtin_fabsf(dir.x), ay = __builtin_fabsf(dir.y), az = __builtin_fabsf(dir.z);
    float major, u, v;
    uint32_t face;
    if (ax >= ay && ax >= az) { major = dir.x; u = dir.y; v = dir.z; face = 0u; }
    else if (ay >= az)        { major = dir.y; u = dir.x; v = dir.z; face = 2u; }
    else                      { major = dir.z; u = dir.x; v = dir.y; face = 4u; }
    if (a > 0.0f && a < __builtin_inff()) {                      // (a zero or non-finite direction hits nothing small)
        face += major < 0.0f ? 1u : 0u;
        const float inv = __builtin_amdgcn_rcpf(__builtin_fabsf(major));
        const int last = (int)G->view_res - 1;
        const int iu = min(max((int)__builtin_floorf(fma_(u * inv, G->view_half_res, G->view_half_res)), 0), last);
        const int iv = min(max((int)__builtin_floorf(fma_(v * inv, G->view_half_res, G->view_half_res)), 0), last);
        const uint32_t bin = (face * G->view_res + (uint32_t)iv) * G->view_res + (uint32_t)iu;
        const uint32_t k1 = G->view_start[bin + 1];
        // the bin's spheres come nearest first: stop once the best hit is nearer than the rest can be
        // (distance = t * |dir|; 1e-3 of slack covers the approximate square root many times over)
        const float len = a * __builtin_amdgcn_rsqf(a) * 1.001f;
        for (uint32_t k = G->view_start[bin]; k < k1; k++) {
            const uint2 it = G->view_items[k];
            if (best_t * len < __builtin_bit_cast(float, it.y)) { break; }
            test(it.x);
        }
    }
    t_out = best_t;
    return best;
}

// ::RGB's shift-and-OR (Renderer.cpp:48-52), or the byte order R,G,B,255 for PACK_RGBA8.
LUMA_DEV uint32_t pack_pixel(int r, int g, int b, uint32_t flags)
{
    if (flags & FLAG_PACK_RGBA8) {
        return (uint32_t)r | ((uint32_t)g << 8) | ((uint32_t)b << 16) | 0xFF000000u;
    }
    return ((uint32_t)r << 16) | ((uint32_t)g << 8) | (uint32_t)b;
}

// Wave-level sum of a per-lane counter, one atomic per wave.  The counter is split over
// kCounterSlots cache lines picked by workgroup id: tens of thousands of waves adding into ONE
// address serialise at the memory side (0.3 ms at 1080p), spread over 256 lines they do not.
LUMA_DEV void wave_add_u64(unsigned long long* slots, uint32_t v)
{
    unsigned long long s = v;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_down(s, off, 64);
    }
    if ((threadIdx.x & 63) == 0 && s != 0) {
        const uint32_t slot = (blockIdx.y * gridDim.x + blockIdx.x) % kCounterSlots;
        atomicAdd(slots + slot * kCounterStride, s);
    }
}

// ---------------------------------------------------------------------------------
// The fused render kernel.  One thread per pixel; a wave covers a TILE_W x (64/TILE_W)
// pixel tile; a 256-thread workgroup covers 2x2 such tiles.
//
// The reference nests  frames > pixels > bounces  (Renderer.cpp:393-395, :300).  Bounce
// counts differ per sample (a miss ends the path, :318), so a literal nest leaves most of a
// wave idle while its longest path finishes.  Here each lane runs ONE flat loop over
// (sample, bounce) steps of its own pixel: a lane whose path ends adds the sample to its
// accumulator and starts its next sample in the same iteration, so lanes only idle at the
// very end of the launch.  Per pixel the samples are still added in frame order, which
// keeps the sum bit-identical to frame-by-frame accumulation (:457-458).  The spp loop
// lives in registers: one read-modify-write of the accumulation planes and one coalesced
// store of the packed pixel per launch.
// ---------------------------------------------------------------------------------
// Workgroups are WAVES waves arranged wg_waves_x(WAVES) wide: 4 waves = 2x2 (32x8 pixels),
// 16 waves = 4x4 (64x16 pixels).  The large shape is for scenes whose LDS image limits a CU to
// one or two workgroups: 16 waves keep 4 per SIMD to cover the LDS latency of the scan.
constexpr int wg_waves_x(int waves) { return waves == 16 ? 4 : waves >= 2 ? 2 : 1; }

// Where the scene lives during the launch: scan records (32 B/sphere) and, when they also
// fit, the material vectors (24 B/sphere) are staged into LDS once per workgroup.
enum : int { SCENE_GLOBAL = 0, SCENE_LDS_ALL = 2 };

//
// SPLIT (1, 2 or 4) spreads the frames of one pixel over SPLIT lanes of the same workgroup when
// the band has too few pixels to fill the GPU (8 GPUs x 1080p leave 259k pixels per GPU for
// 524k thread slots, and one lane needs ~0.4 ms for 64 frames whatever the band size).  Lane 0
// of a pixel adds the frames it renders to the accumulator as before; the others park each
// frame's colour in a global scratch plane, and after a workgroup barrier lane 0 adds those in
// frame order, so the float sum is still the reference's ((acc + f0) + f1) + ...
// (Renderer.cpp:457-458), bit for bit.  Which lane renders which frame is settled at run time
// (see `claim` below).
// FINE (0: none; 4 or 8) gives the launch a second class of workgroups: the band's last rows, from
// RenderArgs::fine_row0 on, run FINE lanes per pixel in workgroups dispatched after all the others, so
// that the launch does not drain for as long as one of the long workgroups takes (plan_split has the
// measurements).  With FINE the grid is one-dimensional; the coarse class (SPLIT lanes per pixel, a plain
// frame loop without claims when SPLIT == 1) comes first in it.
// TAKES gives the workgroups with one lane per pixel the frame taking described at STEAL in the kernel (launch_render
// picks these instances for noisy launches of kStealSppMin .. kStealSppMax frames per call).
// (GRID: the walk waits on dependent loads, so six waves per SIMD with ~40 B of spills beat four
//  without: 2.93 -> 2.15 ms at 4096 spheres; five: 2.42, seven: 2.12 / 3.05 at 65536, eight: 2.60)
#ifdef LUMA_DBG_TIMELINE
// debug builds only (tools/band_timeline.py): when and where every wave of the ray-per-lane kernel ran
// (debug builds are single translation units: make SINGLE_TU=1, so this is the one copy)
__device__ uint4 dbg_timeline[1u << 18];
extern "C" int luma_dbg_timeline_copy(void* host, size_t bytes)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(dbg_timeline), bytes);
}
#endif
// Build-time switches of the two round-3 loop forms (DEFER: hits wait for company; STEAL: lanes take frames inside their
// wave), for the A/B builds under profiles/r03 (make EXTRA=-D...); the values below are the library's.
#ifndef LUMA_DEFER
#define LUMA_DEFER 1
#endif
#ifndef LUMA_DEFER_NUM
#define LUMA_DEFER_NUM 1u
#define LUMA_DEFER_DEN 2u
#endif
#ifndef LUMA_STEAL
#define LUMA_STEAL 1
#endif
#ifndef LUMA_DEFER_SPLIT2
#define LUMA_DEFER_SPLIT2 0     // with two lanes per pixel: 2.5 % slower (profiles/r03/ab_frames_taken.txt)
#endif
#ifndef LUMA_DEFER_GENERAL
// run-time sphere counts: with bounce rays from hit points (both scans of a trip are general-origin ones) -9..-16 %; with
// the reference's camera-origin bounce rays +-2 % at 16 / 64 frames per call and 9-12 % slower at 4 (2: for those too)
// (profiles/r03/mid_scene_takes.txt, scene_size_sweep_deferral.txt)
#define LUMA_DEFER_GENERAL 1
#endif
#ifndef LUMA_STEAL_POOL
#define LUMA_STEAL_POOL 256u
#endif
#ifndef LUMA_STEAL_POOL_GENERAL
#define LUMA_STEAL_POOL_GENERAL 128u   // run-time sphere counts: the scene's records want the LDS (profiles/r03/mid_scene_takes.txt)
#endif
#ifndef LUMA_SPLIT_WAVES
#define LUMA_SPLIT_WAVES 7
#endif
#ifndef LUMA_DEFER_WAVES
#define LUMA_DEFER_WAVES 7
#endif
template <int TILE_W, int WAVES, int SCENE, bool NOISE, bool FROM_HIT, int SPLIT, bool GRID, int NS = 0, int FINE = 0, bool TAKES = false>
__global__ __launch_bounds__(WAVES * 64, GRID ? 6 : (LUMA_DEFER && NS > 0) ? (SPLIT >= 2 ? LUMA_SPLIT_WAVES : LUMA_DEFER_WAVES) : 1) void render_kernel(const RenderArgs A)
{
    // [0,n): camera-origin records; [n,2n): general records; then mat0[n] (float4), mat1[n] (float2)
    extern __shared__ float4 lds_rec[];
#ifdef LUMA_DBG_TIMELINE
    const unsigned long long dbg_t0 = wall_clock64();
#endif

    const uint32_t n = A.n_spheres;
    const float4* rec_cam = A.records;
    const float4* rec_gen = A.records + n;
    const float4* mat0 = A.mat0;
    const float2* mat1 = A.mat1;
    if (SCENE != SCENE_GLOBAL) {
        // Stage global -> LDS, 16 B per lane, coalesced.
        for (uint32_t i = threadIdx.x; i < 2u * n; i += blockDim.x) { lds_rec[i] = A.records[i]; }
        rec_cam = lds_rec;
        rec_gen = lds_rec + n;
        if (SCENE == SCENE_LDS_ALL) {
            float4* l0 = lds_rec + 2u * n;
            float2* l1 = reinterpret_cast<float2*>(l0 + n);
            for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) { l0[i] = A.mat0[i]; l1[i] = A.mat1[i]; }
            mat0 = l0;
            mat1 = l1;
        }
        __syncthreads();
    }

    // FLAG_GRID: the uniform grid's cell offsets as 16-bit words in LDS, when launch_render found room
    const uint16_t* grid_cells = nullptr;
    if (GRID && A.grid.lds_cells) {
        uint16_t* stage = reinterpret_cast<uint16_t*>(lds_rec);
        for (uint32_t i = threadIdx.x; i < A.grid.lds_cells; i += blockDim.x) { stage[i] = (uint16_t)A.grid.cell_start[i]; }
        __syncthreads();
        grid_cells = stage;
    }

    // lane -> pixel inside the wave tile; wave -> (frame block k, tile inside the workgroup)
    constexpr bool SPL = SPLIT > 1 || FINE > 0;                  // frames of a pixel are claimed through LDS
    // Parked colours: three planes per slot (a wave's 64 pixels side by side), or with four lanes per pixel, where three
    // quarters of all frames are parked, a pixel's slots side by side, 12 B each (bands of 4 / 8 of the config #3 frame
    // 0.494 -> 0.478, 0.257 -> 0.251 ms; the whole frame and its band of 2 lose 0.4 / 1.3 % that way: profiles/r03/ab_park_layout.txt)
    constexpr bool PARK_BY_PIXEL = SPLIT == 4;
    // hits wait for company (see the loop skeletons below); not with two lanes per pixel (a band of two of the config #3
    // frame: 0.942 -> 0.965 ms with it, profiles/r03/ab_deferred_hits.txt) and not for the noise-free one-frame launches
    constexpr bool DEFER = LUMA_DEFER && (NS > 0 || (LUMA_DEFER_GENERAL && FROM_HIT) || LUMA_DEFER_GENERAL > 1) && !GRID && NOISE && (SPLIT != 2 || LUMA_DEFER_SPLIT2);
    // TAKES: lanes out of frames take frames of their wave's other pixels (STEAL below); instances of their own, because the
    // code costs launches that cannot gain from it (config #4, 256 frames per pixel: 22.7 ms without it, 23.4 with it
    // switched off at run time, profiles/r03/ab_frames_taken.txt); launch_render picks them up to kStealSppMax frames
    static_assert(!TAKES || (NOISE && SPLIT == 1 && !GRID && (NS > 0 || FINE == 0)), "only these instances exist");
    constexpr bool STEAL = LUMA_STEAL && TAKES;                  // (with the plain loop where DEFER is off: run-time sphere counts)
    constexpr int TILE_H = 64 / TILE_W;
    constexpr int WPK = WAVES / SPLIT;                           // waves per frame block
    constexpr int WGX = wg_waves_x(WPK), WGY = WPK / WGX;
    // FINE > WAVES: a wave of the fine class holds FINE / WAVES lanes of each of its pixels, side by side
    // (FINE = 8 in 4 waves: lanes 0-31 and 32-63 of a wave are two lanes of the same 32x1 pixels)
    constexpr bool HALF = FINE > WAVES;
    constexpr int LPW = HALF ? FINE / WAVES : 1;                 // lanes of one pixel in one wave
    static_assert(!HALF || 64 / LPW == TILE_W, "a split wave of the fine class covers one row of a tile");
    constexpr int WPKF = HALF ? 1 : FINE ? WAVES / FINE : WPK;   // waves per frame block in the workgroups of the fine class
    constexpr int WGXF = wg_waves_x(WPKF), WGYF = WPKF / WGXF;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = SPL ? (uint32_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : threadIdx.x >> 6;
    uint32_t k, wv, x, yl;
    uint32_t cidx;                                               // the pixel's slot in dof_shared / claim
    bool live;
    bool lone = !SPL;                                            // this workgroup has one lane per pixel: frames 0 .. spp-1, no claims
    if (FINE) {
        // one-dimensional grid: the coarse class first (rows [0, fine_row0), SPLIT lanes per pixel), then the
        // fine one (rows [fine_row0, rows), FINE lanes per pixel); see plan_split
        const uint32_t b = blockIdx.x;
        const bool fine = b >= A.coarse_wgs;
        lone = SPLIT == 1 && !fine;
        const uint32_t bb = fine ? b - A.coarse_wgs : b, gx = fine ? A.fine_gx : A.coarse_gx;
        const uint32_t by = bb / gx, bx = bb - by * gx;
        if (fine && HALF) {
            k = wave * LPW + lane / TILE_W; wv = 0;
            x = bx * TILE_W + (lane % TILE_W);
            yl = A.fine_row0 + by;
            live = (x < A.width) && (yl < A.rows);
        } else if (fine) {
            k = wave / WPKF; wv = wave % WPKF;
            x = bx * (WGXF * TILE_W) + (wv % WGXF) * TILE_W + (lane % TILE_W);
            yl = A.fine_row0 + by * (WGYF * TILE_H) + (wv / WGXF) * TILE_H + (lane / TILE_W);
            live = (x < A.width) && (yl < A.rows);
        } else {
            k = wave / WPK; wv = wave % WPK;
            x = bx * (WGX * TILE_W) + (wv % WGX) * TILE_W + (lane % TILE_W);
            yl = by * (WGY * TILE_H) + (wv / WGX) * TILE_H + (lane / TILE_W);
            live = (x < A.width) && (yl < A.fine_row0);
        }
        cidx = (fine && HALF) ? lane % TILE_W : wv * 64 + lane;
    } else {
        k = wave / WPK; wv = wave % WPK;
        x = blockIdx.x * (WGX * TILE_W) + (wv % WGX) * TILE_W + (lane % TILE_W);
        yl = blockIdx.y * (WGY * TILE_H) + (wv / WGX) * TILE_H + (lane / TILE_W);
        live = (x < A.width) && (yl < A.rows);
        cidx = wv * 64 + lane;
    }
    const uint32_t y = global_row(A, yl);

    const uint32_t pixel = y * A.width + x;                      // global index: RNG is band independent (:399)
    const uint32_t local = yl * A.width + x;                     // index into the band's planes
    const size_t npix = A.park_pixels;                           // the parking planes cover rows [park_row0, rows)
    const uint32_t plocal = local - A.park_row0 * A.width;
    const f3 campos = mk3(A.cam_pos[0], A.cam_pos[1], A.cam_pos[2]);
    const uint32_t slots = 2u + A.paths;
    const bool pathtrace = A.mode == MODE_PATHTRACE;
    uint32_t nrays = 0, ntests = 0;                              // ntests: counted under GRID only (else rays * n)
#ifdef LUMA_DBG_STEAL
    uint32_t dbg_iters = 0, dbg_events = 0, dbg_taken = 0, dbg_hits = 0;   // tools/steal_stats.py (with LUMA_DBG_TIMELINE)
#endif

    // Renderer::trace_ray (:234-287): linear scan, or the same tests through the grid
    auto closest_cam = [&](f3 o, f3 d, float& t) {
        if (GRID) {
            if (A.grid.view_res) { return scan_view(&A.grid, rec_cam, campos, d, t, ntests); }   // every such ray starts at camera.pos
            return scan_grid<true>(A.grid, grid_cells, rec_cam, rec_gen, o, d, t, ntests);
        }
        // compile-time sphere count: the records come by scalar loads (s_load_dwordx4, one per sphere and scan,
        // left inside the loop: kept in SGPRs for the whole launch they cost two waves per SIMD, 1.86 against 1.83 ms)
        if (NS > 0) { return scan_spheres<true, false, NS>((LUMA_SCALAR_PTR)A.records, n, o, d, t); }
        return scan_spheres<true, (SCENE == SCENE_GLOBAL), NS>(rec_cam, n, o, d, t);
    };
    auto closest_gen = [&](f3 o, f3 d, float& t) {
        if (GRID) { return scan_grid<false>(A.grid, grid_cells, rec_cam, rec_gen, o, d, t, ntests); }
        if (NS > 0) { return scan_spheres<false, false, NS>((LUMA_SCALAR_PTR)(A.records + NS), n, o, d, t); }
        return scan_spheres<false, (SCENE == SCENE_GLOBAL), NS>(rec_gen, n, o, d, t);
    };

    // DOF pre-pass, :416-432.  Noise free, so it is the same for every frame of the call; with
    // SPLIT > 1 lane 0 of the pixel traces it and hands blur and gain to the others through LDS.
    __shared__ float2 dof_shared[SPL ? (WPK > WPKF ? WPK : WPKF) * 64 : 1];
    // Frames of a pixel are not dealt out in fixed blocks: its SPLIT lanes claim them one at a
    // time from one LDS word (low half: claims from the front, by lane 0; high half: claims from
    // the back, by the others).  Lane 0 takes frames 0, 1, 2, ... and adds them directly, the
    // others take spp-1, spp-2, ... and park them; a claim is good while front + back < spp, so
    // the two ends meet wherever the lanes' luck puts them and no lane waits on a slower sibling.
    __shared__ uint32_t claim[SPL ? (WPK > WPKF ? WPK : WPKF) * 64 : 1];
    constexpr uint32_t kPool = NS > 0 ? LUMA_STEAL_POOL : LUMA_STEAL_POOL_GENERAL;   // slots per wave
    __shared__ uint32_t steal_pool[STEAL ? WAVES * kPool * 4u : 1];
    __shared__ uint8_t steal_pairing[STEAL ? WAVES * 64 : 1];
    static_assert(!STEAL || WAVES * kPool * 16u <= 32768u, "the pools of a workgroup's waves fit beside the scene");
    if (SPL && k == 0) { claim[cidx] = 0u; }
    f3 dir0 = mk3(0.f, 0.f, -1.f);
    float blur = 0.f, gainf = 1.f;
    // The camera's matrices, and further down what only the epilogue needs, are read from the kernel-argument
    // segment where they are used (kernarg_args) instead of through `A`: as members of the by-value struct they
    // are all loaded at kernel entry, and the 92 SGPRs that took cost a wave per SIMD (config #3: 74 SGPRs, eight
    // waves, 1.822 -> 1.800 ms; bands of 2 / 8: 1.003 -> 0.972, 0.290 -> 0.281, profiles/r02/ab_lazy_args.txt).
    const auto KA = kernarg_args();                              // (taken in uniform control flow)
    if (live) { dir0 = camera_ray(KA, x, y); }                    // camera.rays[index], :413
    if (live && k == 0) {
        float t0;
        const int h0 = closest_cam(campos, dir0, t0);
        nrays++;
        const float depth = (h0 >= 0) ? t0 : FLT_MAX;            // :419-423
        const float dd = depth - A.focus_depth;                  // :425
        const float x2 = dd * dd, x4 = x2 * x2, x8 = x4 * x4, x10 = x8 * x2;
        const float focus = 1.0f - luma_expf(-x10);              // :427
        blur = fma_(focus, 4.5f, 0.5f);                          // :429-432
        gainf = (A.show_depth && (focus < .5f)) ? 1.2f : 1.0f;   // :448-452
        if (SPL) { dof_shared[cidx] = make_float2(blur, gainf); }
    }
    if (SPL) {
        __syncthreads();
        if (live && k != 0) { const float2 v = dof_shared[cidx]; blur = v.x; gainf = v.y; }
    }

    // bounces == 0: the loop at :300 never runs, every sample adds tonemap(0) * gain = +0.
    const uint32_t spp_all = A.bounces ? A.spp : 0u;
    float acc_r = 0.f, acc_g = 0.f, acc_b = 0.f;
    uint32_t front = 0;                                          // frames lane 0 added itself: [0, front)

    if (live) {
        if (k == 0) { acc_r = KA->acc[0][local]; acc_g = KA->acc[1][local]; acc_b = KA->acc[2][local]; }

        // per-lane path state (slot == bounce * slots: first RNG slot of the bounce)
        uint32_t s = (FINE && lone && !STEAL) ? ~0u : 0u, slot = 0;
        const uint32_t slot_end = slots * A.bounces;              // slot == bounce * slots
        f3 direct = mk3(0.f, 0.f, 0.f);
        f3 dir = dir0;                                           // :293
        f3 origin = campos;                                      // :294, :303

        // next frame of this lane: s, or false when the pixel has none left
        auto next_frame = [&]() -> bool {
            if (FINE && lone) { s++; return s < spp_all; }
            const uint32_t old = atomicAdd(&claim[cidx], k == 0 ? 1u : 0x10000u);
            const uint32_t f = old & 0xFFFFu, b = old >> 16;
            // claims from the back stop at the last parking slot (refused ones still count up `b`, hence the min)
            if (k == 0 ? f + min(b, A.park_depth) >= spp_all : (b >= A.park_depth) | (f + b >= spp_all)) { return false; }
            s = (k == 0) ? f : spp_all - 1u - b;
            return true;
        };
        bool have = (!SPL || (STEAL && lone)) ? spp_all > 0u : next_frame();   // one lane per pixel everywhere: plainly frames 0 .. spp-1

        // STEAL (workgroups with one lane per pixel): a lane that has rendered the last unclaimed frame of its own pixel
        // takes frames from the BACK of another pixel of its wave (spp-1, spp-2, ...) while that pixel's lane goes on from the
        // front, and leaves each colour in a slot of the wave's pool in LDS; the pixel's lane adds them in frame order when the
        // wave's loop is over, so the sum is still ((acc + f0) + f1) + ... (Renderer.cpp:457-458).  A wave's lanes run in
        // lockstep, so all of this is plain reads and writes.  `s_end`: the pixel's own lane renders frames [0, s_end), the
        // others took [s_end, spp) (0 in a lane that is on another pixel's frame: it has none of its own left).  `taken`: the
        // pool slot of the latest, i.e. lowest, frame others took from this lane's pixel (each slot links to the one before,
        // the first to kOwn).  `slot_out`: where this lane's current frame goes (kOwn: into its own accumulator).
        // Pools of LUMA_STEAL_POOL slots per wave even out 64 frames per pixel; from kStealSppMax frames on the lanes differ
        // by more than that and nothing is gained (config #4, 256 frames).
        constexpr uint32_t kOwn = 0xFFFFu;
        uint32_t taken = kOwn, slot_out = kOwn, pool_used = 0, s_end = spp_all;
#ifdef LUMA_STEAL_NEVER
        bool steals = false;                                     // probe builds: the code without its effect
#else
        bool steals = STEAL && lone;
#endif
        uint32_t rpix = pcg_first(pixel ^ A.key[0]);              // the pixel's word of the RNG counter, through its first step
        uint32_t* const pool = steal_pool + (STEAL ? (uint32_t)__builtin_amdgcn_readfirstlane(wave) * (kPool * 4u) : 0u);
        volatile uint8_t* const pairing = steal_pairing + (STEAL ? (uint32_t)__builtin_amdgcn_readfirstlane(wave) * 64u : 0u);

        // One trip of the flat (sample, bounce) loop, in the four pieces the two loop skeletons below arrange:
        // the ray of the trip and its scan; a miss; a hit (the long piece: the indirect ray with its scan, shading,
        // the reflected direction); the end of a path.
        // (`dir` is the direction the trip starts from until its scan and the jittered one, `dn` of the reference's loop,
        //  from there on: one set of registers, also while a hit waits in the second skeleton)
        float t = 0.f;
        int idx = -1;
        bool path_ends = false;
        auto trip_scan = [&]() {
            const uint32_t sample = A.frame_index + s;
            if (NOISE) { dir = add3(dir, rng_vec3_x(A, rpix, sample, slot, -blur, blur)); }   // :118-119, :302
            else       { dir = add3(dir, mk3(0.f, 0.f, 0.f)); }
            if (FROM_HIT && slot > 0u) { idx = closest_gen(origin, dir, t); }
            else                        { idx = closest_cam(origin, dir, t); }   // :304
            nrays++;
        };
        auto trip_miss = [&]() {                                 // :306-319
            const float clamped = __builtin_fminf(__builtin_fmaxf(dir.y, -1.f), 1.f);
            const float adjusted = (clamped + 1.f) * .5f;
            direct = add3(direct, color_lerp(mk3(.529f, .808f, .922f), mk3(.106f, .275f, .711f), adjusted));
            path_ends = true;
        };
        auto trip_hit = [&]() {
            const uint32_t sample = A.frame_index + s;
            const Hit hit = finish_hit(rec_gen, idx, origin, dir, t);
            const float4 m0 = mat0[idx];                       // diffuse rgb, albedo
            const float2 m1 = mat1[idx];                       // metallic, roughness
            const f3 diffuse = mk3(m0.x, m0.y, m0.z);
            const float cos_theta = dot3(neg3(dir), hit.normal);       // :331-332

            if (cos_theta >= 0.0f) {                             // :334
                f3 indirect = mk3(0.f, 0.f, 0.f);
                if (m0.w > 0.0f) {                               // :337
                    if (pathtrace) {                             // :187
                        for (uint32_t p = 0; p < A.paths; p++) {       // :192
                            f3 d = normalize3(rng_vec3_x(A, rpix, sample, slot + 2u + p, -1.0f, 1.0f));  // :194
                            if (dot3(d, hit.normal) < 0.0f) { d = neg3(d); }   // :197-200
                            const f3 nd = add3(hit.normal, d);         // :202
                            float tc;
                            const int ci = closest_gen(hit.pos, nd, tc);   // :204
                            nrays++;
                            if (ci >= 0) {                       // :206-209
                                const float4 cm = mat0[ci];
                                indirect = add3(indirect, mk3(cm.x, cm.y, cm.z));
                            }
                        }
                    }
                    indirect = scale3(indirect, A.inv_samples);        // :212-213
                    indirect = scale3(indirect, m0.w);           // :340
                }
                const float value = (cos_theta + 1.f) * .5f;     // :343
                const float scalar = value * (1.f - m1.x);       // :345
                const f3 capture = color_lerp(diffuse, indirect, 1.f - scalar);   // :346
                direct = fma3(capture, scalar * .8f, direct);        // :347-349
            }

            path_ends = (m1.x == 0.0f) || (slot + slots == slot_end);  // :353-357, :300 (the bounce just made was the last)
            if (!path_ends) {
                f3 nrm = hit.normal;
                if (NOISE) {
                    const float gain = .2f * m1.y;               // :225
                    nrm = add3(nrm, rng_vec3_x(A, rpix, sample, slot + 1u, -gain, gain));   // :226-227
                } else {
                    nrm = add3(nrm, mk3(0.f, 0.f, 0.f));
                }
                const float k = 2.0f * dot3(dir, nrm);            // fx::reflect, :229
                dir = fma3(nrm, -k, dir);                         // :359
                if (FROM_HIT) { origin = fma3(hit.normal, .001f, hit.pos); }   // :221-222
                slot += slots;
            }
        };
        auto trip_end = [&]() {
            const f3 result = scale3(tonemap3(direct), gainf);        // :362, :57-70; :452 (x * 1.0f is x)
            if (STEAL && lone && slot_out != kOwn) {
                uint32_t* const e = pool + slot_out * 4u;
                e[0] = __float_as_uint(result.x); e[1] = __float_as_uint(result.y); e[2] = __float_as_uint(result.z);
            } else if (!SPL || k == 0) {
                acc_r += result.x; acc_g += result.y; acc_b += result.z;   // :457-458
            } else {
                if (PARK_BY_PIXEL) {
                    float* park = KA->split_scratch + ((size_t)plocal * A.park_depth + (spp_all - 1u - s)) * 3u;   // slot b of the claim
                    park[0] = result.x; park[1] = result.y; park[2] = result.z;
                } else {
                    float* park = KA->split_scratch + (size_t)(spp_all - 1u - s) * 3u * npix + plocal;
                    park[0] = result.x; park[npix] = result.y; park[2 * npix] = result.z;
                }
            }
            if (STEAL && lone) {
                // the pixel's own lane goes on from the front while frames are left; a lane on another pixel's frame
                // (s_end == 0) looks for the next one in steal_frames
                if (s + 1u < s_end) { s++; } else { have = false; }
            } else if (!SPL) { s++; } else { have = next_frame(); }
            slot = 0;
            direct = mk3(0.f, 0.f, 0.f);
            dir = dir0;
            origin = campos;
        };

        // lanes without a frame are paired, in lane order, with the lanes whose own pixel has unclaimed frames left
        auto steal_frames = [&]() {
            const uint64_t wanting = __ballot(!have);
            if (wanting == 0ull) { return; }
            const bool gives = have && s + 1u < s_end;
            const uint64_t giving = __ballot(gives);
            const uint32_t pairs = min(min((uint32_t)__builtin_popcountll(wanting), (uint32_t)__builtin_popcountll(giving)),
                                       kPool - pool_used);
            // no pixel with a frame to give (none will come: unclaimed frames only go), or the pool is full: that was the last try
            if (pairs == 0u) { steals = false; return; }
            const uint64_t mine = have ? giving : wanting;
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mine >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mine, 0u));
            const bool giver = gives && rank < pairs, taker = !have && rank < pairs;
            if (giver) { pairing[rank] = (uint8_t)lane; }
            uint32_t from = lane;
            if (taker) { from = pairing[rank]; }
            const int at = (int)(from << 2);
            const uint32_t g_taken = (uint32_t)__builtin_amdgcn_ds_bpermute(at, (int)taken);
            const uint32_t g_end = (uint32_t)__builtin_amdgcn_ds_bpermute(at, (int)s_end);
            const uint32_t g_pix = (uint32_t)__builtin_amdgcn_ds_bpermute(at, (int)rpix);
            const float g_dx = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(at, (int)__float_as_uint(dir0.x)));
            const float g_dy = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(at, (int)__float_as_uint(dir0.y)));
            const float g_dz = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(at, (int)__float_as_uint(dir0.z)));
            const float g_blur = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(at, (int)__float_as_uint(blur)));
            const float g_gain = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(at, (int)__float_as_uint(gainf)));
            const uint32_t fresh = pool_used + rank;             // the pair's pool slot
            if (taker) {
                slot_out = fresh;
                pool[fresh * 4u + 3u] = g_taken;                 // links to the frame taken before (a higher one)
                s = g_end - 1u;                                  // the last frame of the other pixel that nobody has
                s_end = 0u;
                rpix = g_pix; dir0 = mk3(g_dx, g_dy, g_dz); blur = g_blur; gainf = g_gain;
                dir = dir0;
                have = true;
            }
            if (giver) { taken = fresh; s_end--; }
            pool_used += pairs;
#ifdef LUMA_DBG_STEAL
            dbg_events++; dbg_taken = pool_used;
#endif
        };

        if (!DEFER && !STEAL) {
            while (!SPL ? s < spp_all : have) {
                path_ends = false;
                trip_scan();
                if (idx < 0) { trip_miss(); } else { trip_hit(); }
                if (path_ends) { trip_end(); }
            }
        } else if (!DEFER) {
            // the plain loop with frames taken inside the wave (scenes with a run-time sphere count: a trip costs a whole scan
            // however few lanes are left, so every trip the wave saves counts in full)
            for (;;) {
                path_ends = false;
                if (have) {
                    trip_scan();
                    if (idx < 0) { trip_miss(); } else { trip_hit(); }
                    if (path_ends) { trip_end(); }
                }
                if (steals) { steal_frames(); }
                if (__ballot(have) == 0ull) { break; }
            }
            if (lone) {
                for (uint32_t e = taken; e != kOwn; ) {
                    const uint32_t* const q = pool + e * 4u;
                    acc_r += __uint_as_float(q[0]); acc_g += __uint_as_float(q[1]); acc_b += __uint_as_float(q[2]);
                    e = q[3];
                }
            }
        } else {
            // The hit piece is most of a trip's instructions and about 38 % of a wave's lanes want it on a trip.  Here a
            // lane whose ray hit keeps the hit (idx, t, dn: registers the scan filled anyway) and sits out while the
            // others go on with their next rays, until LUMA_DEFER_NUM / LUMA_DEFER_DEN of the wave's working lanes hold
            // one (or none is left to scan): then the piece runs once for all of them.  Each lane's own sequence of
            // operations is what it was; only how the lanes' sequences interleave changes.
            if (!SPL) { have = s < spp_all; }
            idx = -1;                                            // >= 0: this lane holds a hit
            for (;;) {
#ifdef LUMA_DBG_STEAL
                dbg_iters++;
#endif
                path_ends = false;
                if (have && idx < 0) {
                    trip_scan();
                    if (idx < 0) { trip_miss(); }
                }
                const uint32_t holding = __builtin_popcountll(__ballot(idx >= 0));
                const uint32_t working = __builtin_popcountll(__ballot(have));
                if (working == 0u) { break; }
                if (holding * LUMA_DEFER_DEN >= working * LUMA_DEFER_NUM) {
#ifdef LUMA_DBG_STEAL
                    dbg_hits++;
#endif
                    if (idx >= 0) { trip_hit(); idx = -1; }
                }
                if (path_ends) {
                    trip_end();
                    if (!SPL && !STEAL) { have = s < spp_all; }
                }
                if (steals) { steal_frames(); }
            }
            if (STEAL && lone) {
                // the frames other lanes rendered for this pixel, lowest first
                for (uint32_t e = taken; e != kOwn; ) {
                    const uint32_t* const q = pool + e * 4u;
                    acc_r += __uint_as_float(q[0]); acc_g += __uint_as_float(q[1]); acc_b += __uint_as_float(q[2]);
                    e = q[3];
                }
            }
        }
    }

    if (SPL) {
        __syncthreads();                                         // parked frames are visible to the workgroup
        // lane 0 of the pixel stopped at its first refused claim: its half of the word is one past its frames
        front = (FINE && lone) ? spp_all : (claim[cidx] & 0xFFFFu) - 1u;
    }

    const auto E = kernarg_args();                               // what only the epilogue needs, read here
    if (live && k == 0) {
        if (SPL) {
            // the other lanes' frames, added in frame order (:457-458); eight frames' loads are
            // issued together so the loop waits for memory once per eight frames, not per frame
            uint32_t f = front;
            for (; f + 8u <= spp_all; f += 8u) {
                float pr[8], pg[8], pb[8];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    if (PARK_BY_PIXEL) {
                        const float* park = E->split_scratch + ((size_t)plocal * A.park_depth + (spp_all - 1u - (f + j))) * 3u;
                        pr[j] = park[0]; pg[j] = park[1]; pb[j] = park[2];
                    } else {
                        const float* park = E->split_scratch + (size_t)(spp_all - 1u - (f + j)) * 3u * npix + plocal;
                        pr[j] = park[0]; pg[j] = park[npix]; pb[j] = park[2 * npix];
                    }
                }
#pragma unroll
                for (int j = 0; j < 8; j++) { acc_r += pr[j]; acc_g += pg[j]; acc_b += pb[j]; }
            }
            for (; f < spp_all; f++) {
                if (PARK_BY_PIXEL) {
                    const float* park = E->split_scratch + ((size_t)plocal * A.park_depth + (spp_all - 1u - f)) * 3u;
                    acc_r += park[0]; acc_g += park[1]; acc_b += park[2];
                } else {
                    const float* park = E->split_scratch + (size_t)(spp_all - 1u - f) * 3u * npix + plocal;
                    acc_r += park[0]; acc_g += park[npix]; acc_b += park[2 * npix];
                }
            }
        }
        E->acc[0][local] = acc_r; E->acc[1][local] = acc_g; E->acc[2][local] = acc_b;

        const float inv = 1.0f / E->frame_count_last;            // :460
        int r = (int)((acc_r * inv) * 255.0f);                   // ::RGB, :44-55
        int g = (int)((acc_g * inv) * 255.0f);
        int b = (int)((acc_b * inv) * 255.0f);
        const uint32_t flags = E->flags;
        if (flags & FLAG_CLAMP_PACK) {
            r = min(max(r, 0), 255); g = min(max(g, 0), 255); b = min(max(b, 0), 255);
        }
        const uint32_t px = pack_pixel(r, g, b, flags);
        E->target[local] = px;                                   // :461
        uint32_t* const mirror = E->target_mirror;
        if (mirror) { mirror[E->mirror_is_frame ? frame_pixel(E, local) : local] = px; }
    }

#ifdef LUMA_DBG_TIMELINE
    {
        const unsigned long long dbg_t1 = wall_clock64();
        const uint32_t w = (blockIdx.y * gridDim.x + blockIdx.x) * WAVES + (threadIdx.x >> 6);
        if ((threadIdx.x & 63u) == 0 && w < (1u << 18)) {
            const uint32_t hw = __builtin_amdgcn_s_getreg(4 | (31 << 11)), xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11));
#ifdef LUMA_DBG_STEAL
            dbg_timeline[w] = make_uint4((uint32_t)dbg_t0, (uint32_t)dbg_t1, dbg_iters | (dbg_hits << 16), dbg_taken | (dbg_events << 16));
            (void)hw; (void)xcc;
#else
            dbg_timeline[w] = make_uint4((uint32_t)dbg_t0, (uint32_t)dbg_t1, hw, xcc);
#endif
        }
    }
#endif
    unsigned long long* const counters = E->counters;
    wave_add_u64(counters, nrays);
    if (GRID) { wave_add_u64(counters + 1, ntests); wave_add_u64(counters + 2, nrays); }   // tests, and rays of grid launches
}

// ---------------------------------------------------------------------------------
// Sphere-parallel variant for large scenes (n >= kSphereParallelMin).
//
// With thousands of spheres the scan is >99 % of the work and "one lane = one ray" wastes it
// twice: lanes whose path has ended (or that have no indirect ray this step) idle through a
// whole N-sphere scan, and every lane re-reads the same record.  Here the scan is turned
// around: the wave takes ONE ray at a time (four in large scenes, see wave_scan_many), broadcasts it to SGPRs
// (v_readlane), and its 64 lanes test 64 DIFFERENT spheres per instruction, reading 1 KiB of records per ds_read_b128
// (conflict free: 16 B per lane, contiguous).  Every lane does useful work on every scan
// instruction whatever the divergence of the paths; rays that do not exist are simply not
// scanned.  Each lane keeps the best (t, index) among its own spheres (its indices ascend, so
// the strict `<` of Renderer.cpp:266 holds), and two wave-wide minima (DPP + v_readlane, see
// wave_min_u32) pick the lexicographic minimum of (t, index): the smallest t, and among equal t
// the lowest index, which is exactly what the reference's in-order scan returns.  The rest of
// the loop (RNG, shading, accumulation) stays one lane = one pixel.
// ---------------------------------------------------------------------------------

// Wave-wide unsigned min, result uniform.  Four DPP steps make every row of 16 lanes uniform
// (quad_perm xor 1, xor 2, row_half_mirror, row_mirror), then one lane per row is read back and
// the four values are combined on the scalar unit.  Needs EXEC = all lanes.
LUMA_DEV uint32_t wave_min_u32(uint32_t v)
{
    auto dpp = [](uint32_t x, auto ctrl) {
        return (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, decltype(ctrl)::value, 0xF, 0xF, false);
    };
    v = min(v, dpp(v, std::integral_constant<int, 0xB1>{}));    // quad_perm [1,0,3,2]
    v = min(v, dpp(v, std::integral_constant<int, 0x4E>{}));    // quad_perm [2,3,0,1]
    v = min(v, dpp(v, std::integral_constant<int, 0x141>{}));   // row_half_mirror
    v = min(v, dpp(v, std::integral_constant<int, 0x140>{}));   // row_mirror
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), r3 = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    return min(min(r0, r1), min(r2, r3));
}

// Scan for lane `j`'s ray over records laid out [batch][lane] in LDS; n_pad is a multiple of
// 64 and the padding records are NaN (never a candidate).  All 64 lanes must call this.
template <bool CAM_ORIGIN>
LUMA_DEV unsigned long long wave_scan_one(const float4* __restrict__ rec, uint32_t n_pad, uint32_t lane,
                                          float ox, float oy, float oz, float dx, float dy, float dz, float a)
{
    float best_t = FLT_MAX;                                      // :236-238
    uint32_t best_i = 0xFFFFFFFFu;

    auto test = [&](const float4 s, float& b, float& d) {
        float c;
        if (CAM_ORIGIN) {
            b = fma_(s.z, dz, fma_(s.y, dy, s.x * dx));          // :247 (half), diff precomputed
            c = s.w;
        } else {
            const float fx = ox - s.x, fy = oy - s.y, fz = oz - s.z;   // :244
            b = fma_(fz, dz, fma_(fy, dy, fx * dx));
            c = fma_(-s.w, s.w, fma_(fz, fz, fma_(fy, fy, fx * fx)));  // :248
        }
        d = fma_(b, b, -(a * c));                                // :251 (quarter)
    };
    auto consider = [&](float b, float d, uint32_t i) {
        if (d > 0.0f && b < 0.0f) {                              // :253 (b >= 0 gives t <= 0, rejected at :258)
            const float t = (-b - sqrt_rn(d)) / a;               // :255
            if (t > 0.0f && t < best_t) { best_t = t; best_i = i; }   // :258, :266
        }
    };

    uint32_t base = 0;
    for (; base + 256 <= n_pad; base += 256) {
        const float4 s0 = rec[base + lane], s1 = rec[base + 64 + lane];
        const float4 s2 = rec[base + 128 + lane], s3 = rec[base + 192 + lane];
        float b0, d0, b1, d1, b2, d2, b3, d3;
        test(s0, b0, d0); test(s1, b1, d1); test(s2, b2, d2); test(s3, b3, d3);
        // (a superset test on the discriminants alone: two v_max3 and a compare per four tests where min(d, -b) per test
        //  cost eight; the few passes it lets through for spheres behind the ray find no candidate)
        const float any = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(d0, d1), d2), d3);
        if (any > 0.0f) {
            consider(b0, d0, base + lane); consider(b1, d1, base + 64 + lane);
            consider(b2, d2, base + 128 + lane); consider(b3, d3, base + 192 + lane);
        }
    }
    for (; base < n_pad; base += 64) {                           // up to three more batches of 64
        float b, d;
        test(rec[base + lane], b, d);
        consider(b, d, base + lane);
    }

    // Lexicographic min of (t, index) over the wave, in registers (DPP + readlane, no LDS round
    // trips): first the smallest t (t > 0, so its bit pattern orders like its value), then the
    // lowest index among the lanes that hold it.  0xFFFFFFFF in the index = no hit.
    const uint32_t t_min = wave_min_u32(__builtin_bit_cast(uint32_t, best_t));
    const uint32_t i_min = wave_min_u32(__builtin_bit_cast(uint32_t, best_t) == t_min ? best_i : 0xFFFFFFFFu);
    return ((unsigned long long)t_min << 32) | i_min;
}

// Scan for up to R rays at once (cnt of them, broadcast to SGPRs by the caller) over records laid out
// [batch][lane] in LDS; n_pad is a multiple of 64 and the padding records are NaN (never a candidate).  All 64
// lanes must call this.  Every record a lane reads is tested against all the rays: per ray the arithmetic, its
// order and the candidates are the same whatever R, so the results are the same bits.  Rays per pass, kernel ms
// (tools/scan_rays_ab.sh): config #5 band 147.9 / 141.6 / 139.4 / 137.9 for R = 1 / 2 / 3 / 4, 1080p x 4 spp of
// 5000 spheres 25.5 / 24.0 / 23.5 / 23.2, of 1100: 6.57 / 6.30 / 6.26 / 6.20; of 320 and 640 (four-wave workgroups,
// five per CU while the kernel keeps to 96 VGPRs): 3.26 / 3.33 / 3.29 / 3.32 and 4.31 / 4.41 / 4.43 / 4.38.  So four
// rays in the 16-wave workgroups of large scenes, one in the others (render_kernel_sp; R == 1 goes through
// wave_scan_one, 3 % faster at 640 spheres than this function with one ray).
// (Six / eight rays in the 16-wave workgroups: 137.1 / 136.1 ms against 137.7, and 6.31 / 6.35 against 6.17 at 1100.)
template <bool CAM_ORIGIN, int R>
LUMA_DEV void wave_scan_many(const float4* __restrict__ rec, uint32_t n_pad, uint32_t lane, uint32_t cnt,
                             const float (&ox)[R], const float (&oy)[R], const float (&oz)[R],
                             const float (&dx)[R], const float (&dy)[R], const float (&dz)[R], const float (&a)[R],
                             unsigned long long (&key)[R])
{
    float best_t[R];
    uint32_t best_i[R];
#pragma unroll
    for (int r = 0; r < R; r++) { best_t[r] = FLT_MAX; best_i[r] = 0xFFFFFFFFu; }   // :236-238

    auto test = [&](int r, const float4 s, float& b, float& d) {
        float c;
        if (CAM_ORIGIN) {
            b = fma_(s.z, dz[r], fma_(s.y, dy[r], s.x * dx[r]));  // :247 (half), diff precomputed
            c = s.w;
        } else {
            const float fx = ox[r] - s.x, fy = oy[r] - s.y, fz = oz[r] - s.z;   // :244
            b = fma_(fz, dz[r], fma_(fy, dy[r], fx * dx[r]));
            c = fma_(-s.w, s.w, fma_(fz, fz, fma_(fy, fy, fx * fx)));  // :248
        }
        d = fma_(b, b, -(a[r] * c));                             // :251 (quarter)
    };
    auto consider = [&](int r, float b, float d, uint32_t i) {
        if (d > 0.0f && b < 0.0f) {                              // :253 (b >= 0 gives t <= 0, rejected at :258)
            const float t = (-b - sqrt_rn(d)) / a[r];            // :255
            if (t > 0.0f && t < best_t[r]) { best_t[r] = t; best_i[r] = i; }   // :258, :266
        }
    };

    uint32_t base = 0;
    for (; base + 256 <= n_pad; base += 256) {
        const float4 s0 = rec[base + lane], s1 = rec[base + 64 + lane];
        const float4 s2 = rec[base + 128 + lane], s3 = rec[base + 192 + lane];
#pragma unroll
        for (int r = 0; r < R; r++) {
            if (R == 1 || (uint32_t)r < cnt) {                             // wave-uniform
                float b0, d0, b1, d1, b2, d2, b3, d3;
                test(r, s0, b0, d0); test(r, s1, b1, d1); test(r, s2, b2, d2); test(r, s3, b3, d3);
                const float any = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(d0, d1), d2), d3);   // (as in wave_scan_one)
                if (any > 0.0f) {
                    consider(r, b0, d0, base + lane); consider(r, b1, d1, base + 64 + lane);
                    consider(r, b2, d2, base + 128 + lane); consider(r, b3, d3, base + 192 + lane);
                }
            }
        }
    }
    for (; base < n_pad; base += 64) {                           // up to three more batches of 64
        const float4 s0 = rec[base + lane];
#pragma unroll
        for (int r = 0; r < R; r++) {
            if (R == 1 || (uint32_t)r < cnt) {
                float b, d;
                test(r, s0, b, d);
                consider(r, b, d, base + lane);
            }
        }
    }

#pragma unroll
    for (int r = 0; r < R; r++) {
        key[r] = ~0ull;
        if (R == 1 || (uint32_t)r < cnt) {                                 // the lexicographic minimum of (t, index)
            const uint32_t t_min = wave_min_u32(__builtin_bit_cast(uint32_t, best_t[r]));
            const uint32_t i_min = wave_min_u32(__builtin_bit_cast(uint32_t, best_t[r]) == t_min ? best_i[r] : 0xFFFFFFFFu);
            key[r] = ((unsigned long long)t_min << 32) | i_min;
        }
    }
}

// Scans the ray of every lane with want == true (R at a time) and returns, in that
// lane, the closest hit (index or -1, and t).  Wave-uniform control flow: every lane calls it.
template <bool CAM_ORIGIN, int R>
LUMA_DEV void wave_scan(const float4* __restrict__ rec, uint32_t n_pad, uint32_t lane, bool want,
                        f3 origin, f3 dir, int& idx, float& t)
{
    const float a = dot3(dir, dir);                              // :246, per lane, then broadcast
    idx = -1;
    t = FLT_MAX;
    unsigned long long todo = __builtin_amdgcn_ballot_w64(want);
    auto from_lane = [](float v, int j) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), j)); };
    while (todo) {
        int j[R];
        uint32_t cnt = 0;
        float ox[R], oy[R], oz[R], dx[R], dy[R], dz[R], aj[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            j[r] = todo ? __builtin_ctzll(todo) : j[0];          // (past cnt: lane j[0] again, not scanned)
            cnt += todo ? 1u : 0u;
            todo &= todo - 1;
            dx[r] = from_lane(dir.x, j[r]); dy[r] = from_lane(dir.y, j[r]); dz[r] = from_lane(dir.z, j[r]);
            aj[r] = from_lane(a, j[r]);
            ox[r] = oy[r] = oz[r] = 0.f;
            if (!CAM_ORIGIN) { ox[r] = from_lane(origin.x, j[r]); oy[r] = from_lane(origin.y, j[r]); oz[r] = from_lane(origin.z, j[r]); }
        }
        unsigned long long key[R];
        if constexpr (R == 1) { key[0] = wave_scan_one<CAM_ORIGIN>(rec, n_pad, lane, ox[0], oy[0], oz[0], dx[0], dy[0], dz[0], aj[0]); }
        else { wave_scan_many<CAM_ORIGIN, R>(rec, n_pad, lane, cnt, ox, oy, oz, dx, dy, dz, aj, key); }
#pragma unroll
        for (int r = 0; r < R; r++) {
            if ((R == 1 || (uint32_t)r < cnt) && lane == (uint32_t)j[r] && (uint32_t)key[r] != 0xFFFFFFFFu) {
                idx = (int)(uint32_t)key[r];
                t = __builtin_bit_cast(float, (uint32_t)(key[r] >> 32));
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// FLAG_GRID in the wave-uniform kernel: the walk of the uniform grid shared by the wave.
//
// In render_kernel the rays from hit points walk the grid one ray per lane INSIDE the divergent hit
// side: the lanes without such a ray idle through the whole walk, the lanes with one wait for the
// longest walk of the wave, and inside a hop the lanes whose cell is empty wait for the fullest cell.
// Here every lane of the wave takes part, in two roles:
//  * WALKERS.  Each of the W rays that need a walk gets one lane, and the 64 - W lanes left over are shared out
//    in proportion to the cells a ray may cross (its length inside the grid in cells, summed over the axes):
//    a ray owns ADJACENT lanes (a lane finds its ray through a byte map in LDS and a ballot, and reads origin,
//    direction and range from the owner's registers with ds_bpermute), lane s of its L walking the s-th L-th of
//    the ray's parameter range inside the grid (a 3D-DDA entered at the start of the segment).  A walker does not test the spheres of its cell: it pushes (walker lane, sphere) pairs
//    onto a queue of the wave in LDS, at most kPushPerRound per round.
//  * TESTERS.  Whenever the queue holds a wave's worth of pairs (or no walker is left) all 64 lanes take one
//    pair each, fetch the pair's ray from its walker's registers (ds_bpermute), run the reference's test
//    (:244-258) and fold a hit into the ray's slot with one 64-bit LDS atomic minimum on (t bits, index):
//    smallest t, lowest index among equal t -- the reference's closest-hit rule (Renderer.cpp:266), which
//    does not depend on the order of the tests.
// A walker stops once its ray's slot holds a hit before the exit of its cell by a clear margin (any
// segment's hit counts), at the end of its segment, or outside the grid.  A hit whose point lies in a
// segment is found through that segment's cells, because the sphere is listed in the cell that holds the
// point; pairs pushed beyond that are tested all the same and can only lose.  So the set of tests is a
// superset of what the lane-per-ray walk needs and the result is the linear scan's.
// ---------------------------------------------------------------------------------
constexpr uint32_t kPushPerRound = 3u;                         // (1 / 2 / 3 / 4: config #5 band 7.89 / 7.19 / 7.07 / 7.13 ms)
constexpr uint32_t kQueueSlots = 64u + 64u * kPushPerRound;    // fewer than 64 pairs stay behind a drain
// per wave, in float4 slots: the 64-byte lane map, then 64 keys x 8 B, then the queue
constexpr uint32_t kCoopSlotsPerWave = 4u + 32u + kQueueSlots / 4u;
LUMA_DEV uint32_t lane_of_wave() { return threadIdx.x & 63u; }

struct CoopLds {
    uint8_t* map;                   // per wave: 64 bytes, the first lane of each ray's run of lanes marked with its owner
    unsigned long long* key;        // per wave: 64 x ((t bits << 32) | index), the best hit of ray slot q (= the owner's lane)
    uint32_t* queue;                // per wave: kQueueSlots x (walker lane << 22 | ray slot << 16 | sphere); !RES: two planes
};

LUMA_DEV void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <bool RES, typename GridPtr>   // RES: cell offsets and cell lists in LDS as 16-bit words (sphere indices below 65536)
LUMA_DEV void wave_grid_walk(GridPtr G, const uint16_t* __restrict__ lds_cells, const uint16_t* __restrict__ lds_items,
                             const float4* __restrict__ rec_gen, const CoopLds X, uint32_t lane, bool want, f3 origin, f3 dir,
                             int& idx_out, float& t_out, uint32_t& n_tests)
{
    constexpr uint32_t kNone = 0xFFFFFFFFu;
    const uint32_t kMissT = __builtin_bit_cast(uint32_t, FLT_MAX);
    // the reference's test (:244-258) for the ray (o, d) with a = dot(d, d); a hit is returned as a key
    auto test = [&](uint32_t i, f3 o, f3 d, float a, unsigned long long& key) {
        const float4 s = rec_gen[i];
        const f3 diff = sub3(o, mk3(s.x, s.y, s.z));             // :244
        const float b = dot3(diff, d);
        const float c = fma_(-s.w, s.w, dot3(diff, diff));       // :248
        const float dd = fma_(b, b, -(a * c));
        if (LUMA_DBG_WALK == 0) { n_tests++; }
        if (LUMA_DBG_WALK == 3) { LUMA_WAVE_ONCE(n_tests); }
        key = ~0ull;
        if (dd > 0.0f && b < 0.0f) {                             // :253
            const float t = (-b - sqrt_rn(dd)) / a;              // :255
            // :258 with best_t = FLT_MAX at the start: only 0 < t < FLT_MAX can ever win; such a t orders like its bits
            if (t > 0.0f && t < FLT_MAX) { key = ((unsigned long long)__builtin_bit_cast(uint32_t, t) << 32) | i; }
        }
    };

    // ---- own ray: the always-tested spheres, and the part of the ray inside the grid's box ----
    const float a_own = dot3(dir, dir);
    unsigned long long own = ((unsigned long long)kMissT << 32) | kNone;
    float t0 = 0.0f, t1 = FLT_MAX;
    bool inside = want;
    if (want) {
        for (uint32_t j = 0; j < G->n_large; j++) {
            unsigned long long key;
            test(G->large[j], origin, dir, a_own, key);
            own = key < own ? key : own;                         // :258, :266
        }
        auto clip = [&](float o, float d, float org, uint32_t dim) {
            const float lo = org, hi = org + (float)dim * G->h;
            if (d != 0.0f) {
                const float inv = 1.0f / d;
                const float ta = (lo - o) * inv, tb = (hi - o) * inv;
                t0 = __builtin_fmaxf(t0, __builtin_fminf(ta, tb));
                t1 = __builtin_fminf(t1, __builtin_fmaxf(ta, tb));
            } else {
                inside = inside && (o >= lo && o <= hi);
            }
        };
        clip(origin.x, dir.x, G->org[0], G->dims[0]);
        clip(origin.y, dir.y, G->org[1], G->dims[1]);
        clip(origin.z, dir.z, G->org[2], G->dims[2]);
    }
    const bool walk = inside && t0 <= t1 && a_own > 0.0f;

    // ---- hand the rays out: every walking ray gets one lane, and the lanes left over are shared out in proportion
    //      to the cells the rays may cross (their length inside the grid over the cell size, summed over the axes):
    //      adjacent lanes [start, start + cnt) for a ray, lane s of them walks the s-th cnt-th of its range ----
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(walk);
    const uint32_t W = (uint32_t)__builtin_popcountll(mask);
    if (W != 0u) {                                               // wave uniform
        float est = (t1 - t0) * ((__builtin_fabsf(dir.x) + __builtin_fabsf(dir.y) + __builtin_fabsf(dir.z)) * G->inv_h);
        est = walk ? __builtin_fminf(__builtin_fmaxf(est, 1.0f), 1024.0f) : 0.0f;   // (fmaxf drops a NaN)
        float total = est;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { total += __shfl_xor(total, off, 64); }
        // floor of a little less than the exact share: the shares never add up to more than the lanes left over.
        // (Measured against it: equal shares, or a blend of the two, lose 3-12 % on every scene size from 1024 to
        // 65536 spheres; no more than 4 / 8 / 16 lanes per ray lose 1-8 %.)
        const uint32_t cnt = walk ? 1u + (uint32_t)(est * ((float)(64u - W) * 0.999f / total)) : 0u;
        uint32_t start = cnt;                                    // inclusive prefix sum, then exclusive
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)start, off, 64);
            start += lane >= (uint32_t)off ? up : 0u;
        }
        start -= cnt;
        // lane j finds its ray: owners mark the first lane of their run in a byte map, the marks' ballot gives every
        // lane the start of its run, and the ray comes from the owner's registers (ds_bpermute)
        uint8_t* const map = X.map;
        map[lane] = 0;
        wave_lds_sync();
        if (walk) { map[start] = (uint8_t)(0x80u | lane); }
        X.key[lane] = ((unsigned long long)kMissT << 32) | kNone;    // ray slot = the lane that owns the ray
        wave_lds_sync();
        const uint32_t mark = map[lane];
        const unsigned long long heads = __builtin_amdgcn_ballot_w64(mark != 0u);   // bit 0 is always set (the first ray starts at 0)
        const uint32_t head = 63u - (uint32_t)__builtin_clzll(heads & (~0ull >> (63u - lane)));
        const int q = (int)((uint32_t)__shfl((int)mark, (int)head, 64) & 63u);      // the owner's lane = the ray's slot
        const uint32_t seg = lane - head;
        const uint32_t L = (uint32_t)__shfl((int)cnt, q, 64);
        const bool have = seg < L;
        const f3 o = mk3(__shfl(origin.x, q, 64), __shfl(origin.y, q, 64), __shfl(origin.z, q, 64));
        const f3 d = mk3(__shfl(dir.x, q, 64), __shfl(dir.y, q, 64), __shfl(dir.z, q, 64));
        const float r_t0 = __shfl(t0, q, 64), r_t1 = __shfl(t1, q, 64);
        const float a = dot3(d, d);

        // this lane's share of [t0, t1]; a range too long to cut (t1 overflowed) goes to the first lane whole
        const float span = (r_t1 - r_t0) / (float)L;
        const bool cut = span < FLT_MAX;
        const float ts = (cut && seg > 0u) ? fma_(span, (float)seg, r_t0) : r_t0;
        const float te = (cut && seg + 1u < L) ? fma_(span, (float)(seg + 1u), r_t0) : r_t1;
        bool walking = have && (cut || seg == 0u);

        // enter the grid at ts (3D-DDA state; per-axis scalars: no indexed registers)
        float t_cell = FLT_MAX;                                  // time to cross one cell along the fastest axis
        auto enter = [&](float oc, float dc, float org, uint32_t dim, int& cell, float& t_next, float& inv) {
            const float p = fma_(dc, ts, oc);
            cell = min(max((int)__builtin_floorf((p - org) * G->inv_h), 0), (int)dim - 1);
            if (dc != 0.0f) {
                inv = 1.0f / dc;
                t_next = (org + (float)(cell + (dc > 0.0f ? 1 : 0)) * G->h - oc) * inv;
                t_cell = __builtin_fminf(t_cell, G->h * __builtin_fabsf(inv));
            } else {
                inv = 0.0f;
                t_next = FLT_MAX;
            }
        };
        int cx = 0, cy = 0, cz = 0;
        float tx = FLT_MAX, ty = FLT_MAX, tz = FLT_MAX, ix = 0.f, iy = 0.f, iz = 0.f;
        uint32_t k0 = 0, k1 = 0;
        // what the rounds below read of the grid arguments, pinned in vector registers for the length of the walk:
        // left to the register allocator they are re-read from the kernel-argument segment inside the round loop
        // (a scalar load and its wait per use)
        float cell_h = G->h;
        const uint32_t* tab_cells = RES ? nullptr : G->cell_start;
        const uint32_t* tab_items = RES ? nullptr : G->items;
        asm volatile("" : "+v"(cell_h));
        if (!RES) { asm volatile("" : "+v"(tab_cells), "+v"(tab_items)); }
        // (the grid's dimensions as three scalars of their own: selected by axis as G->dims[axis] they became a
        //  load from the kernel-argument segment with a computed address -- a vector memory round trip in every hop)
        uint32_t dim_x = G->dims[0], dim_y = G->dims[1], dim_z = G->dims[2];
        asm volatile("" : "+s"(dim_x), "+s"(dim_y), "+s"(dim_z));
        auto lookup = [&]() {
            const uint32_t id = ((uint32_t)cz * dim_y + (uint32_t)cy) * dim_x + (uint32_t)cx;
            if (RES || lds_cells) { k0 = lds_cells[id]; k1 = lds_cells[id + 1]; }
            else                  { k0 = tab_cells[id]; k1 = tab_cells[id + 1]; }
        };
        if (walking) {
            enter(o.x, d.x, G->org[0], G->dims[0], cx, tx, ix);
            enter(o.y, d.y, G->org[1], G->dims[1], cy, ty, iy);
            enter(o.z, d.z, G->org[2], G->dims[2], cz, tz, iz);
            lookup();
        }
        const float margin = 1e-3f * t_cell;
        uint32_t hops_left = dim_x + dim_y + dim_z + 3u;         // a straight line crosses no more cells
        uint32_t queued = 0;                                     // wave uniform
        const int step_x = d.x > 0.0f ? 1 : -1, step_y = d.y > 0.0f ? 1 : -1, step_z = d.z > 0.0f ? 1 : -1;
        // face crossed after the step to cell c along an axis: t = (org + (c + (d > 0)) * h - o) / d
        const float face_x = G->org[0] + (d.x > 0.0f ? G->h : 0.0f) - o.x;
        const float face_y = G->org[1] + (d.y > 0.0f ? G->h : 0.0f) - o.y;
        const float face_z = G->org[2] + (d.z > 0.0f ? G->h : 0.0f) - o.z;

        for (;;) {
            // ---- walkers whose cell is used up: stop, or cross the nearest face (one per round when two are due at
            //      once).  Branch free, the stop tests included: the lanes of a wave cross different faces and stop for
            //      different reasons.  (The crossing times only steer
            //      the walk and its margins; the tests' arithmetic is untouched by how they are rounded.  Crossing
            //      up to four empty cells in one round was slower: every round then pays for the longest skip.)
            {
                const bool used_up = walking && k0 == k1;
                const float best_t = __builtin_bit_cast(float, (uint32_t)(X.key[q] >> 32));
                const float t_exit = __builtin_fminf(__builtin_fminf(tx, ty), tz);
                const bool stop = best_t < t_exit - margin || t_exit > te + margin || hops_left <= 1u;
                hops_left -= used_up ? 1u : 0u;
                const bool mx = tx <= ty && tx <= tz, my = !mx && ty <= tz, mz = !mx && !my;
                const int cm = (mx ? cx + step_x : my ? cy + step_y : cz + step_z);
                const uint32_t dim = mx ? dim_x : my ? dim_y : dim_z;
                const bool go = used_up && !stop && (uint32_t)cm < dim;      // the walker moves to a cell of the grid
                walking = walking && (!used_up || go);
                const float t_new = fma_((float)cm, cell_h, mx ? face_x : my ? face_y : face_z) * (mx ? ix : my ? iy : iz);
                cx = (go && mx) ? cm : cx;
                cy = (go && my) ? cm : cy;
                cz = (go && mz) ? cm : cz;
                tx = (go && mx) ? t_new : tx;
                ty = (go && my) ? t_new : ty;
                tz = (go && mz) ? t_new : tz;
                if (go) { lookup(); }
            }

            // ---- walkers: push up to kPushPerRound spheres of the current cell ----
            const uint32_t c = walking ? min(k1 - k0, kPushPerRound) : 0u;
            if (LUMA_DBG_WALK == 1 && walking) { n_tests++; }
            if (LUMA_DBG_WALK == 2) { n_tests += lane == 0u ? 1u : 0u; }
#pragma unroll
            for (uint32_t j = 0; j < kPushPerRound; j++) {
                const unsigned long long m = __builtin_amdgcn_ballot_w64(c > j);
                if (c > j) {
                    const uint32_t at = queued + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    const uint32_t i = RES ? (uint32_t)lds_items[k0 + j] : tab_items[k0 + j];
                    const uint32_t who = (lane << 6) | (uint32_t)q;      // walker lane (the ray is in its registers), ray slot
                    X.queue[at] = RES ? ((who << 16) | i) : i;
                    if (!RES) { X.queue[kQueueSlots + at] = who; }
                }
                queued += (uint32_t)__builtin_popcountll(m);
            }
            k0 += c;
            const bool any = __builtin_amdgcn_ballot_w64(walking) != 0ull;

            // ---- testers: one pair per lane while a wave's worth is queued, or everything once the walkers are done ----
            while (queued >= 64u || (!any && queued > 0u)) {
                wave_lds_sync();
                const uint32_t take = min(queued, 64u), base = queued - take;
                uint32_t who = lane << 6, i = 0;
                if (lane < take) {
                    const uint32_t pr = X.queue[base + lane];
                    if (RES) { who = pr >> 16; i = pr & 0xFFFFu; }
                    else     { who = X.queue[kQueueSlots + base + lane]; i = pr; }
                }
                const int owner = (int)(who >> 6);
                const f3 po = mk3(__shfl(o.x, owner, 64), __shfl(o.y, owner, 64), __shfl(o.z, owner, 64));
                const f3 pd = mk3(__shfl(d.x, owner, 64), __shfl(d.y, owner, 64), __shfl(d.z, owner, 64));
                const float pa = __shfl(a, owner, 64);
                if (lane < take) {
                    unsigned long long key;
                    test(i, po, pd, pa, key);
                    if (key != ~0ull) { atomicMin(&X.key[who & 63u], key); }
                }
                queued = base;
                wave_lds_sync();
            }
            if (!any) { break; }                                 // wave uniform; the queue is empty here
        }

        // ---- back to the rays' own lanes ----
        if (walk) {
            const unsigned long long found = X.key[lane];
            own = found < own ? found : own;
        }
    }
    idx_out = (int)(uint32_t)own;                                // kNone reads as -1
    t_out = __builtin_bit_cast(float, (uint32_t)(own >> 32));
}

template <int TILE_W, int WAVES, bool NOISE, bool FROM_HIT, bool GRID = false, int RES = 0>
__global__ __launch_bounds__(WAVES * 64) void render_kernel_sp(const RenderArgs A)
{
    // [0,n_pad): camera-origin records; [n_pad,2 n_pad): general records; padding = NaN
    // GRID: the camera-origin records stay in global memory; LDS holds the uniform grid's cell offsets
    // as 16-bit words (when launch_render found room) and each wave's exchange area for wave_grid_walk.
    // GRID && RES (the whole walk is LDS resident: launch_render found room for all of it): the general
    // records [0,n), then the cell offsets and the cells' sphere lists as 16-bit words, then the
    // exchange areas -- a hop of the walk and a sphere test wait on LDS, not on L2, round trips.
    extern __shared__ float4 lds_rec[];
    const uint32_t n = A.n_spheres;
    const uint32_t n_pad = (n + 63u) & ~63u;
    const float4* rec_cam = lds_rec;
    const float4* rec_gen = lds_rec + n_pad;
    const uint16_t* grid_cells = nullptr;
    const uint16_t* grid_items = nullptr;
    CoopLds xch = { nullptr, nullptr, nullptr };
    if (GRID) {
        rec_cam = A.records;
        rec_gen = A.records + n;
        float4* next = lds_rec;
        if (RES == 2) {
            for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) { next[i] = A.records[n + i]; }
            rec_gen = next;
            next += n;
        }
        const uint32_t n_cells = RES ? A.grid.n_cells : A.grid.lds_cells;
        if (n_cells) {
            uint16_t* stage = reinterpret_cast<uint16_t*>(next);
            for (uint32_t i = threadIdx.x; i < n_cells; i += blockDim.x) { stage[i] = (uint16_t)A.grid.cell_start[i]; }
            grid_cells = stage;
            next += (n_cells + 7u) / 8u;                         // float4 slots of 16-bit words
        }
        if (RES) {
            uint16_t* stage = reinterpret_cast<uint16_t*>(next);
            for (uint32_t i = threadIdx.x; i < A.grid.n_items; i += blockDim.x) { stage[i] = (uint16_t)A.grid.items[i]; }
            grid_items = stage;
            next += (A.grid.n_items + 7u) / 8u;
        }
        float4* mine = next + (threadIdx.x >> 6) * (RES ? kCoopSlotsPerWave : kCoopSlotsPerWave + kQueueSlots / 4u);
        xch.map = reinterpret_cast<uint8_t*>(mine);
        xch.key = reinterpret_cast<unsigned long long*>(mine + 4);
        xch.queue = reinterpret_cast<uint32_t*>(mine + 36);     // !RES: a second plane of kQueueSlots words holds the walker lanes
        __syncthreads();
    } else {
        const float qnan = __builtin_nanf("");
        const float4 never = make_float4(qnan, qnan, qnan, qnan);
        for (uint32_t i = threadIdx.x; i < n_pad; i += blockDim.x) {
            float4 c = never, g = never;
            if (i < n) { c = A.records[i]; g = A.records[n + i]; }
            lds_rec[i] = c;
            lds_rec[n_pad + i] = g;
        }
        __syncthreads();
    }
    uint32_t ntests = 0;
    // Renderer::trace_ray (:234-287) for the wave: a ray from camera.pos / from anywhere, for the lanes that `want` one
    constexpr int kScanRays = (WAVES == 16) ? 4 : 1;             // rays per pass of the linear scan (wave_scan_many)
    auto scan_cam = [&](bool want, f3 o, f3 d, int& idx, float& t) {
        if (GRID) {
            idx = -1;
            t = FLT_MAX;
            if (want) {                                          // direction grid: no walk, one lane per ray
                const auto G = kernarg_grid();
                idx = RES == 2 ? scan_view<true>(G, rec_gen, o, d, t, ntests) : scan_view<false>(G, rec_cam, o, d, t, ntests);
            }
        } else {
            wave_scan<true, kScanRays>(rec_cam, n_pad, lane_of_wave(), want, o, d, idx, t);
        }
    };
    auto scan_gen = [&](bool want, f3 o, f3 d, int& idx, float& t) {
        if (GRID) {
            wave_grid_walk<(RES != 0)>(kernarg_grid(), grid_cells, grid_items, rec_gen, xch, lane_of_wave(), want, o, d, idx, t, ntests);
        }
        else      { wave_scan<false, kScanRays>(rec_gen, n_pad, lane_of_wave(), want, o, d, idx, t); }
    };

    // Persistent waves: the grid is one resident set of workgroups (the 128-KiB record image
    // allows one 16-wave workgroup per CU), and each wave pulls 16x4-pixel tiles from a global
    // counter until none are left.  The records are staged once per workgroup instead of once
    // per 1024 pixels, and a CU never idles waiting for the slowest wave of a workgroup.
    constexpr int TILE_H = 64 / TILE_W;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t tiles_x = (A.width + TILE_W - 1) / TILE_W, tiles_y = (A.rows + TILE_H - 1) / TILE_H;
    const uint32_t n_tiles = tiles_x * tiles_y;
    // The last tiles are handed out one row (TILE_W pixels) at a time: a wave scans for a few rays at a time whatever
    // the number of its pixels, so a row costs a quarter of a tile, and the launch drains for a quarter as long.
    // About half a round of the resident waves; a frame of fewer tiles than that goes row by row throughout.
    // Config #5 band 153.8 -> 150.6 ms, 640x360 x 16 spp of 2000 spheres 7.26 -> 5.12 (tools/ab_config5_both.sh).  Not
    // under GRID: a wave walks the grid for all its rays together, and a quarter of the rays is less than a quarter
    // of the work (the same band 5.99 -> 6.31 ms, the small frame 0.60 -> 0.93).
    const uint32_t tail = GRID ? 0u : min(n_tiles, (gridDim.x * WAVES) / 2u);
    const uint32_t n_whole = n_tiles - tail, n_items = n_whole + tail * TILE_H;
    const f3 campos = mk3(A.cam_pos[0], A.cam_pos[1], A.cam_pos[2]);
    const uint32_t slots = 2u + A.paths;
    const bool pathtrace = A.mode == MODE_PATHTRACE;
    uint32_t nrays = 0;

  for (;;) {
    uint32_t item = 0;
    if (lane == 0) { item = atomicAdd(A.tile_counter, 1u); }
    item = (uint32_t)__builtin_amdgcn_readfirstlane((int)item);
    if (item >= n_items) { break; }                              // every wave reaches this: the counter only grows
    const bool by_row = item >= n_whole;
    const uint32_t tile = by_row ? n_whole + (item - n_whole) / TILE_H : item;
    const uint32_t row = by_row ? (item - n_whole) % TILE_H : 0u;

    const uint32_t x = (tile % tiles_x) * TILE_W + (lane % TILE_W);
    const uint32_t yl = (tile / tiles_x) * TILE_H + (lane / TILE_W);
    const uint32_t y = global_row(A, yl);
    const bool live = (x < A.width) && (yl < A.rows) && (!by_row || lane / TILE_W == row);

    const uint32_t pixel = y * A.width + x;                      // global index (:399); unused when !live
    const uint32_t local = yl * A.width + x;

    // (the two matrices come from the kernel-argument segment here, once per tile, instead of living in SGPRs or
    //  spill lanes across the scans and walks; so do the accumulator and target pointers at the end of the tile)
    const auto KA = kernarg_args();
    const f3 dir0 = camera_ray(KA, live ? x : 0u, live ? y : 0u); // camera.rays[index], :413

    // DOF pre-pass, :416-432
    int h0;
    float t0;
    scan_cam(live, campos, dir0, h0, t0);
    nrays += live ? 1u : 0u;
    const float depth = (h0 >= 0) ? t0 : FLT_MAX;                // :419-423
    const float dd = depth - A.focus_depth;                      // :425
    const float x2 = dd * dd, x4 = x2 * x2, x8 = x4 * x4, x10 = x8 * x2;
    const float focus = 1.0f - luma_expf(-x10);                  // :427
    const float blur = fma_(focus, 4.5f, 0.5f);                  // :429-432
    const float gainf = (A.show_depth && (focus < .5f)) ? 1.2f : 1.0f;   // :448-452

    float acc_r = 0.f, acc_g = 0.f, acc_b = 0.f;
    if (live) { acc_r = KA->acc[0][local]; acc_g = KA->acc[1][local]; acc_b = KA->acc[2][local]; }

    uint32_t s = 0, bounce = 0;
    f3 direct = mk3(0.f, 0.f, 0.f);
    f3 dir = dir0;                                               // :293
    f3 origin = campos;                                          // :294, :303
    const uint32_t spp = (A.bounces && live) ? A.spp : 0u;       // bounces == 0: every sample adds +0

    // One (sample, bounce) step per trip for every lane that still has samples; the loop itself
    // is wave uniform because the scans are cooperative.
    while (__builtin_amdgcn_ballot_w64(s < spp)) {
        const bool active = s < spp;
        const uint32_t sample = A.frame_index + s;
        const uint32_t slot = bounce * slots;

        f3 dn = dir;
        if (active) {
            if (NOISE) { dn = add3(dir, rng_vec3(A, pixel, sample, slot, -blur, blur)); }   // :118-119, :302
            else       { dn = add3(dir, mk3(0.f, 0.f, 0.f)); }
        }

        int idx;
        float t;
        if (FROM_HIT) {
            int i2; float t2;
            scan_cam(active && bounce == 0, origin, dn, idx, t);
            scan_gen(active && bounce > 0, origin, dn, i2, t2);
            if (bounce > 0) { idx = i2; t = t2; }
        } else {
            scan_cam(active, origin, dn, idx, t);                // :304
        }
        nrays += active ? 1u : 0u;

        // shading up to the point where indirect rays are needed
        bool path_ends = false, want_indirect = false, lit = false;
        Hit hit; hit.pos = campos; hit.normal = mk3(0.f, 0.f, 1.f);
        float4 m0 = make_float4(0.f, 0.f, 0.f, 0.f);
        float2 m1 = make_float2(0.f, 0.f);
        float cos_theta = 0.f;
        if (active) {
            if (idx < 0) {                                       // :306-319
                const float clamped = __builtin_fminf(__builtin_fmaxf(dn.y, -1.f), 1.f);
                const float adjusted = (clamped + 1.f) * .5f;
                direct = add3(direct, color_lerp(mk3(.529f, .808f, .922f), mk3(.106f, .275f, .711f), adjusted));
                path_ends = true;
            } else {
                hit = finish_hit(rec_gen, idx, origin, dn, t);
                m0 = A.mat0[idx];                                // diffuse rgb, albedo
                m1 = A.mat1[idx];                                // metallic, roughness
                cos_theta = dot3(neg3(dn), hit.normal);          // :331-332
                lit = cos_theta >= 0.0f;                         // :334
                want_indirect = lit && (m0.w > 0.0f) && pathtrace;   // :337, :187
            }
        }

        // indirect_illumination, :182-216: `paths` cooperative scans from the hit points
        f3 indirect = mk3(0.f, 0.f, 0.f);
        for (uint32_t p = 0; p < A.paths; p++) {                 // :192
            f3 nd = mk3(0.f, 0.f, 1.f);
            if (want_indirect) {
                f3 d = normalize3(rng_vec3(A, pixel, sample, slot + 2u + p, -1.0f, 1.0f));   // :194
                if (dot3(d, hit.normal) < 0.0f) { d = neg3(d); } // :197-200
                nd = add3(hit.normal, d);                        // :202
            }
            int ci; float tc;
            scan_gen(want_indirect, hit.pos, nd, ci, tc);        // :204
            if (want_indirect) {
                nrays++;
                if (ci >= 0) {                                   // :206-209
                    const float4 cm = A.mat0[ci];
                    indirect = add3(indirect, mk3(cm.x, cm.y, cm.z));
                }
            }
        }

        if (active && idx >= 0) {
            if (lit) {
                if (m0.w > 0.0f) {
                    indirect = scale3(indirect, A.inv_samples);  // :212-213
                    indirect = scale3(indirect, m0.w);           // :340
                }
                const float value = (cos_theta + 1.f) * .5f;     // :343
                const float scalar = value * (1.f - m1.x);       // :345
                const f3 capture = color_lerp(mk3(m0.x, m0.y, m0.z), indirect, 1.f - scalar);   // :346
                direct = fma3(capture, scalar * .8f, direct);    // :347-349
            }
            bounce++;
            path_ends = (m1.x == 0.0f) || (bounce == A.bounces); // :353-357, :300
            if (!path_ends) {
                f3 nrm = hit.normal;
                if (NOISE) {
                    const float gain = .2f * m1.y;               // :225
                    nrm = add3(nrm, rng_vec3(A, pixel, sample, slot + 1u, -gain, gain));   // :226-227
                } else {
                    nrm = add3(nrm, mk3(0.f, 0.f, 0.f));
                }
                const float k = 2.0f * dot3(dn, nrm);            // fx::reflect, :229
                dir = fma3(nrm, -k, dn);                         // :359
                if (FROM_HIT) { origin = fma3(hit.normal, .001f, hit.pos); }   // :221-222
            }
        }

        if (active && path_ends) {
            const f3 result = scale3(tonemap3(direct), gainf);   // :362, :57-70; :452 (x * 1.0f is x)
            acc_r += result.x; acc_g += result.y; acc_b += result.z;   // :457-458
            s++;
            bounce = 0;
            direct = mk3(0.f, 0.f, 0.f);
            dir = dir0;
            origin = campos;
        }
    }

    const auto E = kernarg_args();
    if (live) {
        E->acc[0][local] = acc_r; E->acc[1][local] = acc_g; E->acc[2][local] = acc_b;
        const float inv = 1.0f / E->frame_count_last;            // :460
        int r = (int)((acc_r * inv) * 255.0f);                   // ::RGB, :44-55
        int g = (int)((acc_g * inv) * 255.0f);
        int b = (int)((acc_b * inv) * 255.0f);
        const uint32_t flags = E->flags;
        if (flags & FLAG_CLAMP_PACK) {
            r = min(max(r, 0), 255); g = min(max(g, 0), 255); b = min(max(b, 0), 255);
        }
        const uint32_t px = pack_pixel(r, g, b, flags);
        E->target[local] = px;                                   // :461
        uint32_t* const mirror = E->target_mirror;
        if (mirror) { mirror[E->mirror_is_frame ? frame_pixel(E, local) : local] = px; }
    }
  }  // next tile

    wave_add_u64(A.counters, nrays);
    if (GRID) { wave_add_u64(A.counters + 1, ntests); wave_add_u64(A.counters + 2, nrays); }   // tests, and rays of grid launches
}

// Pixels per wave: 16x4 tiles in the wave-uniform kernels, 32x2 in the ray-per-lane kernel (whole 1080p frame the
// same either way; bands of 4 / 8 of it 0.527 -> 0.505, 0.283 -> 0.277 ms; the wave-uniform kernels 1 % slower at 32x2)
static constexpr int kTileW = 16;
static constexpr int kLaneTileW = 32;

}  // namespace luma
