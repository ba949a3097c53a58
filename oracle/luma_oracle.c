/* oracle/luma_oracle.c -- TEST INFRASTRUCTURE ONLY (see luma_oracle.h for the header note:
 * who may load this, why parity is "unpinned" at the flux boundary, and the arithmetic
 * contract).  Plain C99, no dependencies beyond libm.  Build: oracle/Makefile.
 *
 * Every function cites the reference lines it restates (paths relative to /root/reference).
 * Where the reference calls into the absent `flux` library the inferred semantics of
 * SURVEY.md section 8a-0 are used and marked [flux-inferred].
 */
#include "luma_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* Two builds of this file (oracle/Makefile):
 *   default      the arithmetic CONTRACT the HIP kernels share: every multiply-add written FMA()
 *                below is one fused operation, normalize multiplies by 1/sqrt, exp and the DOF
 *                term are fixed float evaluations.  libluma_oracle.so, the parity checker.
 *   -DLO_LITERAL the reference's expressions taken literally, with none of those choices: no
 *                fusion anywhere (a*b+c is two roundings), normalize divides by sqrtf, tonemap
 *                calls libm expf (std::exp(float), Renderer.cpp:61), and the DOF term runs in
 *                double exactly as Renderer.cpp:427-432 writes it (std::pow / std::exp /
 *                std::lerp on promoted operands, `focus < .5f` compared in double).
 *                libluma_oracle_literal.so: only tests/golden/make_contract_sensitivity.py and
 *                tests/test_contract_sensitivity.py load it, to MEASURE how far frames move when
 *                the contract's choices flip (the reference is built /fp:fast, which leaves them
 *                open: raytracer.vcxproj:187-190). */
#ifdef LO_LITERAL
#define FMA(a, b, c) ((a) * (b) + (c))
#else
#define FMA(a, b, c) fmaf((a), (b), (c))
#endif

typedef struct { float x, y, z; } v3;

/* ---------------------------------------------------------------------------------
 * fx:: vector helpers [flux-inferred]: elementwise, one rounding per operation.
 * ------------------------------------------------------------------------------- */
static v3 v3_make(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static v3 v3_scale(v3 a, float s) { return v3_make(a.x * s, a.y * s, a.z * s); }
static v3 v3_neg(v3 a) { return v3_make(-a.x, -a.y, -a.z); }                 /* fx::invert */
/* fx::dot: x first, then y and z folded in with one rounding each. */
static float v3_dot(v3 a, v3 b) { return FMA(a.z, b.z, FMA(a.y, b.y, a.x * b.x)); }
/* fx::normalize: v * (1 / sqrt(dot(v,v))). */
#ifdef LO_LITERAL
static v3 v3_normalize(v3 a) { float len = sqrtf(v3_dot(a, a)); return v3_make(a.x / len, a.y / len, a.z / len); }
#else
static v3 v3_normalize(v3 a) { float inv = 1.0f / sqrtf(v3_dot(a, a)); return v3_scale(a, inv); }
#endif
/* fx::reflect(d, n) = d - 2*dot(d,n)*n, one rounding per component. */
static v3 v3_reflect(v3 d, v3 n)
{
    float k = 2.0f * v3_dot(d, n);
    return v3_make(FMA(-k, n.x, d.x), FMA(-k, n.y, d.y), FMA(-k, n.z, d.z));
}
/* ::cross, Camera.cpp:64-72 */
static v3 v3_cross(v3 a, v3 b)
{
    return v3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

/* ---------------------------------------------------------------------------------
 * exp() used by tonemap (Renderer.cpp:61) and the focus term (Renderer.cpp:427).
 * The reference calls std::exp; to be bit-reproducible on CPU and GPU the contract
 * fixes one float implementation: Cody-Waite reduction by ln2 and a degree-5 minimax
 * polynomial (Cephes expf coefficients), Horner with fmaf.  x < -87 returns 0, x > 88
 * returns +inf.  Max error vs exp() is checked in tests/test_oracle_kat.py (<= 2 ulp).
 * ------------------------------------------------------------------------------- */
float lo_expf(float x)
{
#ifdef LO_LITERAL
    return expf(x);
#endif
    if (!(x >= -87.0f)) { return (x != x) ? x : 0.0f; }
    if (x > 88.0f) { return INFINITY; }
    float kf = rintf(x * 1.44269504088896341f);
    float r = FMA(kf, -0.693359375f, x);
    r = FMA(kf, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = FMA(p, r, 1.3981999507e-3f);
    p = FMA(p, r, 8.3334519073e-3f);
    p = FMA(p, r, 4.1665795894e-2f);
    p = FMA(p, r, 1.6666665459e-1f);
    p = FMA(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float e = FMA(p, r2, r) + 1.0f;
    /* scale by 2^k through the exponent field; k is in [-126, 127] here */
    int32_t k = (int32_t)kf;
    union { float f; uint32_t u; } s;
    s.u = (uint32_t)(k + 127) << 23;
    return e * s.f;
}

/* ---------------------------------------------------------------------------------
 * Counter-based RNG.  The reference draws from a global stateful fx::Random whose
 * algorithm and seed are unknown; BASELINE.json mandates a counter-based
 * generator shared by CPU and GPU.  Contract: pcg3d (Jarzynski & Olano, "Hash Functions
 * for GPU Rendering", JCGT 9(3), 2020) over the counter (pixel, sample, slot) XORed with
 * three 32-bit keys derived from the 64-bit seed by two splitmix64 steps.
 * ------------------------------------------------------------------------------- */
void lo_pcg3d(uint32_t x, uint32_t y, uint32_t z, uint32_t out[3])
{
    x = x * 1664525u + 1013904223u;
    y = y * 1664525u + 1013904223u;
    z = z * 1664525u + 1013904223u;
    x += y * z; y += z * x; z += x * y;
    x ^= x >> 16; y ^= y >> 16; z ^= z >> 16;
    x += y * z; y += z * x; z += x * y;
    out[0] = x; out[1] = y; out[2] = z;
}

static uint64_t splitmix64(uint64_t* state)
{
    uint64_t z = (*state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void lo_seed_keys(uint64_t seed, uint32_t keys[3])
{
    uint64_t s = seed;
    uint64_t a = splitmix64(&s);
    uint64_t b = splitmix64(&s);
    keys[0] = (uint32_t)a;
    keys[1] = (uint32_t)(a >> 32);
    keys[2] = (uint32_t)b;
}

/* u32 -> [0,1): top 24 bits, exact. */
static float u01(uint32_t bits) { return (float)(bits >> 8) * 5.9604644775390625e-8f; }

/* fx::Random::vec3(lo, hi) [flux-inferred]: three independent uniforms in [lo,hi):
 * lo + k * ((hi-lo) * 2^-24) with k the top 24 bits of each hash word, one rounding. */
void lo_rng_vec3(const uint32_t keys[3], uint32_t pixel, uint32_t sample, uint32_t slot,
                 float lo, float hi, float out[3])
{
    uint32_t h[3];
    lo_pcg3d(pixel ^ keys[0], sample ^ keys[1], slot ^ keys[2], h);
    float w = (hi - lo) * 5.9604644775390625e-8f;
    out[0] = FMA((float)(h[0] >> 8), w, lo);
    out[1] = FMA((float)(h[1] >> 8), w, lo);
    out[2] = FMA((float)(h[2] >> 8), w, lo);
}

/* fx::Random::vec3_sphere() [flux-inferred, low confidence]: normalize(vec3(-1,1)). */
void lo_rng_vec3_sphere(const uint32_t keys[3], uint32_t pixel, uint32_t sample, uint32_t slot,
                        float out[3])
{
    float c[3];
    lo_rng_vec3(keys, pixel, sample, slot, -1.0f, 1.0f, c);
    v3 n = v3_normalize(v3_make(c[0], c[1], c[2]));
    out[0] = n.x; out[1] = n.y; out[2] = n.z;
}

/* ---------------------------------------------------------------------------------
 * Renderer.cpp helpers
 * ------------------------------------------------------------------------------- */

/* ::tonemap, Renderer.cpp:57-70 */
void lo_tonemap(const float in[3], float out[3])
{
    out[0] = 1.0f - lo_expf(-in[0]);
    out[1] = 1.0f - lo_expf(-in[1]);
    out[2] = 1.0f - lo_expf(-in[2]);
}

/* ::RGB, Renderer.cpp:44-55: scale by 255, truncate, shift, OR; no clamp. */
uint32_t lo_pack_rgb(const float in[3], uint32_t flags)
{
    int32_t r = (int32_t)(in[0] * 255.0f);
    int32_t g = (int32_t)(in[1] * 255.0f);
    int32_t b = (int32_t)(in[2] * 255.0f);
    if (flags & LO_FLAG_CLAMP_PACK) {
        r = r < 0 ? 0 : (r > 255 ? 255 : r);
        g = g < 0 ? 0 : (g > 255 ? 255 : g);
        b = b < 0 ? 0 : (b > 255 ? 255 : b);
    }
    uint32_t out = 0;
    if (flags & LO_FLAG_PACK_RGBA8) {
        out |= (uint32_t)r << 0;
        out |= (uint32_t)g << 8;
        out |= (uint32_t)b << 16;
        out |= 0xFF000000u;
        return out;
    }
    out |= (uint32_t)r << 16;
    out |= (uint32_t)g << 8;
    out |= (uint32_t)b << 0;
    return out;
}

/* ::lerp, Renderer.cpp:72-83: scale(c1, 1-t) + scale(c2, t), the add fused. */
static v3 color_lerp(v3 c1, v3 c2, float t)
{
    float t1 = 1.0f - t;
    return v3_make(FMA(c2.x, t, c1.x * t1), FMA(c2.y, t, c1.y * t1), FMA(c2.z, t, c1.z * t1));
}

/* DOF pre-pass arithmetic, Renderer.cpp:425-432.  The reference evaluates
 * 1 - exp(-pow(dd, 10)) and std::lerp(.5, 5, focus) in double; the contract restates
 * them in float (dd^10 by four multiplies, lo_expf) so that CPU and GPU agree bit for
 * bit.  The deviation from the double evaluation is bounded in tests/test_oracle_kat.py. */
/* *in_focus_out = the `focus < .5f` decision of Renderer.cpp:450 (taken in double in the reference
 * because `focus` is a double there). */
static float focus_term(float depth, float focus_depth, float* blur_out, int* in_focus_out)
{
#ifdef LO_LITERAL
    const float depth_difference = depth - focus_depth;                       /* :425 */
    const double focus = 1.f - exp(-pow((double)depth_difference, 10));       /* :427 */
    const double a = .5f, b = 5.f;
    /* std::lerp(a, b, t) for finite a, b of the same sign: a + t * (b - a), exact at t == 1 */
    const double blur = (focus == 1.0) ? b : a + focus * (b - a);             /* :432 */
    if (blur_out) { *blur_out = (float)blur; }                                /* :434 static_cast */
    if (in_focus_out) { *in_focus_out = focus < .5f; }                        /* :450 */
    return (float)focus;
#else
    float blur;
    const float focus = lo_focus(depth, focus_depth, &blur);
    if (blur_out) { *blur_out = blur; }
    if (in_focus_out) { *in_focus_out = focus < .5f; }
    return focus;
#endif
}

float lo_focus(float depth, float focus_depth, float* blur_out)
{
#ifdef LO_LITERAL
    return focus_term(depth, focus_depth, blur_out, NULL);
#endif
    float dd = depth - focus_depth;
    float x2 = dd * dd;
    float x4 = x2 * x2;
    float x8 = x4 * x4;
    float x10 = x8 * x2;
    float focus = 1.0f - lo_expf(-x10);
    if (blur_out) { *blur_out = FMA(focus, 4.5f, 0.5f); }
    return focus;
}

/* ---------------------------------------------------------------------------------
 * Renderer::trace_ray, Renderer.cpp:234-287; miss() sentinel Renderer.cpp:172-175.
 * ------------------------------------------------------------------------------- */
typedef struct {
    v3 pos, normal;
    float distance, exit;
    int32_t object; /* index into the sphere vector, -1 == nullptr */
} isect_t;

typedef struct {
    const lo_sphere* spheres;
    uint32_t n;
    lo_stats* stats;
} scene_t;

static isect_t trace_ray(const scene_t* sc, v3 pos, v3 dir)
{
    float distance = FLT_MAX;
    isect_t hit;
    memset(&hit, 0, sizeof hit);
    hit.object = -1;

    if (sc->stats) { sc->stats->rays += 1; sc->stats->sphere_tests += sc->n; }

    for (uint32_t i = 0; i < sc->n; i++) {
        const lo_sphere* s = &sc->spheres[i];
        v3 centre = v3_make(s->pos[0], s->pos[1], s->pos[2]);
        v3 diff = v3_sub(pos, centre);                               /* :244 */

        float a = v3_dot(dir, dir);                                  /* :246 */
        float b = 2.0f * v3_dot(diff, dir);                          /* :247 */
        float c = FMA(-s->radius, s->radius, v3_dot(diff, diff));    /* :248 */

        float d = FMA(b, b, -((4.0f * a) * c));                      /* :251 */

        if (d > 0.0f) {                                              /* :253 */
            float sq = sqrtf(d);
            float t = (-b - sq) / (2.0f * a);                        /* :255 */
            float e = (-b + sq) / (2.0f * a);                        /* :256 */
            if (t > 0.0f) {                                          /* :258 */
                v3 p = v3_make(FMA(dir.x, t, pos.x), FMA(dir.y, t, pos.y), FMA(dir.z, t, pos.z)); /* :260-261 */
                v3 normal = v3_normalize(v3_sub(p, centre));         /* :263-264 */
                if (t < distance) {                                  /* :266 */
                    distance = t;
                    hit.pos = p; hit.normal = normal; hit.distance = t; hit.exit = e;
                    hit.object = (int32_t)i;
                }
            }
        }
    }

    if (distance == FLT_MAX) {                                       /* :276-279 -> miss() :172-175 */
        isect_t m;
        m.pos = v3_make(.6f, .7f, .95f);
        m.normal = v3_make(1.f, 1.f, 1.f);
        m.distance = 1.f; m.exit = 0.f; m.object = -1;
        return m;
    }
    return hit;
}

typedef struct {
    scene_t scene;
    const lo_camera* cam;
    const lo_params* params;
    uint32_t keys[3];
    uint32_t slots_per_bounce;  /* 2 + paths */
} frame_t;

/* Renderer::indirect_illumination, Renderer.cpp:182-216 */
static v3 indirect_illumination(const frame_t* f, const isect_t* is,
                                uint32_t pixel, uint32_t sample, uint32_t slot_base)
{
    v3 out = v3_make(0.f, 0.f, 0.f);
    if (f->params->mode != 1u) { return out; }                       /* :187-190 */

    for (uint32_t s = 0; s < f->params->paths; s++) {                /* :192 */
        float d[3];
        lo_rng_vec3_sphere(f->keys, pixel, sample, slot_base + s, d);/* :194 */
        v3 dir = v3_make(d[0], d[1], d[2]);
        if (v3_dot(dir, is->normal) < 0.0f) { dir = v3_neg(dir); }   /* :197-200 */
        v3 new_dir = v3_add(is->normal, dir);                        /* :202 */
        isect_t cast = trace_ray(&f->scene, is->pos, new_dir);       /* :203-204 */
        if (cast.object >= 0) {                                      /* :206-209 */
            const float* df = f->scene.spheres[cast.object].diffuse;
            out = v3_add(out, v3_make(df[0], df[1], df[2]));
        }
    }
    float divisor = 1.0f / (float)f->params->samples;                /* :212 */
    return v3_scale(out, divisor);                                   /* :213 */
}

/* Renderer::render_pixel, Renderer.cpp:289-364 (+ ::noise :116-122,
 * Renderer::reflect_intersection :218-232).  RNG slots per bounce: 0 = direction noise,
 * 1 = roughness noise, 2.. = indirect path samples; fixed so the stream does not depend
 * on control flow. */
static v3 render_pixel(const frame_t* f, v3 primary_dir, uint32_t pixel, uint32_t sample, float blur)
{
    const lo_params* P = f->params;
    const int noise_off = (P->flags & LO_FLAG_NOISE_OFF) != 0;
    v3 campos = v3_make(f->cam->pos[0], f->cam->pos[1], f->cam->pos[2]);

    v3 direct = v3_make(0.f, 0.f, 0.f);
    v3 dir = primary_dir;                                            /* :293 */
    v3 origin = campos;                                              /* :294, :303 */

    for (uint32_t bounce = 0; bounce < P->bounces; bounce++) {       /* :300 */
        uint32_t slot = bounce * f->slots_per_bounce;
        v3 n = v3_make(0.f, 0.f, 0.f);
        if (!noise_off) {
            float r[3];
            lo_rng_vec3(f->keys, pixel, sample, slot + 0, -blur, blur, r); /* :118 */
            n = v3_make(r[0], r[1], r[2]);
        }
        v3 dir_noised = v3_add(dir, n);                              /* :119, :302 */
        isect_t is = trace_ray(&f->scene, origin, dir_noised);       /* :303-304 */

        if (is.object < 0) {                                         /* :306-319 */
            const v3 top = v3_make(.529f, .808f, .922f);
            const v3 bottom = v3_make(.106f, .275f, .711f);
            float clamped = fminf(fmaxf(dir_noised.y, -1.f), 1.f);
            float adjusted = (clamped + 1.f) * .5f;
            direct = v3_add(direct, color_lerp(top, bottom, adjusted));
            break;
        }

        const lo_sphere* obj = &f->scene.spheres[is.object];
        v3 diffuse = v3_make(obj->diffuse[0], obj->diffuse[1], obj->diffuse[2]);
        float cos_theta = v3_dot(v3_neg(dir_noised), is.normal);     /* :331-332 */

        if (cos_theta >= 0.0f) {                                     /* :334 */
            v3 indirect = v3_make(0.f, 0.f, 0.f);
            if (obj->albedo > 0.0f) {                                /* :337-341 */
                indirect = indirect_illumination(f, &is, pixel, sample, slot + 2);
                indirect = v3_scale(indirect, obj->albedo);
            }
            float value = (cos_theta + 1.f) * .5f;                   /* :343 */
            float scalar = value * (1.f - obj->metallic);            /* :345 */
            v3 capture = color_lerp(diffuse, indirect, 1.f - scalar);/* :346 */
            float k = scalar * .8f;                                  /* :347 */
            direct = v3_make(FMA(capture.x, k, direct.x), FMA(capture.y, k, direct.y),
                             FMA(capture.z, k, direct.z));           /* :349 */
        }

        if (obj->metallic == 0.0f) { break; }                        /* :353-357 */

        /* reflect_intersection, :218-232 (only .dir survives at :359 unless ORIGIN_FROM_HIT) */
        v3 rn = v3_make(0.f, 0.f, 0.f);
        if (!noise_off) {
            float gain = .2f * obj->roughness;                       /* :225 */
            float r[3];
            lo_rng_vec3(f->keys, pixel, sample, slot + 1, -gain, gain, r); /* :226 */
            rn = v3_make(r[0], r[1], r[2]);
        }
        v3 normal = v3_add(is.normal, rn);                           /* :227 */
        dir = v3_reflect(dir_noised, normal);                        /* :229, :359 */
        if (P->flags & LO_FLAG_ORIGIN_FROM_HIT) {
            origin = v3_make(FMA(is.normal.x, .001f, is.pos.x), FMA(is.normal.y, .001f, is.pos.y),
                             FMA(is.normal.z, .001f, is.pos.z));     /* :221-222 */
        }
    }

    float in[3] = { direct.x, direct.y, direct.z }, out[3];
    lo_tonemap(in, out);                                             /* :362 */
    return v3_make(out[0], out[1], out[2]);
}

/* ---------------------------------------------------------------------------------
 * Camera.cpp
 * ------------------------------------------------------------------------------- */
#define M(m, i, j) ((m)[(i) * 4 + (j)])

static float det2(const float* m, int r0, int r1, int c0, int c1)
{
    return M(m, r0, c0) * M(m, r1, c1) - M(m, r0, c1) * M(m, r1, c0);
}
static float tri(float x, float A, float y, float B, float z, float C)
{
    return x * A - y * B + z * C;
}

/* ::inverse, Camera.cpp:14-62 (cofactor expansion over 2x2 minors A{cols}{rows}). */
void lo_mat4_inverse(const float m[16], float out[16])
{
    const float A2323 = det2(m, 2, 3, 2, 3), A1323 = det2(m, 2, 3, 1, 3), A1223 = det2(m, 2, 3, 1, 2);
    const float A0323 = det2(m, 2, 3, 0, 3), A0223 = det2(m, 2, 3, 0, 2), A0123 = det2(m, 2, 3, 0, 1);
    const float A2313 = det2(m, 1, 3, 2, 3), A1313 = det2(m, 1, 3, 1, 3), A1213 = det2(m, 1, 3, 1, 2);
    const float A2312 = det2(m, 1, 2, 2, 3), A1312 = det2(m, 1, 2, 1, 3), A1212 = det2(m, 1, 2, 1, 2);
    const float A0313 = det2(m, 1, 3, 0, 3), A0213 = det2(m, 1, 3, 0, 2), A0312 = det2(m, 1, 2, 0, 3);
    const float A0212 = det2(m, 1, 2, 0, 2), A0113 = det2(m, 1, 3, 0, 1), A0112 = det2(m, 1, 2, 0, 1);

    float det = M(m, 0, 0) * tri(M(m, 1, 1), A2323, M(m, 1, 2), A1323, M(m, 1, 3), A1223)
              - M(m, 0, 1) * tri(M(m, 1, 0), A2323, M(m, 1, 2), A0323, M(m, 1, 3), A0223)
              + M(m, 0, 2) * tri(M(m, 1, 0), A1323, M(m, 1, 1), A0323, M(m, 1, 3), A0123)
              - M(m, 0, 3) * tri(M(m, 1, 0), A1223, M(m, 1, 1), A0223, M(m, 1, 2), A0123);
    det = 1 / det;

    M(out, 0, 0) = det *  tri(M(m, 1, 1), A2323, M(m, 1, 2), A1323, M(m, 1, 3), A1223);
    M(out, 0, 1) = det * -tri(M(m, 0, 1), A2323, M(m, 0, 2), A1323, M(m, 0, 3), A1223);
    M(out, 0, 2) = det *  tri(M(m, 0, 1), A2313, M(m, 0, 2), A1313, M(m, 0, 3), A1213);
    M(out, 0, 3) = det * -tri(M(m, 0, 1), A2312, M(m, 0, 2), A1312, M(m, 0, 3), A1212);
    M(out, 1, 0) = det * -tri(M(m, 1, 0), A2323, M(m, 1, 2), A0323, M(m, 1, 3), A0223);
    M(out, 1, 1) = det *  tri(M(m, 0, 0), A2323, M(m, 0, 2), A0323, M(m, 0, 3), A0223);
    M(out, 1, 2) = det * -tri(M(m, 0, 0), A2313, M(m, 0, 2), A0313, M(m, 0, 3), A0213);
    M(out, 1, 3) = det *  tri(M(m, 0, 0), A2312, M(m, 0, 2), A0312, M(m, 0, 3), A0212);
    M(out, 2, 0) = det *  tri(M(m, 1, 0), A1323, M(m, 1, 1), A0323, M(m, 1, 3), A0123);
    M(out, 2, 1) = det * -tri(M(m, 0, 0), A1323, M(m, 0, 1), A0323, M(m, 0, 3), A0123);
    M(out, 2, 2) = det *  tri(M(m, 0, 0), A1313, M(m, 0, 1), A0313, M(m, 0, 3), A0113);
    M(out, 2, 3) = det * -tri(M(m, 0, 0), A1312, M(m, 0, 1), A0312, M(m, 0, 3), A0112);
    M(out, 3, 0) = det * -tri(M(m, 1, 0), A1223, M(m, 1, 1), A0223, M(m, 1, 2), A0123);
    M(out, 3, 1) = det *  tri(M(m, 0, 0), A1223, M(m, 0, 1), A0223, M(m, 0, 2), A0123);
    M(out, 3, 2) = det * -tri(M(m, 0, 0), A1213, M(m, 0, 1), A0213, M(m, 0, 2), A0113);
    M(out, 3, 3) = det *  tri(M(m, 0, 0), A1212, M(m, 0, 1), A0212, M(m, 0, 2), A0112);
}

static void mat4_identity(float m[16])
{
    memset(m, 0, 16 * sizeof(float));
    M(m, 0, 0) = M(m, 1, 1) = M(m, 2, 2) = M(m, 3, 3) = 1.f;
}

/* fx::apply(mat4, vec4) [flux-inferred]: glm M*v, m[col][row]; left-to-right fused chain. */
static void mat4_apply(const float m[16], const float v[4], float out[4])
{
    for (int r = 0; r < 4; r++) {
        out[r] = FMA(M(m, 3, r), v[3], FMA(M(m, 2, r), v[2], FMA(M(m, 1, r), v[1], M(m, 0, r) * v[0])));
    }
}

/* Camera::recompute_projection (Camera.cpp:169-174) with fx::perspective == glm
 * perspectiveFov (RH, z in [-1,1]) [flux-inferred]; Camera::recompute_view
 * (Camera.cpp:176-183) with fx::lookat == glm lookAt RH [flux-inferred]. */
void lo_camera_matrices(float fov_deg, float nearclip, float farclip, uint32_t width, uint32_t height,
                        const float pos[3], const float dir[3],
                        float proj[16], float proj_inv[16], float view[16], float view_inv[16])
{
    const float rad = fov_deg * 0.01745329251994329577f;             /* fx::radians */
    const float w_f = (float)width, h_f = (float)height;
    const float h = cosf(0.5f * rad) / sinf(0.5f * rad);
    const float w = h * h_f / w_f;

    memset(proj, 0, 16 * sizeof(float));
    M(proj, 0, 0) = w;
    M(proj, 1, 1) = h;
    M(proj, 2, 2) = -(farclip + nearclip) / (farclip - nearclip);
    M(proj, 2, 3) = -1.f;
    M(proj, 3, 2) = -(2.f * farclip * nearclip) / (farclip - nearclip);
    lo_mat4_inverse(proj, proj_inv);                                 /* :173 */

    v3 eye = v3_make(pos[0], pos[1], pos[2]);
    v3 at = v3_add(eye, v3_make(dir[0], dir[1], dir[2]));            /* :178 */
    v3 up = v3_make(0.f, 1.f, 0.f);
    v3 f = v3_normalize(v3_sub(at, eye));
    v3 s = v3_normalize(v3_cross(f, up));
    v3 u = v3_cross(s, f);
    mat4_identity(view);
    M(view, 0, 0) = s.x; M(view, 1, 0) = s.y; M(view, 2, 0) = s.z;
    M(view, 0, 1) = u.x; M(view, 1, 1) = u.y; M(view, 2, 1) = u.z;
    M(view, 0, 2) = -f.x; M(view, 1, 2) = -f.y; M(view, 2, 2) = -f.z;
    M(view, 3, 0) = -v3_dot(s, eye);
    M(view, 3, 1) = -v3_dot(u, eye);
    M(view, 3, 2) = v3_dot(f, eye);
    lo_mat4_inverse(view, view_inv);                                 /* :182 */
}

/* Camera::Camera, Camera.cpp:77-90: dir (0,0,-1), pos (0,0,0); depth / show_depth
 * defaults camera.h:33-35. */
void lo_camera_init(lo_camera* cam, float fov_deg, float nearclip, float farclip,
                    uint32_t width, uint32_t height)
{
    const float pos[3] = { 0.f, 0.f, 0.f }, dir[3] = { 0.f, 0.f, -1.f };
    float proj[16], view[16];
    memcpy(cam->pos, pos, sizeof pos);
    lo_camera_matrices(fov_deg, nearclip, farclip, width, height, pos, dir,
                       proj, cam->proj_inv, view, cam->view_inv);
    cam->focus_depth = 10.f;
    cam->show_depth = 1u;
}

/* Camera::update's pose arithmetic, Camera.cpp:135-152, followed by the moved branch
 * :160-164 (recompute_view). */
void lo_camera_pose(lo_camera* cam, float fov_deg, float nearclip, float farclip,
                    uint32_t width, uint32_t height, const float pos[3], float pitch, float yaw)
{
    const float quarter_pi = 3.14159265358979323846f / 4;
    pitch = fminf(fmaxf(pitch, -quarter_pi), quarter_pi);            /* :135 */
    const float cp = cosf(pitch), sp = sinf(pitch), cy = cosf(yaw), sy = sinf(yaw);
    const float base[3] = { 0.f, 0.f, -1.f };
    float dir[3];
    dir[0] =  (base[0] * cy + base[2] * sy);
    dir[1] = -(base[0] * sp * sy + base[1] * cp - base[2] * sp * cy);
    dir[2] =  (base[0] * cp * sy + base[1] * sp + base[2] * cp * cy);
    float proj[16], view[16];
    memcpy(cam->pos, pos, 3 * sizeof(float));
    lo_camera_matrices(fov_deg, nearclip, farclip, width, height, pos, dir,
                       proj, cam->proj_inv, view, cam->view_inv);
    cam->focus_depth = 10.f;
    cam->show_depth = 1u;
}

/* One entry of Camera::recompute_rays, Camera.cpp:193-213 */
static v3 camera_ray(const lo_camera* cam, uint32_t x, uint32_t y, uint32_t width, uint32_t height)
{
    float cx = (float)x / (float)width;                              /* :195 */
    float cy = (float)y / (float)height;                             /* :196 */
    cx = cx * 2.f - 1.f;                                             /* :199-200 */
    cy = cy * 2.f - 1.f;
    const float extended[4] = { cx, cy, 1.f, 1.f };                  /* :202 */
    float target[4];
    mat4_apply(cam->proj_inv, extended, target);                     /* :203 */
    const float scalar = 1 / target[3];                              /* :206 */
    v3 corrected = v3_scale(v3_make(target[0], target[1], target[2]), scalar); /* :208 */
    v3 n = v3_normalize(corrected);                                  /* :209 */
    const float padded[4] = { n.x, n.y, n.z, 0.f };                  /* :210 */
    float ray[4];
    mat4_apply(cam->view_inv, padded, ray);                          /* :212 */
    return v3_make(ray[0], ray[1], ray[2]);                          /* :213 */
}

/* Camera::recompute_rays, Camera.cpp:185-216 */
void lo_camera_rays(const lo_camera* cam, uint32_t width, uint32_t height, float* rays)
{
    for (uint32_t y = 0; y < height; y++) {
        for (uint32_t x = 0; x < width; x++) {
            v3 r = camera_ray(cam, x, y, width, height);
            float* o = rays + 3 * ((size_t)y * width + x);
            o[0] = r.x; o[1] = r.y; o[2] = r.z;
        }
    }
}

/* ---------------------------------------------------------------------------------
 * Scenes
 * ------------------------------------------------------------------------------- */
static lo_sphere make_sphere(float x, float y, float z, float radius, float dr, float dg, float db,
                             float albedo, float metallic, float roughness)
{
    lo_sphere s = { { x, y, z }, radius, { dr, dg, db }, albedo, metallic, roughness };
    return s;
}

/* renderer.h:62-70: spheres{ s, a, q, r, t } in that order */
void lo_default_scene(lo_sphere out[5])
{
    out[0] = make_sphere(0.f, .5f, -10.f, 1.0f, 0.f, 0.f, 1.f, 1.f, .001f, .4f);      /* s :62 */
    out[1] = make_sphere(0.f, 102.f, 0.f, 100.f, .6f, .6f, .6f, 1.f, .1f, .5f);       /* a :68 */
    out[2] = make_sphere(3.f, .5f, -10.f, 1.0f, 0.f, 1.f, 0.f, 1.f, .001f, .4f);      /* q :63 */
    out[3] = make_sphere(6.f, .5f, -10.f, 1.0f, 1.f, 0.f, 0.f, 1.f, .001f, .4f);      /* r :64 */
    out[4] = make_sphere(3.f, .5f, -23.f, 10.0f, 1.f, 1.f, 1.f, 1.f, .4f, 0.f);       /* t :66 */
}

/* SURVEY 8d config #5 (build-defined; the reference has no large scene): sphere 0 is the
 * ground `a`; sphere i >= 1 takes three pcg3d draws keyed by (i, draw). */
void lo_synthetic_scene(lo_sphere* out, uint32_t n, uint64_t seed)
{
    static const float metallic_cycle[3] = { .001f, .1f, .4f };
    uint32_t keys[3];
    lo_seed_keys(seed, keys);
    if (n == 0) { return; }
    out[0] = make_sphere(0.f, 102.f, 0.f, 100.f, .6f, .6f, .6f, 1.f, .1f, .5f);
    for (uint32_t i = 1; i < n; i++) {
        uint32_t h0[3], h1[3], h2[3];
        lo_pcg3d(i ^ keys[0], 0u ^ keys[1], keys[2], h0);
        lo_pcg3d(i ^ keys[0], 1u ^ keys[1], keys[2], h1);
        lo_pcg3d(i ^ keys[0], 2u ^ keys[1], keys[2], h2);
        lo_sphere s;
        s.pos[0] = FMA(u01(h0[0]), 80.f, -40.f);
        s.pos[1] = FMA(u01(h0[1]), 21.5f, -20.f);
        s.pos[2] = FMA(u01(h0[2]), 80.f, -90.f);
        s.radius = FMA(u01(h1[0]), .6f, .2f);
        s.roughness = u01(h1[1]) * .5f;
        s.diffuse[0] = FMA(u01(h2[0]), .9f, .1f);
        s.diffuse[1] = FMA(u01(h2[1]), .9f, .1f);
        s.diffuse[2] = FMA(u01(h2[2]), .9f, .1f);
        s.albedo = 1.f;
        s.metallic = metallic_cycle[i % 3u];
        out[i] = s;
    }
}

/* ---------------------------------------------------------------------------------
 * Renderer::render_to, Renderer.cpp:366-471 (pge calls :368, :437-445 are display only
 * and dropped; camera.update/moved :375-390 is the caller's reset, see luma_oracle.h).
 * ------------------------------------------------------------------------------- */
void lo_render(const lo_sphere* spheres, uint32_t n_spheres, const lo_camera* cam,
               const lo_params* params, uint32_t frame_index,
               uint32_t row_begin, uint32_t row_end,
               float* accum, uint32_t* target, lo_stats* stats)
{
    frame_t f;
    f.scene.spheres = spheres;
    f.scene.n = n_spheres;
    f.scene.stats = stats;
    f.cam = cam;
    f.params = params;
    f.slots_per_bounce = 2u + params->paths;
    lo_seed_keys(params->seed, f.keys);

    const uint32_t width = params->width, height = params->height;
    if (row_end > height) { row_end = height; }
    const v3 campos = v3_make(cam->pos[0], cam->pos[1], cam->pos[2]);
    const int show_depth = cam->show_depth && !(params->flags & LO_FLAG_NO_SHOW_DEPTH);
    if (row_begin >= row_end) { return; }

    /* camera.rays for the band: the reference builds the table once (Camera.cpp:89) and
     * reads it at :413 and :293. */
    v3* rays = (v3*)malloc(sizeof(v3) * (size_t)(row_end - row_begin) * width);
    if (!rays) { return; }
    for (uint32_t y = row_begin; y < row_end; y++) {
        for (uint32_t x = 0; x < width; x++) {
            rays[(size_t)(y - row_begin) * width + x] = camera_ray(cam, x, y, width, height);
        }
    }

    for (uint32_t frame = 0; frame < params->spp; frame++) {
        const uint32_t sample = frame_index + frame;
        const float frame_count = (float)sample + 1.f;               /* renderer.h:53, :470 */

        for (uint32_t y = row_begin; y < row_end; y++) {             /* :393 */
            for (uint32_t x = 0; x < width; x++) {                   /* :395 */
                const uint32_t index = y * width + x;                /* :399 */
                v3 dir = rays[(size_t)(y - row_begin) * width + x];  /* :413 (camera.rays) */

                isect_t is = trace_ray(&f.scene, campos, dir);       /* :418 */
                float depth = FLT_MAX;                               /* :419-423 */
                if (is.object >= 0) { depth = is.distance; }

                float blur;
                int in_focus;
                focus_term(depth, cam->focus_depth, &blur, &in_focus); /* :425-432 */

                v3 result = render_pixel(&f, dir, index, sample, blur); /* :434-435 */

                if (show_depth && in_focus) {                        /* :448-454 */
                    result = v3_scale(result, 1.2f);
                }

                float* data = accum + 3 * (size_t)index;             /* :457-458 */
                data[0] += result.x; data[1] += result.y; data[2] += result.z;

                const float inv = 1 / frame_count;                   /* :460 */
                const float mean[3] = { data[0] * inv, data[1] * inv, data[2] * inv };
                target[index] = lo_pack_rgb(mean, params->flags);    /* :461 */
            }
        }
    }
    free(rays);
}

/* The DOF pre-pass alone (Renderer.cpp:413-432) for rows [row_begin,row_end): per pixel the blur
 * handed to render_pixel and the `focus < .5f` decision.  For tests/golden/make_contract_sensitivity.py,
 * which counts the pixels whose decision differs between the two builds of this file. */
void lo_dof_prepass(const lo_sphere* spheres, uint32_t n_spheres, const lo_camera* cam,
                    uint32_t width, uint32_t height, uint32_t row_begin, uint32_t row_end,
                    float* blur, uint8_t* in_focus)
{
    scene_t sc = { spheres, n_spheres, NULL };
    const v3 campos = v3_make(cam->pos[0], cam->pos[1], cam->pos[2]);
    if (row_end > height) { row_end = height; }
    for (uint32_t y = row_begin; y < row_end; y++) {
        for (uint32_t x = 0; x < width; x++) {
            const size_t index = (size_t)y * width + x;
            isect_t is = trace_ray(&sc, campos, camera_ray(cam, x, y, width, height));   /* :418 */
            float depth = FLT_MAX;
            if (is.object >= 0) { depth = is.distance; }
            int flag;
            focus_term(depth, cam->focus_depth, &blur[index], &flag);
            in_focus[index] = (uint8_t)flag;
        }
    }
}

/* ---------------------------------------------------------------------------------
 * image.cpp
 * ------------------------------------------------------------------------------- */

/* Image::export_to_ppm, image.cpp:30-64: P3, x outer / y inner, "r g b  " triples, one
 * text line per x.  Does not free the caller's buffer (image.cpp:56 does). */
int lo_export_ppm(const uint32_t* data, int32_t width, int32_t height, const char* path)
{
    FILE* fp = fopen(path, "wb");
    if (!fp) { return -1; }
    fprintf(fp, "P3\n%d %d\n255\n", width, height);                  /* :33 */
    for (int32_t x = 0; x < width; x++) {                            /* :35 */
        for (int32_t y = 0; y < height; y++) {                       /* :37 */
            const uint32_t d = data[(size_t)y * width + x];          /* :39-40 */
            fprintf(fp, "%u %u %u  ", (d & 0xFF0000u) >> 16, (d & 0xFF00u) >> 8, d & 0xFFu); /* :42-46 */
        }
        fputc('\n', fp);                                             /* :49 */
    }
    return fclose(fp) == 0 ? 0 : -1;
}

/* Image::export_to_bmp, image.cpp:66-153: 24-bit BI_RGB, bottom-up, rows padded to 4 B. */
int lo_export_bmp(const uint32_t* data, int32_t width, int32_t height, const char* path)
{
    const int row_stride = width * 3;
    const int padding = (4 - (row_stride % 4)) % 4;                  /* :74 */
    const int pixel_bytes = (row_stride + padding) * height;
    const int file_size = 14 + 40 + pixel_bytes;                     /* :77 */
    unsigned char fh[14] = { 'B', 'M', 0, 0, 0, 0, 0, 0, 0, 0, 54, 0, 0, 0 };
    unsigned char ih[40] = { 40, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 24, 0 };
    for (int i = 0; i < 4; i++) {
        fh[2 + i] = (unsigned char)(file_size >> (8 * i));           /* :104-107 */
        ih[4 + i] = (unsigned char)(width >> (8 * i));               /* :110-113 */
        ih[8 + i] = (unsigned char)(height >> (8 * i));              /* :116-119 */
    }
    FILE* fp = fopen(path, "wb");
    if (!fp) { return -1; }
    fwrite(fh, 1, sizeof fh, fp);
    fwrite(ih, 1, sizeof ih, fp);
    const unsigned char pad[3] = { 0, 0, 0 };
    for (int y = height - 1; y >= 0; --y) {                          /* :134 */
        for (int x = 0; x < width; ++x) {
            const uint32_t d = data[(size_t)y * width + x];
            const unsigned char bgr[3] = { (unsigned char)(d & 0xFF), (unsigned char)((d >> 8) & 0xFF),
                                           (unsigned char)((d >> 16) & 0xFF) };
            fwrite(bgr, 1, 3, fp);                                   /* :143-145 */
        }
        fwrite(pad, 1, (size_t)padding, fp);                         /* :148 */
    }
    return fclose(fp) == 0 ? 0 : -1;
}
