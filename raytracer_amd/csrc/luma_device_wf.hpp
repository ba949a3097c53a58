// luma_device_wf.hpp -- the wavefront form of the ray-per-lane render kernel (device code; needs luma_device.hpp).
// Included by luma_inst_wf.hip, which instantiates it, and by luma_kernels.hip, which only names the instances.
#pragma once
#include "luma_device.hpp"

namespace luma {

// ---------------------------------------------------------------------------------
// The wavefront form of the ray-per-lane kernel (round 3): ONE wave owns P pixels (P > 64) and runs
// the steps of their paths in two separate, full-width phases instead of one divergent loop body.
//
// Why.  In render_kernel a lane's (sample, bounce) step hits a sphere with probability ~0.38, decided by the
// reference's +-blur direction noise (Renderer.cpp:302), so the ~250-instruction hit side (hit point, normal,
// indirect ray and its scan, shading, reflection) runs on nearly every trip of a wave for 38 % of its lanes and
// the sky / tonemap side for the other 62 %: 44 % active lanes per VALU instruction at the VALU issue limit
// (profiles/r02/config3_pmc_summary.txt).  Pooling the hits of a whole workgroup cost more in barriers than it
// saved (profiles/r02/ab_compacted_hits_rejected.txt).  A wave needs no barrier to talk to itself.
//
// How.  The state of every path (direction, radiance so far, frame and bounce numbers, the pixel's accumulator)
// lives in the wave's own LDS, 64 B per pixel; the wave keeps two FIFO queues of pixel slots there:
//   ready: paths whose next step is "draw the noise, scan the spheres from camera.pos" (phase A, :302-319;
//          a path that ended is first tonemapped, added to its pixel and restarted as the next frame, :362, :457)
//   hit:   paths that found a sphere and need the hit side (phase C, :321-359)
// Each trip of the wave runs ONE phase on the 64 oldest entries of one queue -- every lane busy -- and pushes the
// slots onto the queue of their next phase (ballot + mbcnt, no atomics).  With P = 128 paths in flight one of the
// two queues always holds 64 entries until the wave's last pixels run out of frames.  A pixel has one path in flight
// at a time, so its frames are still added in frame order (:457-458) and every path's arithmetic is the same
// instruction sequence as in render_kernel: bit-identical frames.
// ---------------------------------------------------------------------------------
// LDS per pixel slot: U = (direction of the next step | after phase A the noised direction, t), V = (radiance of the
// path so far, state word), W = (the pixel's accumulator, blur): 48 B.  The pixel's camera ray, read once per FRAME,
// stays out of LDS: it comes from a table the wave fills at its start (RenderArgs::split_scratch: float4 per band
// pixel).  (The accumulator read-modify-written in the accumulation planes themselves instead -- 36 B, 30 waves per
// CU -- cost 0.67 of 2.01 ms in scattered L2 traffic: seven vector memory instructions per phase instead of one.)
template <int P, bool NOISE, int NS>
__global__ __launch_bounds__(64) void render_kernel_wf(const RenderArgs A)
{
    static_assert(P == 128, "a wave's tile is 32 x 4 pixels; slots are 7 bits of a queue entry");
    static_assert(NS >= 1 && NS <= 8, "the sphere index is 3 bits of a queue entry");
    extern __shared__ float4 lds_rec[];
    float4* const U = lds_rec;
    float4* const V = lds_rec + P;
    float4* const l_mat0 = lds_rec + 2 * P;     // [8]
    float4* const l_gen = l_mat0 + 8;           // [8] general records (centre, radius), for the hit point's normal
    float2* const l_mat1 = reinterpret_cast<float2*>(l_gen + 8);           // [8]
    float4* const W = reinterpret_cast<float4*>(l_mat1 + 8);               // [P]
    constexpr uint32_t QCAP = 128u;                                         // ring size: a power of two >= P
    uint16_t* const readyQ = reinterpret_cast<uint16_t*>(W + P);           // [QCAP] ring
    uint16_t* const hitQ = readyQ + QCAP;                                   // [QCAP] ring
    constexpr uint32_t kEnded = 1u << 29, kGain = 1u << 30;                // state word: frame [0,16), bounce [16,29)

    const uint32_t lane = threadIdx.x;
    if (lane < (uint32_t)NS) { l_mat0[lane] = A.mat0[lane]; l_mat1[lane] = A.mat1[lane]; l_gen[lane] = A.records[NS + lane]; }
    const float4* rec_gen = l_gen;
    const f3 campos = mk3(A.cam_pos[0], A.cam_pos[1], A.cam_pos[2]);
    const uint32_t slots = 2u + A.paths;
    const bool pathtrace = A.mode == MODE_PATHTRACE;
    const uint32_t spp_all = A.bounces ? A.spp : 0u;
    uint32_t nrays = 0;
    uint32_t rhead = 0, rtail = 0, hhead = 0, htail = 0;        // free-running; wave-uniform

    auto closest_cam = [&](f3 o, f3 d, float& t) { return scan_spheres<true, false, NS>((LUMA_SCALAR_PTR)A.records, (uint32_t)NS, o, d, t); };
    auto closest_gen = [&](f3 o, f3 d, float& t) { return scan_spheres<false, false, NS>((LUMA_SCALAR_PTR)(A.records + NS), (uint32_t)NS, o, d, t); };
    // entry -> the queue's tail, for the lanes with `pred`, in lane order (every lane of the wave calls this)
    auto push = [&](uint16_t* q, uint32_t& tail, bool pred, uint32_t entry) {
        const unsigned long long m = __builtin_amdgcn_ballot_w64(pred);
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (pred) { q[(tail + rank) & (QCAP - 1u)] = (uint16_t)entry; }
        tail += (uint32_t)__builtin_popcountll(m);
    };

    // the wave's tile: 32 pixels wide, 4 rows; slot = row in the tile * 32 + column
    const uint32_t tiles_x = (A.width + 31u) / 32u;
    const uint32_t ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const uint32_t x0 = tx * 32u, yl0 = ty * 4u;
    // global index of the first pixel of each of the tile's rows (RNG, :399), wave-uniform
    const uint32_t gp0 = global_row(A, min(yl0, A.rows - 1u)) * A.width + x0, gp1 = global_row(A, min(yl0 + 1u, A.rows - 1u)) * A.width + x0;
    const uint32_t gp2 = global_row(A, min(yl0 + 2u, A.rows - 1u)) * A.width + x0, gp3 = global_row(A, min(yl0 + 3u, A.rows - 1u)) * A.width + x0;
    const uint32_t local0 = yl0 * A.width + x0;
    auto pixel_of = [&](uint32_t slot) {
        const uint32_t r = slot >> 5;
        return (r == 0u ? gp0 : r == 1u ? gp1 : r == 2u ? gp2 : gp3) + (slot & 31u);
    };
    auto local_of = [&](uint32_t slot) { return local0 + (slot >> 5) * A.width + (slot & 31u); };
    float4* const raytab = reinterpret_cast<float4*>(A.split_scratch);
    const auto KA = kernarg_args();
    __builtin_amdgcn_wave_barrier();
#pragma unroll 1
    for (uint32_t j = 0; j < (uint32_t)(P / 64); j++) {
        const uint32_t slot = j * 64u + lane;
        const uint32_t x = x0 + (slot & 31u), yl = yl0 + (slot >> 5);
        const bool live = (x < A.width) && (yl < A.rows);
        if (live) {
            const uint32_t y = global_row(A, yl);
            const f3 dir0 = camera_ray(KA, x, y);                // camera.rays[index], :413
            float t0;                                            // DOF pre-pass, :416-432
            const int h0 = closest_cam(campos, dir0, t0);
            nrays++;
            const float depth = (h0 >= 0) ? t0 : FLT_MAX;
            const float dd = depth - A.focus_depth;
            const float x2 = dd * dd, x4 = x2 * x2, x8 = x4 * x4, x10 = x8 * x2;
            const float focus = 1.0f - luma_expf(-x10);
            const uint32_t sb = (A.show_depth && (focus < .5f)) ? kGain : 0u;
            const uint32_t local = local_of(slot);
            raytab[local] = make_float4(dir0.x, dir0.y, dir0.z, 0.f);
            U[slot] = make_float4(dir0.x, dir0.y, dir0.z, 0.f);
            V[slot] = make_float4(0.f, 0.f, 0.f, __builtin_bit_cast(float, sb));
            W[slot] = make_float4(KA->acc[0][local], KA->acc[1][local], KA->acc[2][local], fma_(focus, 4.5f, 0.5f));   // :429-432
        }
        push(readyQ, rtail, live && spp_all > 0u, slot);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");       // the table entries are written before any lane reads one

#pragma unroll 1
    for (;;) {
        __builtin_amdgcn_wave_barrier();
        const uint32_t rc = rtail - rhead, hc = htail - hhead;
        bool hit_phase;
        uint32_t n;
        if (hc >= 64u)            { hit_phase = true;  n = 64u; }
        else if (rc >= 64u)       { hit_phase = false; n = 64u; }
        else if ((rc | hc) == 0u) { break; }
        else if (rc >= hc)        { hit_phase = false; n = rc; }   // the wave's last pixels: whichever queue holds more
        else                      { hit_phase = true;  n = hc; }

        if (!hit_phase) {
            // ---- phase A: [finish the path that ended, start the pixel's next frame], noise, scan from camera.pos ----
            uint32_t slot = 0;
            int idx = -1;
            bool alive = false;
            if (lane < n) {
                slot = readyQ[(rhead + lane) & (QCAP - 1u)];
                const float4 u = U[slot], v = V[slot];
                float4 w = W[slot];
                const float blur = w.w;
                uint32_t sb = __builtin_bit_cast(uint32_t, v.w);
                f3 direct = mk3(v.x, v.y, v.z);
                f3 dir = mk3(u.x, u.y, u.z);
                alive = true;
                if (sb & kEnded) {
                    const float4 rt = raytab[local_of(slot)];
                    const float gainf = (sb & kGain) ? 1.2f : 1.0f;                 // :448-452
                    const f3 result = scale3(tonemap3(direct), gainf);             // :362, :57-70
                    w.x += result.x; w.y += result.y; w.z += result.z;             // :457-458
                    W[slot] = w;
                    sb = (sb & kGain) | ((sb & 0xFFFFu) + 1u);                      // next frame, bounce 0
                    direct = mk3(0.f, 0.f, 0.f);
                    dir = mk3(rt.x, rt.y, rt.z);                                   // :293
                    alive = (sb & 0xFFFFu) < spp_all;
                }
                if (alive) {
                    const uint32_t sample = A.frame_index + (sb & 0xFFFFu);
                    const uint32_t slot0 = ((sb >> 16) & 0x1FFFu) * slots;
                    f3 dn;
                    if (NOISE) { dn = add3(dir, rng_vec3(A, pixel_of(slot), sample, slot0, -blur, blur)); }   // :118-119, :302
                    else       { dn = add3(dir, mk3(0.f, 0.f, 0.f)); }
                    float t;
                    idx = closest_cam(campos, dn, t);                              // :304
                    nrays++;
                    if (idx < 0) {                                                 // :306-319
                        const float clamped = __builtin_fminf(__builtin_fmaxf(dn.y, -1.f), 1.f);
                        const float adjusted = (clamped + 1.f) * .5f;
                        direct = add3(direct, color_lerp(mk3(.529f, .808f, .922f), mk3(.106f, .275f, .711f), adjusted));
                        sb |= kEnded;
                    }
                    U[slot] = make_float4(dn.x, dn.y, dn.z, t);
                    V[slot] = make_float4(direct.x, direct.y, direct.z, __builtin_bit_cast(float, sb));
                }
            }
            rhead += n;
            __builtin_amdgcn_wave_barrier();
            push(hitQ, htail, alive && idx >= 0, slot | ((uint32_t)idx << 8));
            push(readyQ, rtail, alive && idx < 0, slot);
        } else {
            // ---- phase C: the hit side, :321-359 ----
            uint32_t slot = 0;
            if (lane < n) {
                const uint32_t e = hitQ[(hhead + lane) & (QCAP - 1u)];
                slot = e & 0xFFu;
                const int idx = (int)(e >> 8);
                const float4 u = U[slot], v = V[slot];
                uint32_t sb = __builtin_bit_cast(uint32_t, v.w);
                const uint32_t pixel = pixel_of(slot);
                const uint32_t sample = A.frame_index + (sb & 0xFFFFu);
                uint32_t bounce = (sb >> 16) & 0x1FFFu;
                const uint32_t slot0 = bounce * slots;
                const f3 dn = mk3(u.x, u.y, u.z);
                f3 direct = mk3(v.x, v.y, v.z);

                const Hit hit = finish_hit(rec_gen, idx, campos, dn, u.w);
                const float4 m0 = l_mat0[idx];                   // diffuse rgb, albedo
                const float2 m1 = l_mat1[idx];                   // metallic, roughness
                const f3 diffuse = mk3(m0.x, m0.y, m0.z);
                const float cos_theta = dot3(neg3(dn), hit.normal);   // :331-332
                if (cos_theta >= 0.0f) {                         // :334
                    f3 indirect = mk3(0.f, 0.f, 0.f);
                    if (m0.w > 0.0f) {                           // :337
                        if (pathtrace) {                         // :187
                            for (uint32_t p = 0; p < A.paths; p++) {   // :192
                                f3 d = normalize3(rng_vec3(A, pixel, sample, slot0 + 2u + p, -1.0f, 1.0f));  // :194
                                if (dot3(d, hit.normal) < 0.0f) { d = neg3(d); }   // :197-200
                                const f3 nd = add3(hit.normal, d);     // :202
                                float tc;
                                const int ci = closest_gen(hit.pos, nd, tc);   // :204
                                nrays++;
                                if (ci >= 0) {                   // :206-209
                                    const float4 cm = l_mat0[ci];
                                    indirect = add3(indirect, mk3(cm.x, cm.y, cm.z));
                                }
                            }
                        }
                        indirect = scale3(indirect, A.inv_samples);    // :212-213
                        indirect = scale3(indirect, m0.w);       // :340
                    }
                    const float value = (cos_theta + 1.f) * .5f; // :343
                    const float scalar = value * (1.f - m1.x);   // :345
                    const f3 capture = color_lerp(diffuse, indirect, 1.f - scalar);   // :346
                    direct = fma3(capture, scalar * .8f, direct);    // :347-349
                }
                bounce++;
                const bool path_ends = (m1.x == 0.0f) || (bounce == A.bounces);   // :353-357, :300
                if (!path_ends) {
                    f3 nrm = hit.normal;
                    if (NOISE) {
                        const float gain = .2f * m1.y;           // :225
                        nrm = add3(nrm, rng_vec3(A, pixel, sample, slot0 + 1u, -gain, gain));   // :226-227
                    } else {
                        nrm = add3(nrm, mk3(0.f, 0.f, 0.f));
                    }
                    const float k2 = 2.0f * dot3(dn, nrm);       // fx::reflect, :229
                    const f3 dir = fma3(nrm, -k2, dn);           // :359
                    U[slot] = make_float4(dir.x, dir.y, dir.z, 0.f);
                    sb += 1u << 16;
                } else {
                    sb |= kEnded;
                }
                V[slot] = make_float4(direct.x, direct.y, direct.z, __builtin_bit_cast(float, sb));
            }
            hhead += n;
            __builtin_amdgcn_wave_barrier();
            push(readyQ, rtail, lane < n, slot);
        }
    }

    // one write of the accumulation planes and the packed pixel per launch, as in render_kernel
    __builtin_amdgcn_wave_barrier();
    const auto E = kernarg_args();
#pragma unroll 1
    for (uint32_t j = 0; j < (uint32_t)(P / 64); j++) {
        const uint32_t slot = j * 64u + lane;
        const uint32_t x = x0 + (slot & 31u), yl = yl0 + (slot >> 5);
        if ((x < E->width) && (yl < E->rows)) {
            const uint32_t local = yl * E->width + x;
            const float4 w = W[slot];
            const float ar = w.x, ag = w.y, ab = w.z;
            E->acc[0][local] = ar; E->acc[1][local] = ag; E->acc[2][local] = ab;
            const float inv = 1.0f / E->frame_count_last;            // :460
            int r = (int)((ar * inv) * 255.0f);                      // ::RGB, :44-55
            int g = (int)((ag * inv) * 255.0f);
            int b = (int)((ab * inv) * 255.0f);
            const uint32_t flags = E->flags;
            if (flags & FLAG_CLAMP_PACK) {
                r = min(max(r, 0), 255); g = min(max(g, 0), 255); b = min(max(b, 0), 255);
            }
            const uint32_t px = pack_pixel(r, g, b, flags);
            E->target[local] = px;                                   // :461
            uint32_t* const mirror = E->target_mirror;
            if (mirror) { mirror[E->mirror_is_frame ? frame_pixel(E, local) : local] = px; }
        }
    }
    wave_add_u64(E->counters, nrays);
}

}  // namespace luma
