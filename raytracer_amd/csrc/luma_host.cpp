// luma_host.cpp -- host-side (CPU, once per camera move / scene load) parts of the C-ABI:
// camera matrices, the built-in scenes and the RNG seed schedule.  No GPU work here.
// Built with -ffp-contract=off: the matrices feed the kernels' arithmetic contract.
#include "luma_host.h"

#include <cmath>
#include <cstring>

namespace luma {
namespace {

struct Vec3 {
    float x, y, z;
};

inline Vec3 operator+(Vec3 a, Vec3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline Vec3 operator*(Vec3 a, float s) { return { a.x * s, a.y * s, a.z * s }; }

inline float dot(Vec3 a, Vec3 b) { return std::fmaf(a.z, b.z, std::fmaf(a.y, b.y, a.x * b.x)); }
inline Vec3 normalize(Vec3 a) { return a * (1.0f / std::sqrt(dot(a, a))); }
inline Vec3 cross(Vec3 a, Vec3 b)   // Camera.cpp:64-72
{
    return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x };
}

// fx::mat4 as the reference indexes it: e[i][j], flattened [i*4+j] at the ABI.
struct Mat4 {
    float e[4][4];
    static Mat4 zero() { Mat4 m; std::memset(&m, 0, sizeof m); return m; }
    static Mat4 identity()
    {
        Mat4 m = zero();
        for (int i = 0; i < 4; i++) { m.e[i][i] = 1.f; }
        return m;
    }
};

// 2x2 minor over rows (r0,r1) and columns (c0,c1).
inline float minor2(const Mat4& m, int r0, int r1, int c0, int c1)
{
    return m.e[r0][c0] * m.e[r1][c1] - m.e[r0][c1] * m.e[r1][c0];
}

// General 4x4 inverse by cofactors of 3x3 blocks built from 2x2 minors: the formula of
// ::inverse, Camera.cpp:14-62, evaluated in the same association order so the result is
// bit-identical to the reference's arithmetic under the contract.
Mat4 inverse(const Mat4& m)
{
    // minors of the lower row pairs, named by (rows | cols)
    const float r23c23 = minor2(m, 2, 3, 2, 3), r23c13 = minor2(m, 2, 3, 1, 3), r23c12 = minor2(m, 2, 3, 1, 2);
    const float r23c03 = minor2(m, 2, 3, 0, 3), r23c02 = minor2(m, 2, 3, 0, 2), r23c01 = minor2(m, 2, 3, 0, 1);
    const float r13c23 = minor2(m, 1, 3, 2, 3), r13c13 = minor2(m, 1, 3, 1, 3), r13c12 = minor2(m, 1, 3, 1, 2);
    const float r12c23 = minor2(m, 1, 2, 2, 3), r12c13 = minor2(m, 1, 2, 1, 3), r12c12 = minor2(m, 1, 2, 1, 2);
    const float r13c03 = minor2(m, 1, 3, 0, 3), r13c02 = minor2(m, 1, 3, 0, 2), r12c03 = minor2(m, 1, 2, 0, 3);
    const float r12c02 = minor2(m, 1, 2, 0, 2), r13c01 = minor2(m, 1, 3, 0, 1), r12c01 = minor2(m, 1, 2, 0, 1);

    auto cof = [](float a, float A, float b, float B, float c, float C) { return a * A - b * B + c * C; };
    const auto& e = m.e;

    const float c00 = cof(e[1][1], r23c23, e[1][2], r23c13, e[1][3], r23c12);
    const float c01 = cof(e[1][0], r23c23, e[1][2], r23c03, e[1][3], r23c02);
    const float c02 = cof(e[1][0], r23c13, e[1][1], r23c03, e[1][3], r23c01);
    const float c03 = cof(e[1][0], r23c12, e[1][1], r23c02, e[1][2], r23c01);

    float det = e[0][0] * c00 - e[0][1] * c01 + e[0][2] * c02 - e[0][3] * c03;
    det = 1 / det;

    Mat4 o;
    o.e[0][0] = det *  c00;
    o.e[0][1] = det * -cof(e[0][1], r23c23, e[0][2], r23c13, e[0][3], r23c12);
    o.e[0][2] = det *  cof(e[0][1], r13c23, e[0][2], r13c13, e[0][3], r13c12);
    o.e[0][3] = det * -cof(e[0][1], r12c23, e[0][2], r12c13, e[0][3], r12c12);
    o.e[1][0] = det * -c01;
    o.e[1][1] = det *  cof(e[0][0], r23c23, e[0][2], r23c03, e[0][3], r23c02);
    o.e[1][2] = det * -cof(e[0][0], r13c23, e[0][2], r13c03, e[0][3], r13c02);
    o.e[1][3] = det *  cof(e[0][0], r12c23, e[0][2], r12c03, e[0][3], r12c02);
    o.e[2][0] = det *  c02;
    o.e[2][1] = det * -cof(e[0][0], r23c13, e[0][1], r23c03, e[0][3], r23c01);
    o.e[2][2] = det *  cof(e[0][0], r13c13, e[0][1], r13c03, e[0][3], r13c01);
    o.e[2][3] = det * -cof(e[0][0], r12c13, e[0][1], r12c03, e[0][3], r12c01);
    o.e[3][0] = det * -c03;
    o.e[3][1] = det *  cof(e[0][0], r23c12, e[0][1], r23c02, e[0][2], r23c01);
    o.e[3][2] = det * -cof(e[0][0], r13c12, e[0][1], r13c02, e[0][2], r13c01);
    o.e[3][3] = det *  cof(e[0][0], r12c12, e[0][1], r12c02, e[0][2], r12c01);
    return o;
}

// fx::perspective(fovy, w, h, near, far) read as glm::perspectiveFov, right-handed,
// depth -1..1 (Camera.cpp:171).
Mat4 perspective(float fovy_rad, float width, float height, float z_near, float z_far)
{
    const float h = std::cos(0.5f * fovy_rad) / std::sin(0.5f * fovy_rad);
    const float w = h * height / width;
    Mat4 m = Mat4::zero();
    m.e[0][0] = w;
    m.e[1][1] = h;
    m.e[2][2] = -(z_far + z_near) / (z_far - z_near);
    m.e[2][3] = -1.f;
    m.e[3][2] = -(2.f * z_far * z_near) / (z_far - z_near);
    return m;
}

// fx::lookat(eye, at, up) read as glm::lookAt, right-handed (Camera.cpp:180).
Mat4 lookat(Vec3 eye, Vec3 at, Vec3 up)
{
    const Vec3 f = normalize(at - eye);
    const Vec3 s = normalize(cross(f, up));
    const Vec3 u = cross(s, f);
    Mat4 m = Mat4::identity();
    m.e[0][0] = s.x;  m.e[1][0] = s.y;  m.e[2][0] = s.z;
    m.e[0][1] = u.x;  m.e[1][1] = u.y;  m.e[2][1] = u.z;
    m.e[0][2] = -f.x; m.e[1][2] = -f.y; m.e[2][2] = -f.z;
    m.e[3][0] = -dot(s, eye);
    m.e[3][1] = -dot(u, eye);
    m.e[3][2] = dot(f, eye);
    return m;
}

void fill_camera(float fov_deg, float nearclip, float farclip, uint32_t width, uint32_t height,
                 Vec3 pos, Vec3 dir, luma_camera* out)
{
    const float fov_rad = fov_deg * 0.01745329251994329577f;         // fx::radians
    Mat4 projection = perspective(fov_rad, static_cast<float>(width), static_cast<float>(height),
                                  nearclip, farclip);                // Camera.cpp:171
    Mat4 view = lookat(pos, pos + dir, Vec3{ 0.f, 1.f, 0.f });       // Camera.cpp:178-180
    const Mat4 pinv = inverse(projection);                           // Camera.cpp:173
    const Mat4 vinv = inverse(view);                                 // Camera.cpp:182
    out->pos[0] = pos.x; out->pos[1] = pos.y; out->pos[2] = pos.z;
    std::memcpy(out->proj_inv, pinv.e, sizeof pinv.e);
    std::memcpy(out->view_inv, vinv.e, sizeof vinv.e);
    out->focus_depth = 10.f;                                         // camera.h:33
    out->show_depth = 1u;                                            // camera.h:35
}

inline uint64_t splitmix64(uint64_t& state)
{
    uint64_t z = (state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

inline void pcg3d(uint32_t& x, uint32_t& y, uint32_t& z)
{
    x = x * 1664525u + 1013904223u;
    y = y * 1664525u + 1013904223u;
    z = z * 1664525u + 1013904223u;
    x += y * z; y += z * x; z += x * y;
    x ^= x >> 16; y ^= y >> 16; z ^= z >> 16;
    x += y * z; y += z * x; z += x * y;
}

inline float unit(uint32_t bits) { return static_cast<float>(bits >> 8) * 5.9604644775390625e-8f; }

}  // namespace

void seed_keys(uint64_t seed, uint32_t keys[3])
{
    uint64_t s = seed;
    const uint64_t a = splitmix64(s);
    const uint64_t b = splitmix64(s);
    keys[0] = static_cast<uint32_t>(a);
    keys[1] = static_cast<uint32_t>(a >> 32);
    keys[2] = static_cast<uint32_t>(b);
}

void camera_default(float fov_deg, float nearclip, float farclip, uint32_t width, uint32_t height,
                    luma_camera* out)
{
    // Camera::Camera, Camera.cpp:80-81
    fill_camera(fov_deg, nearclip, farclip, width, height, Vec3{ 0.f, 0.f, 0.f }, Vec3{ 0.f, 0.f, -1.f }, out);
}

void camera_from_pose(float fov_deg, float nearclip, float farclip, uint32_t width, uint32_t height,
                      const float pos[3], float pitch, float yaw, luma_camera* out)
{
    const float quarter_pi = 3.14159265358979323846f / 4;
    pitch = std::fmin(std::fmax(pitch, -quarter_pi), quarter_pi);    // Camera.cpp:135
    const float cp = std::cos(pitch), sp = std::sin(pitch);
    const float cy = std::cos(yaw), sy = std::sin(yaw);
    const float base[3] = { 0.f, 0.f, -1.0f };                       // Camera.cpp:143
    const Vec3 dir = {                                               // Camera.cpp:147-152
         (base[0] * cy + base[2] * sy),
        -(base[0] * sp * sy + base[1] * cp - base[2] * sp * cy),
         (base[0] * cp * sy + base[1] * sp + base[2] * cp * cy),
    };
    fill_camera(fov_deg, nearclip, farclip, width, height, Vec3{ pos[0], pos[1], pos[2] }, dir, out);
}

void scene_default(luma_sphere out[5])
{
    // renderer.h:62-70; vector order s, a, q, r, t
    out[0] = luma_sphere{ 0.f, .5f, -10.f, 1.0f, 0.f, 0.f, 1.f, 1.f, .001f, .4f };
    out[1] = luma_sphere{ 0.f, 102.f, 0.f, 100.f, .6f, .6f, .6f, 1.f, .1f, .5f };
    out[2] = luma_sphere{ 3.f, .5f, -10.f, 1.0f, 0.f, 1.f, 0.f, 1.f, .001f, .4f };
    out[3] = luma_sphere{ 6.f, .5f, -10.f, 1.0f, 1.f, 0.f, 0.f, 1.f, .001f, .4f };
    out[4] = luma_sphere{ 3.f, .5f, -23.f, 10.0f, 1.f, 1.f, 1.f, 1.f, .4f, 0.f };
}

void scene_synthetic(luma_sphere* out, uint32_t n, uint64_t seed)
{
    if (n == 0) { return; }
    uint32_t keys[3];
    seed_keys(seed, keys);
    out[0] = luma_sphere{ 0.f, 102.f, 0.f, 100.f, .6f, .6f, .6f, 1.f, .1f, .5f };   // ground
    const float metallic_by_residue[3] = { .001f, .1f, .4f };
    for (uint32_t i = 1; i < n; i++) {
        uint32_t h[3][3];
        for (uint32_t draw = 0; draw < 3; draw++) {
            h[draw][0] = i ^ keys[0]; h[draw][1] = draw ^ keys[1]; h[draw][2] = keys[2];
            pcg3d(h[draw][0], h[draw][1], h[draw][2]);
        }
        luma_sphere s;
        s.cx = std::fmaf(unit(h[0][0]), 80.f, -40.f);
        s.cy = std::fmaf(unit(h[0][1]), 21.5f, -20.f);
        s.cz = std::fmaf(unit(h[0][2]), 80.f, -90.f);
        s.radius = std::fmaf(unit(h[1][0]), .6f, .2f);
        s.roughness = unit(h[1][1]) * .5f;
        s.dr = std::fmaf(unit(h[2][0]), .9f, .1f);
        s.dg = std::fmaf(unit(h[2][1]), .9f, .1f);
        s.db = std::fmaf(unit(h[2][2]), .9f, .1f);
        s.albedo = 1.f;
        s.metallic = metallic_by_residue[i % 3u];
        out[i] = s;
    }
}

bool plan_gather(int n, const int* device_ids, uint32_t width, uint32_t height, GatherPlan* out)
{
    if (!out || n < 1 || n > kMaxGroupMembers || width == 0 || height == 0 || height < (uint32_t)n) { return false; }
    GatherPlan& P = *out;
    P = GatherPlan{};
    P.n_members = n;
    constexpr uint32_t kStripeRows = 15;   // 1080, 2160 and 4320 are multiples of 15 * {1, 2, 4, 8}
    P.striped = (n > 1 && height % (kStripeRows * (uint32_t)n) == 0) ? 1u : 0u;
    P.stripe_rows = P.striped ? kStripeRows : 0u;
    uint64_t staged = 0;
    for (int i = 0; i < n; i++) {
        const int dev = device_ids ? device_ids[i] : i;
        if (dev < 0) { return false; }
        int r = 0;
        while (r < P.n_ranks && P.rank_device[r] != dev) { r++; }
        if (r == P.n_ranks) { P.rank_device[P.n_ranks++] = dev; }
        GatherMember& m = P.member[i];
        m.device = dev;
        m.comm_rank = r;
        if (P.striped) {
            m.row_begin = (uint32_t)i * kStripeRows;
            m.rows = height / (uint32_t)n;
        } else {
            m.row_begin = (uint32_t)((uint64_t)i * height / (uint32_t)n);
            m.rows = (uint32_t)((uint64_t)(i + 1) * height / (uint32_t)n) - m.row_begin;
        }
        m.pixels = (uint64_t)m.rows * width;
        m.recv_offset = P.striped ? staged : (uint64_t)m.row_begin * width;
        staged += m.pixels;
    }
    return true;
}

}  // namespace luma
