// renderer.h -- luma::Renderer with the reference's public surface (renderer.h:49-85):
// same types, same default scene, same render_to signature.  With --backend hip the frame
// loop runs in libluma_hip.so; there is no CPU loop in this tree (the reference's own
// Renderer.cpp is the CPU backend), so any other backend is refused loudly.
#pragma once
#include <cstdint>
#include <vector>

#include "camera.h"
#include "luma_hip.h"

namespace luma {

struct Ray { vec3 pos, dir; };                                   // renderer.h:12-15
struct Material { vec3 diffuse; float albedo, metallic, roughness; };   // renderer.h:23-29
struct Sphere { vec3 pos; float radius; Material material; };    // renderer.h:31-36
static_assert(sizeof(Sphere) == sizeof(luma_sphere), "Sphere must keep the reference's 40-byte layout");

class Renderer {
public:
    float frametime = 0.f;
    float frame_count = 1.f;
    bool accumulate = true;
    std::vector<vec3> accumulated_data;     // host copy, filled by sync_accumulated_data()
    Camera camera;
    std::vector<Sphere> spheres;            // renderer.h:62-70: s, a, q, r, t

public:
    Renderer() noexcept;
    ~Renderer();
    Renderer(const Renderer&) = delete;
    Renderer& operator=(const Renderer&) = delete;

    void render_to(std::uint32_t* target, olc::PixelGameEngine* pge) noexcept;   // renderer.h:76

    // Call after editing `spheres` (the reference rescans the vector every ray).
    void scene_changed() noexcept { scene_dirty_ = true; }
    // Copies the device-resident accumulation buffer into accumulated_data (renderer.h:56).
    bool sync_accumulated_data();
    const luma_stats& last_stats() const noexcept { return stats_; }
    luma_group* group() const noexcept { return group_; }

private:
    luma_group* group_ = nullptr;           // _options.gpus devices sharing the frame (one is the usual case)
    bool scene_dirty_ = true;
    luma_stats stats_{};
};

}  // namespace luma
