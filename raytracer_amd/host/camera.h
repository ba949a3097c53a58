// camera.h -- luma::Camera with the reference's public surface (camera.h:11-45).  The matrix
// arithmetic is the C-ABI's (luma_camera_default / luma_camera_from_pose); the W*H ray table
// of the reference is not kept on the host: the kernels form the same directions in registers.
#pragma once
#include <array>
#include <cstdint>
#include <vector>

#include "luma_hip.h"

namespace olc { class PixelGameEngine; }   // display only; never dereferenced here

namespace luma {

using vec3 = std::array<float, 3>;
using mat4 = std::array<float, 16>;        // fx::mat4 m[i][j] as [i*4+j]

class Camera {
public:
    Camera(float fov, float nearclip, float farclip, std::uint32_t width, std::uint32_t height) noexcept;

    // Camera::update (Camera.cpp:92-167).  Key handling needs a window and is out of scope;
    // with pge == nullptr (the headless call, main.cpp:267) the reference only recomputes
    // `dir` from pitch/yaw and reports moved = false.  Pose changes queued with move_to()
    // are applied here and reported as moved = true, like a key press would be.
    bool update(float ts, olc::PixelGameEngine* pge) noexcept;

    void move_to(const vec3& position, float new_pitch, float new_yaw) noexcept;

    // Camera::rays (camera.h:40) on demand: computed by the GPU (luma_hip_generate_rays).
    std::vector<vec3> rays_from(luma_ctx* ctx) const;

    luma_camera abi() const noexcept;      // what the render loop reads

public:
    mat4 projection_inverse{}, view_inverse{};
    float fov = 70.0f;
    float nearclip = 0.1f, farclip = 100.0f;
    bool moved = false;
    float pitch = 0.f, yaw = 0.f;
    float depth = 10.f;
    bool show_depth = true;
    vec3 pos{ 0.f, 0.f, 0.f }, dir{ 0.f, 0.f, -1.f };
    std::uint32_t width = 0, height = 0;

private:
    bool pending_move_ = false;
    void refresh(const luma_camera& c) noexcept;
};

}  // namespace luma
