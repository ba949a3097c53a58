#include "camera.h"

#include <algorithm>
#include <cmath>

namespace luma {

Camera::Camera(float fov_, float nearclip_, float farclip_, std::uint32_t width_, std::uint32_t height_) noexcept
    : fov(fov_), nearclip(nearclip_), farclip(farclip_), width(width_), height(height_)
{
    // Camera.cpp:80-89: dir (0,0,-1), pos (0,0,0), projection + view + their inverses
    luma_camera c{};
    if (width && height && luma_camera_default(fov, nearclip, farclip, width, height, &c) == LUMA_OK) { refresh(c); }
}

void Camera::refresh(const luma_camera& c) noexcept
{
    std::copy(c.proj_inv, c.proj_inv + 16, projection_inverse.begin());
    std::copy(c.view_inv, c.view_inv + 16, view_inverse.begin());
    std::copy(c.pos, c.pos + 3, pos.begin());
}

void Camera::move_to(const vec3& position, float new_pitch, float new_yaw) noexcept
{
    pos = position;
    pitch = new_pitch;
    yaw = new_yaw;
    pending_move_ = true;
}

bool Camera::update(float /*ts*/, olc::PixelGameEngine* /*pge*/) noexcept
{
    moved = pending_move_;                                            // Camera.cpp:98
    pending_move_ = false;
    const float quarter_pi = 3.14159265358979323846f / 4;
    pitch = std::clamp(pitch, -quarter_pi, quarter_pi);               // Camera.cpp:135
    const float cp = std::cos(pitch), sp = std::sin(pitch), cy = std::cos(yaw), sy = std::sin(yaw);
    dir = { -sy, -(sp * cy), -(cp * cy) };                            // Camera.cpp:147-152 with base (0,0,-1)
    if (moved) {                                                      // Camera.cpp:160-164
        luma_camera c{};
        if (luma_camera_from_pose(fov, nearclip, farclip, width, height, pos.data(), pitch, yaw, &c) == LUMA_OK) { refresh(c); }
    }
    return moved;
}

luma_camera Camera::abi() const noexcept
{
    luma_camera c{};
    std::copy(pos.begin(), pos.end(), c.pos);
    std::copy(projection_inverse.begin(), projection_inverse.end(), c.proj_inv);
    std::copy(view_inverse.begin(), view_inverse.end(), c.view_inv);
    c.focus_depth = depth;
    c.show_depth = show_depth ? 1u : 0u;
    return c;
}

std::vector<vec3> Camera::rays_from(luma_ctx* ctx) const
{
    std::vector<vec3> rays(static_cast<std::size_t>(width) * height);
    if (!ctx || luma_hip_generate_rays(ctx, rays.data()->data()) != LUMA_OK) { rays.clear(); }
    return rays;
}

}  // namespace luma
