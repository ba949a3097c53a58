// luma_host.h -- host-only helpers behind the luma_camera_* / luma_scene_* entry points.
#pragma once

#include <stdint.h>

#include "luma_hip.h"

namespace luma {

void seed_keys(uint64_t seed, uint32_t keys[3]);
void camera_default(float fov_deg, float nearclip, float farclip, uint32_t width, uint32_t height,
                    luma_camera* out);
void camera_from_pose(float fov_deg, float nearclip, float farclip, uint32_t width, uint32_t height,
                      const float pos[3], float pitch, float yaw, luma_camera* out);
void scene_default(luma_sphere out[5]);
void scene_synthetic(luma_sphere* out, uint32_t n, uint64_t seed);

}  // namespace luma
