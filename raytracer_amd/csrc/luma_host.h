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

// How a group of members (one per entry of device_ids, ids may repeat) cuts a width x height frame and how
// luma_hip_group_render_device moves the bands to the root: pure host arithmetic, shared by
// luma_hip_group_set_frame, the RCCL exchange and luma_gather_plan_describe (so the CPU tests check the
// very plan the exchange runs).
struct GatherMember {
    int      device;        // as given
    int      comm_rank;     // index into GatherPlan::rank_device: the communicator rank that sends this band
    uint32_t row_begin;     // first frame row of the band (of its first stripe when striped)
    uint32_t rows;          // rows in the band
    uint64_t pixels;        // rows * width: elements of the send and of the matching receive
    uint64_t recv_offset;   // where the root receives it, in pixels: into the frame (contiguous bands, == row_begin *
                            // width) or into the staging buffer (striped: bands back to back in member order)
};
constexpr int kMaxGroupMembers = 64;
struct GatherPlan {
    int      n_members = 0, n_ranks = 0;
    uint32_t striped = 0;          // 1: interleaved stripes of stripe_rows rows (member i owns stripes i, i + n, ...)
    uint32_t stripe_rows = 0;      // rows per stripe (striped) else 0
    int      rank_device[kMaxGroupMembers];   // distinct devices in order of first appearance; [0] is the root's
    GatherMember member[kMaxGroupMembers];
};
// false: n out of [1, kMaxGroupMembers], a negative device id, a zero dimension or fewer rows than members
bool plan_gather(int n, const int* device_ids, uint32_t width, uint32_t height, GatherPlan* out);

}  // namespace luma
