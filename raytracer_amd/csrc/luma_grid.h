// luma_grid.h -- uniform grid over the scene's spheres (host side).  An optional accelerator for
// the closest-hit scan (LUMA_FLAG_GRID): it only decides WHICH spheres a ray is tested against;
// the test itself and the choice among hits are the reference's (Renderer.cpp:242-285), so the
// frame is the linear scan's, bit for bit.  No counterpart in the reference.
#pragma once

#include <stdint.h>

#include <vector>

#include "luma_hip.h"

namespace luma {

struct HostGrid {
    uint32_t dims[3] = { 1, 1, 1 };
    float org[3] = { 0.f, 0.f, 0.f };       // minimum corner
    float h = 1.f;                          // cell edge
    std::vector<uint32_t> cell_start;       // dims.x * dims.y * dims.z + 1 offsets into items
    std::vector<uint32_t> items;            // sphere indices, ascending inside a cell
    std::vector<uint32_t> large;            // spheres tested against every ray (ground plane, non-finite, ...)
};

// Never fails for finite n; spheres the grid cannot hold go to `large`.
void build_grid(const luma_sphere* spheres, uint32_t n, HostGrid& out);

// Direction grid around ONE origin: every primary and bounce ray of the reference starts at
// camera.pos (Renderer.cpp:303, :416), so for those rays the spheres a ray can hit depend on its
// direction alone.  The directions are binned on the six faces of a cube map, res x res bins per
// face (bin = (face * res + iv) * res + iu; face = 2 * major axis + (negative ? 1 : 0); u, v = the
// other two components in axis order, divided by |major|, in [-1, 1]); a bin lists every sphere that
// some direction of the bin can hit, NEAREST FIRST, each with a lower bound of the distance from the
// origin to any point of it, so a ray stops reading its bin once its best hit is nearer than whatever
// follows (the closest-hit rule -- smallest t, lowest index among equal t -- does not depend on the
// order of the tests).  Rebuilt when the scene or camera.pos changes.  `large`: spheres listed in too
// many bins (the ground) or that contain the origin.
struct HostViewGrid {
    uint32_t res = 1;
    std::vector<uint32_t> cell_start;       // 6 * res * res + 1
    std::vector<uint32_t> items;            // sphere indices, by ascending `nearest` inside a bin
    std::vector<float> nearest;             // per item: no point of the sphere is closer to the origin than this
    std::vector<uint32_t> large;
};
void build_view_grid(const luma_sphere* spheres, uint32_t n, const float origin[3], HostViewGrid& out);

}  // namespace luma
