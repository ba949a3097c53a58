#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "luma_grid.h"
#include "luma_host.h"
int main() {
    srand(1);
    for (int it = 0; it < 200; it++) {
        uint32_t n = (uint32_t)(rand() % 3000);
        std::vector<luma_sphere> s(n);
        double scale = std::pow(10.0, (rand() % 9) - 4);
        for (auto& q : s) {
            q.cx = (float)((rand() / (double)RAND_MAX - .5) * 100 * scale); q.cy = (float)((rand() / (double)RAND_MAX - .5) * 20 * scale);
            q.cz = (float)((rand() / (double)RAND_MAX - .5) * 100 * scale); q.radius = (float)((rand() / (double)RAND_MAX) * scale * (rand() % 50 == 0 ? 100 : 1));
            if (rand() % 500 == 0) q.radius = INFINITY;
            if (rand() % 500 == 0) q.cx = NAN;
            if (rand() % 100 == 0) q.radius = 0.f;
        }
        luma::HostGrid g;
        luma::build_grid(s.data(), n, g);
        size_t cells = (size_t)g.dims[0] * g.dims[1] * g.dims[2];
        if (g.cell_start.size() != cells + 1 || g.cell_start.back() != g.items.size()) { printf("bad sizes\n"); return 1; }
    }
    std::vector<luma_sphere> syn(4096);
    luma::scene_synthetic(syn.data(), 4096, 0x4C554D41ull);
    luma_camera c; luma::camera_default(70.f, .1f, 100.f, 1920, 1080, &c);
    float pos[3] = {1, 2, 3}; luma::camera_from_pose(70.f, .1f, 100.f, 64, 36, pos, .3f, -2.f, &c);
    uint32_t k[3]; luma::seed_keys(123, k);
    printf("sanitizer run ok\n");
    return 0;
}
