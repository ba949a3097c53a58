#include "renderer.h"

#include <chrono>
#include <cstdlib>
#include <string>

#include "arguments.h"
#include "log.h"

namespace luma {

namespace {

[[noreturn]] void die(const char* where)
{
    log(std::string(where) + ": " + luma_hip_last_error());
    std::abort();   // render_to is noexcept with no status channel (renderer.h:76): fail loudly
}

}  // namespace

Renderer::Renderer() noexcept
    : camera(70.0f, 0.1f, 100.0f, _options.width, _options.height)   // Renderer.cpp:164
{
    spheres.resize(5);
    luma_scene_default(reinterpret_cast<luma_sphere*>(spheres.data()));   // renderer.h:62-70

    if (_options.backend != Backend::HIP) {
        log("this build has only the HIP backend (--backend=hip); the CPU loop is the reference's own Renderer.cpp");
        std::abort();
    }
    // --gpus=N: N devices of this process share the frame (stripes or bands, luma_hip.h);
    // --devices=a,b,... names them.  The default is the current device alone.
    std::vector<int> ids = _options.devices;
    if (ids.empty()) {
        int count = 0;
        if (luma_hip_device_count(&count) != LUMA_OK) { die("luma_hip_device_count"); }
        if (_options.gpus > static_cast<std::uint32_t>(count)) {
            log("--gpus=" + std::to_string(_options.gpus) + " but only " + std::to_string(count) + " HIP device(s) are visible");
            std::abort();
        }
        for (std::uint32_t i = 0; i < _options.gpus; i++) { ids.push_back(static_cast<int>(i)); }
    }
    if (luma_hip_group_create(&group_, static_cast<int>(ids.size()), ids.data()) != LUMA_OK) { die("luma_hip_group_create"); }
    if (luma_hip_group_set_frame(group_, _options.width, _options.height) != LUMA_OK) { die("luma_hip_group_set_frame"); }
}

Renderer::~Renderer() { luma_hip_group_destroy(group_); }

void Renderer::render_to(std::uint32_t* target, olc::PixelGameEngine* pge) noexcept
{
    // pge->Clear / DrawCircle (Renderer.cpp:368, :445) are display only and stay out of this path.
    const auto begin = std::chrono::steady_clock::now();             // fx::Timer, :370

    camera.update(frametime, pge);                                   // :375
    if (scene_dirty_) {
        if (luma_hip_group_set_scene(group_, reinterpret_cast<const luma_sphere*>(spheres.data()),
                                     static_cast<std::uint32_t>(spheres.size())) != LUMA_OK) { die("luma_hip_group_set_scene"); }
        scene_dirty_ = false;
    }
    const luma_camera cam = camera.abi();
    if (luma_hip_group_set_camera(group_, &cam) != LUMA_OK) { die("luma_hip_group_set_camera"); }
    if (camera.moved) {                                              // :377-390
        frame_count = 1.f;
        if (luma_hip_group_reset(group_) != LUMA_OK) { die("luma_hip_group_reset"); }
        camera.moved = false;
    }

    luma_params p{};
    p.width = _options.width;
    p.height = _options.height;
    p.samples = _options.samples;
    p.bounces = _options.bounces;
    p.paths = _options.paths;
    p.mode = _options.mode == RenderMode::PATHTRACE ? LUMA_MODE_PATHTRACE : LUMA_MODE_RAYTRACE;
    p.spp = 1;                                                       // one render_to == one frame
    p.flags = _options.flags;
    p.seed = _options.seed;
    if (_options.gather_rccl) {
        // bands gathered on the root GPU over RCCL, then one D2H copy of the whole frame
        if (luma_hip_group_render_device(group_, &p, nullptr, &stats_) != LUMA_OK) { die("luma_hip_group_render_device"); }
        if (luma_hip_group_read_frame(group_, target) != LUMA_OK) { die("luma_hip_group_read_frame"); }
    } else if (luma_hip_group_render(group_, &p, target, &stats_) != LUMA_OK) { die("luma_hip_group_render"); }   // :392-467

    frametime = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - begin).count();   // :469
    frame_count += 1.f;                                              // :470
}

bool Renderer::sync_accumulated_data()
{
    accumulated_data.resize(static_cast<std::size_t>(_options.width) * _options.height);
    return luma_hip_group_read_accum(group_, accumulated_data.data()->data()) == LUMA_OK;
}

}  // namespace luma
