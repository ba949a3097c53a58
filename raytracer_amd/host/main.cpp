// main.cpp -- headless Linux entry point, shaped like the reference's HEADLESS context
// (main.cpp:241-287): construct the Renderer, call render_to once per frame, report
// "[DELIVERED FRAME]:n:ms" per frame, then write the frame with Image::export_to_ppm
// (the reference's Ctrl+P path, main.cpp:95-102).  Display (olc, D3D11) is not built.
//
//   luma_headless --backend hip --gpus 8 --width=1920 --height=1080 --mode=pathtrace --bounces=5
//                 --samples=1 --paths=1 --frames=64 --output=frame.ppm
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <cerrno>
#include <memory>
#include <string>
#include <vector>

#include "arguments.h"
#include "image.h"
#include "log.h"
#include "renderer.h"

namespace {

// The reference's HEADLESS context publishes every frame in a named shared-memory block for an
// external viewer (CreateFileMapping "LumaFramebuffer", main.cpp:249-256, memcpy at :35-41).
// POSIX equivalent: shm_open + mmap; the object is left in place at exit so the last frame
// stays readable.
class FrameShare {
public:
    FrameShare(const std::string& name, std::size_t bytes) : bytes_(bytes)
    {
        if (name.empty()) { return; }
        const std::string path = name[0] == '/' ? name : "/" + name;
        const int fd = shm_open(path.c_str(), O_CREAT | O_RDWR, 0644);
        if (fd < 0 || ftruncate(fd, static_cast<off_t>(bytes)) != 0) {
            std::printf("failed to create the memory-mapped resource: %s\n", std::strerror(errno));   // main.cpp:252
            if (fd >= 0) { close(fd); }
            return;
        }
        void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (p != MAP_FAILED) { view_ = p; }
    }
    ~FrameShare() { if (view_) { munmap(view_, bytes_); } }
    void publish(const std::uint32_t* frame) { if (view_) { std::memcpy(view_, frame, bytes_); } }

private:
    void* view_ = nullptr;
    std::size_t bytes_;
};

}  // namespace

int main(int argc, char** argv)
{
    std::vector<std::string> args(argv + (argc > 0 ? 1 : 0), argv + argc);
    luma::_options.context = luma::Context::HEADLESS;
    luma::_options.backend = luma::Backend::HIP;
    luma::Arguments{}.parse(args);

    if (luma::_options.context != luma::Context::HEADLESS) {
        luma::log("only --context=headless is built on Linux (the interactive and directx contexts need a window)");
        return 2;
    }
    if (luma::_options.width == 0 || luma::_options.height == 0) {
        luma::log("--width and --height must be positive");
        return 2;
    }

    luma::log(std::to_string(luma::_options.width) + "x" + std::to_string(luma::_options.height) + " on "
              + std::to_string(luma::_options.gpus) + " device(s)");
    luma::Renderer renderer{};
    if (luma::_options.have_pose) {      // applied by Camera::update in the first render_to, as a key press would be
        renderer.camera.move_to({ luma::_options.pos[0], luma::_options.pos[1], luma::_options.pos[2] },
                                luma::_options.pitch, luma::_options.yaw);
    }
    const std::size_t pixels = static_cast<std::size_t>(luma::_options.width) * luma::_options.height;
    // main.cpp:259 has `new std::uint32_t[width * height]`; a page-locked frame is written by the render
    // kernels themselves while they run (no copy after them; pageable memory costs 0.5-1.1 ms per 1080p frame)
    struct HostFree { void operator()(std::uint32_t* p) const { luma_hip_host_free(p); } };
    void* raw = nullptr;
    if (luma_hip_host_alloc(&raw, pixels * sizeof(std::uint32_t)) != LUMA_OK) {
        luma::log(std::string("luma_hip_host_alloc: ") + luma_hip_last_error());
        return 1;
    }
    std::unique_ptr<std::uint32_t[], HostFree> framebuffer(static_cast<std::uint32_t*>(raw));
    std::memset(framebuffer.get(), 0, pixels * sizeof(std::uint32_t));

    FrameShare share(luma::_options.shm, pixels * sizeof(std::uint32_t));

    auto then = std::chrono::high_resolution_clock::now();
    double kernel_ms = 0.0;
    unsigned long long rays = 0;
    for (std::uint32_t frame = 0; frame < luma::_options.frames; frame++) {
        renderer.render_to(framebuffer.get(), nullptr);
        const auto now = std::chrono::high_resolution_clock::now();
        const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(now - then).count();
        then = now;
        share.publish(framebuffer.get());                            // main.cpp:274
        kernel_ms += renderer.last_stats().kernel_ms;
        rays += renderer.last_stats().rays;
        std::printf("[DELIVERED FRAME]:%u:%lld\n", frame, static_cast<long long>(ms));
    }
    if (kernel_ms > 0.0) {
        std::printf("[STATS]:frames=%u:rays=%llu:kernel_ms=%.3f:Mrays/s=%.1f\n", luma::_options.frames, rays, kernel_ms,
                    static_cast<double>(rays) / kernel_ms / 1e3);
    }

    luma::Image image{ static_cast<std::int32_t>(luma::_options.width), static_cast<std::int32_t>(luma::_options.height), false };
    image.shadow_from(framebuffer.get());
    const std::string& out = luma::_options.output;
    if (out.size() > 4 && out.compare(out.size() - 4, 4, ".bmp") == 0) { image.export_to_bmp(out); }
    else if (out.size() > 4 && out.compare(out.size() - 4, 4, ".pnm") == 0) { image.export_to_pnm(out); }
    else { image.export_to_ppm(out); }
    luma::log("wrote `" + out + "`");
    return 0;
}
