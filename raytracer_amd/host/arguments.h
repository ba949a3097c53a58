// arguments.h -- luma::Options / luma::Arguments as in the reference (arguments.h:9-49), plus
// the flags this backend adds: --backend, --gpus, --devices, --gather, --seed, --frames, --output, --flags, --shm,
// --pos/--pitch/--yaw.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace luma {

enum class RenderMode { RAYTRACE, PATHTRACE };          // arguments.h:9-13
enum class Context { INTERACTIVE, DIRECTX, HEADLESS };  // arguments.h:21-26
enum class Backend { CPU, HIP };                        // new: which loop render_to runs

struct Options {
    // the reference's fields (arguments.h:35-41); it leaves the integers uninitialised,
    // here they default to the author's habitual run (raytracer.vcxproj.user:4)
    std::uint32_t width = 500, height = 500;
    std::uint32_t samples = 1, bounces = 4, paths = 1;
    RenderMode mode = RenderMode::RAYTRACE;
    Context context = Context::INTERACTIVE;
    // additions
    Backend backend = Backend::CPU;
    std::uint32_t gpus = 1;         // devices that share one frame (one process, luma_hip_group_*)
    std::vector<int> devices;       // explicit device ids for them (default 0..gpus-1; an id may repeat)
    bool gather_rccl = false;       // --gather=rccl: assemble the frame on GPU 0 over RCCL (xGMI), then one copy to the
                                    // host; default --gather=pcie: every GPU copies its rows into the host frame
    std::uint64_t seed = 0x4C554D41ull;
    std::uint32_t frames = 1;       // headless: frames to accumulate before writing the image
    std::uint32_t flags = 0;        // LUMA_FLAG_* (0 == reference behaviour)
    std::string output = "output.ppm";
    bool have_pose = false;         // headless stand-in for the fly-cam keys (Camera.cpp:107-129):
    float pos[3] = { 0.f, 0.f, 0.f }, pitch = 0.f, yaw = 0.f;   //   --pos=x,y,z --pitch=rad --yaw=rad
    std::string shm;                // headless: POSIX shared-memory name that receives every frame ("" = off)
};

extern Options _options;

class Arguments {
public:
    // Same contract as the reference's parse (arguments.cpp:94-232): `--key=value` options,
    // anything malformed is logged and skipped, nothing throws.  Also accepts `--key value`.
    void parse(const std::vector<std::string>& arguments);
};

}  // namespace luma
