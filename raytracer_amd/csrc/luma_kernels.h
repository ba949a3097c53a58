// luma_kernels.h -- kernel argument block and launcher prototypes shared by
// luma_kernels.hip (device) and luma_abi.cpp (the C-ABI layer).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace luma {

enum : uint32_t {
    FLAG_NOISE_OFF       = 1u,
    FLAG_ORIGIN_FROM_HIT = 2u,
    FLAG_CLAMP_PACK      = 4u,
    FLAG_NO_SHOW_DEPTH   = 8u,
    FLAG_PACK_RGBA8      = 16u,
    FLAG_GRID            = 32u,
};
enum : uint32_t { MODE_RAYTRACE = 0, MODE_PATHTRACE = 1 };

// gfx950 has 160 KiB of LDS per CU; keep a little for the compiler.  2 x 16 B per sphere
// => up to 5056 spheres are scanned from LDS, larger scenes from global memory.
static constexpr size_t kMaxSceneLdsBytes = 158 * 1024;

// Scenes with at least this many spheres (and whose padded records fit in LDS) use the
// sphere-parallel scan (render_kernel_sp); smaller ones the ray-per-lane scan.
static constexpr uint32_t kSphereParallelMin = 320;   // measured crossover (tools/scene_size_sweep.py)

// Scenes of up to this many spheres run ray-per-lane instances with the sphere count as a template constant.
static constexpr uint32_t kCompileTimeSpheresMax = 16;
// launches with one lane per pixel and up to this many frames per call: the lanes of a wave take frames of each
// other's pixels when their own run out (STEAL in render_kernel; plan_split sizes the fine class for it)
static constexpr uint32_t kStealSppMax = 128;
// ... and from this many: 1080p calls of 2 / 3 / 4 / 7 frames 0.134 / 0.169 / 0.200 / 0.284 -> 0.123 / 0.150 / 0.177 / 0.255 ms
// (tools/low_spp_probe.py, profiles/r03/small_probes.txt); the lanes-per-pixel rule and the fine class start at 8 frames
#ifndef LUMA_STEAL_SPP_MIN
#define LUMA_STEAL_SPP_MIN 2
#endif
static constexpr uint32_t kStealSppMin = LUMA_STEAL_SPP_MIN;
#ifndef LUMA_TAKES_GENERAL
#define LUMA_TAKES_GENERAL 1
#endif
#ifndef LUMA_TAKES_GENERAL_MAX
#define LUMA_TAKES_GENERAL_MAX 256
#endif
static constexpr uint32_t kTakesGeneralSpheresMax = LUMA_TAKES_GENERAL_MAX;   // (8.4 KB of pools + 56 B per sphere) x 7 workgroups <= 160 KB

// FLAG_GRID: scenes with at least this many spheres walk the grid wave-cooperatively (wave_grid_walk).
static constexpr uint32_t kGridWaveWalkMin = 512;

// The ray counter is kCounterSlots 64-bit slots, one per 64-byte line (stride in u64).
static constexpr uint32_t kCounterSlots = 256;
static constexpr uint32_t kCounterStride = 8;

// Uniform grid over the spheres (luma_grid.h), used under FLAG_GRID.
struct GridArgs {
    uint32_t dims[3];
    float    org[3];
    float    h, inv_h;
    const uint32_t* cell_start;    // dims.x * dims.y * dims.z + 1
    const uint32_t* items;         // sphere indices per cell, ascending
    const uint32_t* large;         // spheres tested against every ray
    uint32_t n_large;
    uint32_t lds_cells;            // entries of cell_start to stage in LDS as 16-bit words (0: read it from global memory)
    uint32_t n_cells, n_items;     // entries of cell_start / items
    // direction grid around camera.pos (luma_grid.h HostViewGrid), for the rays that start there
    uint32_t view_res;             // bins per cube-map face edge; 0 = none (walk the uniform grid instead)
    float    view_half_res;        // view_res / 2
    const uint32_t* view_start;    // 6 * view_res^2 + 1
    const uint2*    view_items;    // per bin, nearest first: (sphere index, float bits of a lower bound of its distance from camera.pos)
    const uint32_t* view_large;
    uint32_t n_view_large;
};

// Passed by value as the kernel argument (lands in SGPRs / the kernarg segment).
struct RenderArgs {
    // frame geometry: this launch renders `rows` rows of a width x height frame; local row yl is
    // global row  row_begin + (yl / stripe) * pitch + yl % stripe.  A contiguous band is one
    // stripe (stripe == rows); interleaved stripes have pitch == stripe * number of ranks.
    uint32_t width, height, row_begin, rows, stripe, pitch;
    // luma::Options snapshot (arguments.h:35-41)
    uint32_t bounces, paths, mode, flags, spp;
    uint32_t frame_index;       // frames accumulated before this launch
    float    frame_count_last;  // frame_count of the last frame of this launch (renderer.h:53)
    float    inv_samples;       // 1.f / _options.samples (Renderer.cpp:212)
    uint32_t key[3];            // counter-RNG keys derived from the seed
    // camera (camera.h:23-37)
    float    cam_pos[3];
    float    proj_inv[16];
    float    view_inv[16];
    float    focus_depth;
    uint32_t show_depth;
    // scene
    uint32_t n_spheres;
    const float4* records;      // [0,n): (camera.pos-centre, c)   [n,2n): (centre, radius)
    const float4* mat0;         // diffuse rgb, albedo
    const float2* mat1;         // metallic, roughness
    // outputs (band-local)
    float*    acc[3];           // accumulation planes r, g, b
    uint32_t* target;           // 0x00RRGGBB
    uint32_t* target_mirror;    // the same pixels a second time, or null: page-locked host memory written by the kernel itself,
    uint32_t  mirror_is_frame;  // a band laid out like `target` (0) or the whole width x height frame (1: indexed by frame pixel)
    // frame splitting (ray-per-lane kernel): lanes per pixel (1, 2 or 4) and the planes where
    // lanes 1.. park their frames: park_depth x 3 x park_pixels floats
    uint32_t  split;
    float*    split_scratch;
    // two classes of workgroups in one launch (SplitPlan): rows [0, fine_row0) with `split` lanes per pixel, dispatched
    // first, rows [fine_row0, rows) with `fine` lanes per pixel; fine == 0: one class, fine_row0 == rows
    uint32_t  fine, fine_row0;
    uint32_t  coarse_wgs, coarse_gx, fine_gx;   // filled in by launch_render: workgroups of the coarse class, workgroups per row of each
    uint32_t  park_row0, park_pixels;           // the parking planes cover rows [park_row0, rows): park_pixels pixels each
    uint32_t  park_depth;                       // slots per pixel: frame spp-1-b of a pixel is parked in slot b, b < park_depth
    uint32_t* tile_counter;        // sphere-parallel kernel: next wave tile, zeroed before every launch
    unsigned long long* counters;  // kCounterSlots x kCounterStride: [0] rays traced, [1] sphere tests; summed by the host
    GridArgs  grid;                // only read under FLAG_GRID
};

// luma_hip_render_device_noalloc: the scratch of the render kernels (ray counters, the tile counter
// of the sphere-parallel kernel, the parking planes of SPLIT renders) in one file-scope __device__
// array of luma_kernels.hip instead of hipMalloc memory.  Layout: [0, kStaticCounterBytes) counters
// + tile counter, [kStaticCounterBytes, kStaticScratchBytes) parking planes.
static constexpr size_t kStaticScratchBytes = size_t(64) << 20;   // == LUMA_HIP_NOALLOC_SCRATCH_BYTES
static constexpr size_t kStaticCounterBytes = 32 * 1024;
// Device address of that array on the CURRENT device (each GPU holds its own copy at its own
// address); looked up once per device with hipGetSymbolAddress and kept.
hipError_t static_scratch_address(void** ptr);

size_t     scene_lds_bytes(uint32_t n_spheres);
bool       sphere_parallel_scene(uint32_t n_spheres);
bool       persistent_launch(uint32_t n_spheres, bool grid);
// How a launch of the ray-per-lane kernel spreads the frames of a pixel over lanes (render_kernel, SPLIT / FINE)
struct SplitPlan {
    uint32_t split = 1;            // lanes per pixel (of the coarse class when there are two)
    uint32_t fine = 0;             // lanes per pixel of the fine class, 0: none
    uint32_t fine_row0 = 0;        // first band row of the fine class (== rows when none)
    uint32_t park_row0 = 0;        // rows [park_row0, rows) have lanes that park frames
    uint32_t park_pixels = 0;
    uint32_t park_depth = 0;       // slots per parked pixel (back claims only; see park_depth_for)
    size_t   park_bytes = 0;       // park_depth x 3 planes x park_pixels floats; 0: nothing parks
    bool     refused = false;      // more than one lane per pixel was wanted, but its wanted_bytes of parking exceed the budget
    size_t   wanted_bytes = 0;
};
// park_budget: the most parking memory the caller will provide; a fine class that does not fit is left out
SplitPlan  plan_split(uint32_t width, uint32_t rows, uint32_t spp, uint32_t n_spheres, bool grid, bool from_hit,
                      size_t park_budget);
void       apply_plan(RenderArgs& args, const SplitPlan& plan, float* parking);
hipError_t launch_prepare_records(const float* cx, const float* cy, const float* cz, const float* rad,
                                  uint32_t n, const float pos[3], float4* records, hipStream_t stream);
hipError_t launch_render(const RenderArgs& args, hipStream_t stream);
hipError_t launch_generate_rays(const RenderArgs& args, float* rays, hipStream_t stream);
// Interleaved-stripe bands gathered back to back in `staging` (member r's compact band, height /
// n_ranks rows, at r * (height / n_ranks) * width) -> the frame in row order.
hipError_t launch_deinterleave(const uint32_t* staging, uint32_t* frame, uint32_t width, uint32_t height,
                               uint32_t stripe, uint32_t n_ranks, hipStream_t stream);

}  // namespace luma
