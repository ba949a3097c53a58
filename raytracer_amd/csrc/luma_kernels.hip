// luma_kernels.hip -- gfx950 (MI355X / CDNA4) kernels for the Luma per-pixel render loop.
//
// Replaces the inner loops of Renderer::render_to (Renderer.cpp:366-471) and its callees
// render_pixel (:289-364), trace_ray (:234-287), indirect_illumination (:182-216),
// reflect_intersection (:218-232), plus Camera::recompute_rays (Camera.cpp:185-216), all
// fused into one launch per render call.  Written for wave64 / 4xSIMD32 CUs: FP32 VALU
// work, no MFMA (the path has no dense contraction).
//
// Arithmetic contract (DESIGN.md "Arithmetic contract"): IEEE binary32, RNE, subnormals
// kept, built with -ffp-contract=off; every fusion is an explicit __builtin_fmaf.  div and
// sqrt are hipcc's correctly rounded forms (default -fhip-fp32-correctly-rounded-divide-sqrt).
// The closest-hit scan uses the half-b quadratic (b' = b/2, d' = d/4), which is bit-identical
// to the reference's b = 2*dot, d = b*b - 4ac form because scaling by powers of two commutes
// with rounding.
#include "luma_device.hpp"
#include "luma_instances.h"

namespace luma {

#ifndef LUMA_SINGLE_TU
// the render kernels are instantiated in the luma_inst_*.hip files
LUMA_LIST_ALL(LUMA_KERNEL_EXTERN)
#endif

// Per-sphere records for the scan, rebuilt when the scene or the camera position changes:
// records[i]     = (camera.pos - centre, |camera.pos - centre|^2 - r^2)   (Renderer.cpp:244,:248)
// records[n + i] = (centre, radius)
__global__ void prepare_records_kernel(const float* __restrict__ cx, const float* __restrict__ cy,
                                       const float* __restrict__ cz, const float* __restrict__ rad,
                                       uint32_t n, float px, float py, float pz, float4* __restrict__ records)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) { return; }
    const f3 c = mk3(cx[i], cy[i], cz[i]);
    const float r = rad[i];
    const f3 diff = sub3(mk3(px, py, pz), c);
    records[i] = make_float4(diff.x, diff.y, diff.z, fma_(-r, r, dot3(diff, diff)));
    records[n + i] = make_float4(c.x, c.y, c.z, r);
}

// Camera::recompute_rays (Camera.cpp:185-216) as a table, for read-back / parity only.
__global__ void generate_rays_kernel(const RenderArgs A, float* __restrict__ rays)
{
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t yl = blockIdx.y;
    const uint32_t y = global_row(A, yl);
    if (x >= A.width || yl >= A.rows) { return; }
    const f3 d = camera_ray(&A, x, y);
    float* o = rays + 3 * ((size_t)yl * A.width + x);
    o[0] = d.x; o[1] = d.y; o[2] = d.z;
}

// The root's step after the band gather when the frame was cut into interleaved stripes
// (luma_hip_set_frame_striped): frame row y belongs to rank (y / stripe) % n and is row
// (y / (stripe * n)) * stripe + y % stripe of that rank's compact band.  One pixel per thread, rows
// are contiguous on both sides, so loads and stores are coalesced dwords; 8.3 MB each way at 1080p.
__global__ void deinterleave_kernel(const uint32_t* __restrict__ staging, uint32_t* __restrict__ frame,
                                    uint32_t width, uint32_t height, uint32_t stripe, uint32_t n_ranks)
{
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t y = blockIdx.y;
    if (x >= width || y >= height) { return; }
    const uint32_t chunk = y / stripe, rank = chunk % n_ranks, q = chunk / n_ranks;
    const uint32_t band_rows = height / n_ranks;
    const size_t src = ((size_t)rank * band_rows + (size_t)q * stripe + (y % stripe)) * width + x;
    frame[(size_t)y * width + x] = staging[src];
}

// How many lanes share one pixel's frames in the ray-per-lane kernel (see SPLIT there): enough to
// put ~400k threads on the GPU, never more than the frames of the call.
bool sphere_parallel_scene(uint32_t n_spheres)
{
#ifdef LUMA_SWEEP_FINE_PCT
    // sweep builds (tools/scene_size_sweep.py): the crossover as a run-time value
    static const uint32_t sp_min = [] { const char* e = getenv("LUMA_SP_MIN"); return e ? (uint32_t)atoi(e) : kSphereParallelMin; }();
    return n_spheres >= sp_min
#else
    return n_spheres >= kSphereParallelMin
#endif
        && (size_t)((n_spheres + 63u) & ~63u) * 2u * sizeof(float4) <= kMaxSceneLdsBytes;
}

// FLAG_GRID launches whose rays from hit points walk the grid as a wave (render_kernel_sp<GRID>, wave_grid_walk)
// instead of lane by lane (render_kernel<GRID>).  Measured on MI355X, 1080p, lane / wave kernel ms
// (tools/grid_threshold_sweep.py, final kernels): at 16 spp 160 spheres 1.74 / 1.88, 250: 2.04 / 1.99, 400: 2.30 / 2.03,
// 511: 2.46 / 2.07; at 2 spp the wave kernel's fixed cost (one resident set of large workgroups, ~0.5 ms per call
// whatever the scene) still shows: 250: 0.40 / 0.50, 511: 0.47 / 0.50.  512 serves both.
bool grid_wave_walk(uint32_t n_spheres)
{
    return n_spheres >= kGridWaveWalkMin;
}

// Launches that hand out wave tiles through RenderArgs::tile_counter (it must be zero at the start)
bool persistent_launch(uint32_t n_spheres, bool grid)
{
    return grid ? grid_wave_walk(n_spheres) : sphere_parallel_scene(n_spheres);
}

// Launches whose one-lane-per-pixel workgroups run the TAKES instances (launch_render): the lanes of a wave take frames of
// each other's pixels.  (Without LUMA_FLAG_NOISE_OFF, which plan_split is not told about: a noise-free launch of many
// frames renders the same frame many times, and is planned as if it could.)
static bool fine_class_launch(uint32_t split, uint32_t spp, uint32_t n_spheres, bool grid, bool from_hit);

static bool takes_launch(uint32_t spp, uint32_t n_spheres, bool grid)
{
    return !grid && n_spheres >= 1 && n_spheres <= kCompileTimeSpheresMax && spp >= 8 && spp <= kStealSppMax;
}

static uint32_t choose_split(size_t band_pixels, uint32_t spp, uint32_t n_spheres, bool grid, bool from_hit)
{
    if (persistent_launch(n_spheres, grid)) { return 1; }
    // measured on MI355X, bands of the 1080p x 64 spp frame (kernel ms for 1 / 2 / 4 lanes per pixel):
    // 2.07M px 2.09 / 2.13 / 2.32, 1.04M px 1.17 / 1.11 / 1.17, 518k px 0.66 / 0.61 / 0.60,
    // 259k px 0.50 / 0.36 / 0.33
    uint32_t split = band_pixels >= 1500000 ? 1u : band_pixels >= 750000 ? 2u : 4u;
    // Round 3: where lanes take frames inside their waves one lane per pixel needs no parking and keeps 56 % of its lanes
    // busy, and wins down to bands that fill the GPU once; kernel ms of bands of 2 / 3 / 4 / 5 / 6 of the 1080p x 64 spp
    // frame (1.04M / 691k / 518k / 415k / 346k px) with 4 lanes: 0.960 / 0.632 / 0.483 / 0.372 / 0.326, with 2: 0.970 / 0.645 /
    // 0.490 / 0.385 / 0.344, with 1 and a fine class: 0.810 / 0.577 / 0.453 / 0.354 / 0.329, with 1 and none: 0.881 / - / 0.489
    // (tools/split_sweep.sh, split_sweep2.sh, profiles/r03/split_sweep.txt)
    // run-time sphere counts: the two-lane tier has gone (its instances neither defer hits nor take frames): 1080p x 64 spp bands
    // of 540 rows of 64 / 200 spheres 4.58 / 12.10 ms with two lanes, 4.43 / 11.42 with four; from hit points 7.05 / 20.09 against
    // 6.06 / 16.76 (tools/mid_scene_bands.py, profiles/r03/mid_scene_takes.txt)
    if (n_spheres > kCompileTimeSpheresMax && !grid && split == 2) { split = 4; }
    if (takes_launch(spp, n_spheres, grid)) {
        const bool fine = fine_class_launch(1, spp, n_spheres, grid, from_hit);
        split = band_pixels >= (fine ? 380000u : 600000u) ? 1u : 4u;
    }
#ifdef LUMA_SWEEP_FINE_PCT
    if (const char* e = getenv("LUMA_FORCE_SPLIT")) { split = (uint32_t)atoi(e); }   // sweep builds: lanes per pixel by hand
#endif
    while (split > spp) { split >>= 1; }
    // the claim word holds two 16-bit counts, and a failed claim still increments its half: the
    // front can reach spp + 1 and the back spp + split - 1
    if (spp > 0x7FFFu) { split = 1; }
    return split ? split : 1u;
}

// The fine class.  Every workgroup of a launch carries the same number of frames, so when the dispatcher runs out
// of workgroups the GPU drains for as long as one of them takes: 0.38 of the 1.80 ms of the 1080p x 64 spp frame were
// spent at half occupancy on average (tools/band_timeline.py).  The last rows of the band are therefore cut into
// workgroups of a quarter (half) the frames each, four lanes per pixel, dispatched after all the others: they fill
// the slots the long workgroups leave and end within 0.11 ms of each other.  Kernel ms by share of rows in the
// fine class (tools/fine_pct_sweep.sh), whole frame: 0 % 1.796, 8 % 1.754, 10 % 1.754, 12 % 1.760, 14 % 1.763,
// 18 % 1.775, 26 % 1.792; band of 2 (two lanes per pixel above the fine rows): 0.975, 0.959, 0.955, 0.952, 0.954,
// 0.960, 0.959.  Bands that already run four lanes per pixel gain nothing from eight in eight-wave workgroups
// (band of 4: 0.513 against 0.506; the whole frame with eight: 1.79 at best) and keep one class.  Only the instances
// with a compile-time sphere count have the class (scenes of up to 8 spheres, camera-origin bounce rays: launch_render).
static bool fine_class_launch(uint32_t split, uint32_t spp, uint32_t n_spheres, bool grid, bool from_hit)
{
    return split <= 4 && spp >= 8 && spp <= 0x7FFFu && !grid && !from_hit && n_spheres >= 1 && n_spheres <= 8;
}

// Slots of the parking planes.  Only the lanes that claim frames from the BACK park (lane 0 of a pixel adds its frames
// directly), and they park frame spp-1-b in slot b: with L lanes per pixel they render (L-1)/L of the frames on average,
// so the planes hold that share plus a margin (a quarter of lane 0's share, at least two frames), never more than
// spp - 1; a back claim beyond the last slot is refused (render_kernel `next_frame`) and lane 0 renders what is left.
// Round 2 kept a slot per frame: 64 instead of 52 at 64 spp x 4 lanes, 40 at 2 lanes, 58 at 8.
static uint32_t park_depth_for(uint32_t spp, uint32_t lanes)
{
    if (lanes <= 1 || spp <= 1) { return 0; }
    const uint32_t share = (uint32_t)(((uint64_t)spp * (lanes - 1) + lanes - 1) / lanes);
#ifdef LUMA_SWEEP_FINE_PCT
    if (const char* e = getenv("LUMA_PARK_MARGIN_DIV")) { return std::min(spp - 1u, share + std::max(2u, spp / ((uint32_t)atoi(e) * lanes))); }
#endif
    return std::min(spp - 1u, share + std::max(2u, spp / (4u * lanes)));
}

SplitPlan plan_split(uint32_t width, uint32_t rows, uint32_t spp, uint32_t n_spheres, bool grid, bool from_hit,
                     size_t park_budget)
{
    SplitPlan plan;
    plan.fine_row0 = rows;
    plan.split = choose_split((size_t)width * rows, spp, n_spheres, grid, from_hit);
    // the most lanes per pixel this launch may run (the fine class has twice the coarse one's, four at least)
    const auto lanes_most = [&](uint32_t split) {
        return fine_class_launch(split, spp, n_spheres, grid, from_hit) ? (split == 4 ? 8u : 4u) : split;
    };
    size_t per_pixel = (size_t)park_depth_for(spp, lanes_most(plan.split)) * 3u * sizeof(float);   // three planes of slots
    if (plan.split > 1) {
        plan.wanted_bytes = per_pixel * width * rows;
        if (plan.wanted_bytes > park_budget) {                   // one lane per pixel (a fine class may still fit)
            plan.split = 1;
            plan.refused = true;
            per_pixel = (size_t)park_depth_for(spp, lanes_most(1)) * 3u * sizeof(float);
        }
    }
    if (fine_class_launch(plan.split, spp, n_spheres, grid, from_hit)) {
        // The drain lasts one coarse workgroup and runs half empty on average, so the fine class should hold about
        // half a round of the coarse waves: a share of 0.45 / rounds, where rounds = coarse waves / wave slots
        // (256 CUs x 28; whole config #3 frame and its band of 2: 4.5 rounds, 10 %, the measured optimum above)
        const double rounds = (double)width * rows / 64.0 * plan.split / (256.0 * 28.0);
        // (one lane per pixel above the class: those waves' lanes take frames of each other's pixels, STEAL in render_kernel,
        //  and end closer together, so less is left to fill: 0 % 1.643, 4 % 1.610, 6 % 1.602, 8 % 1.607, 10 % 1.611, 15 % 1.620,
        //  profiles/r03/fine_pct_steal.txt)
        double share = rounds > 0.0 ? 0.45 / rounds : 0.0;
        // (and bands of one or two rounds want more than that rule leaves them: band of 4, 1.13 rounds, 0.476 ms with 24 % of its
        //  rows in the class, 0.462 with 30 %, 0.453 with 40 %, 0.457 with 50 %; band of 3, 1.5 rounds, 0.581 / 0.577 / 0.588 with
        //  18 / 30 / 40 %; band of 2, 2.26 rounds: 0.861 / 0.809 / 0.820 with 6 / 12 / 20 %)
        const bool takes = plan.split == 1 && takes_launch(spp, n_spheres, grid);
        if (takes) { share = 0.27 / rounds + 0.15 * std::max(0.0, 2.0 - rounds); }
        // (from nine rounds up the drain is too small a part of the launch to pay for the parked frames: config #4's
        //  whole frame, 18 rounds of 256 frames, 25.10 ms with one class, 25.48 with 2 % of its rows in a second;
        //  one of its eight bands, 2.3 rounds: 3.74 -> 3.40, tools/fine_config4.sh)
        share = share < 0.05 ? 0.0 : std::min(share, takes ? 0.40 : 0.30);
        // Four lanes per pixel above: eight in the fine class, two of each pixel side by side in every wave of a
        // four-wave workgroup (HALF in render_kernel; eight-wave workgroups lose more than the class gains).  Those
        // workgroups cost more per frame, and a few of them are enough to fill the drain of 0.1 ms workgroups:
        // bands of 4 / 8 of the config #3 frame 0.507 / 0.278 ms with one class, 0.497 / 0.275 with 2 % of the rows
        // in the class, 0.496 / 0.264 with 5 %, 0.498 / 0.266 with 10 %, 0.502 / 0.267 with 20 % (tools/fine8_share_sweep.sh)
        if (plan.split == 4) { share = 0.05; }
#ifdef LUMA_SWEEP_FINE_PCT
        if (const char* e = getenv("LUMA_FINE_PCT")) { share = atoi(e) / 100.0; }
#endif
        size_t fine_rows = (size_t)((double)rows * share + 0.999);
        // with one lane per pixel above them only the fine rows park: as many of them as the budget holds
        if (plan.split == 1) { fine_rows = std::min(fine_rows, park_budget / std::max<size_t>(per_pixel * width, 1)); }
        // whole coarse workgroups (4 rows each) above the class: rounded towards fewer fine rows where their parking is
        // what the budget limits, towards more where the whole band parks anyway
        const uint32_t above = rows - (uint32_t)std::min<size_t>(fine_rows, rows);
        const uint32_t row0 = plan.split == 1 ? (above + 3u) & ~3u : above & ~3u;
        if (row0 < rows) {
            plan.fine = plan.split == 4 ? 8 : 4;
#ifdef LUMA_SWEEP_FINE_PCT
            if (const char* e = getenv("LUMA_FINE_LANES")) { plan.fine = (uint32_t)atoi(e); }
#endif
            plan.fine_row0 = row0;
        }
    }
    plan.park_row0 = plan.split > 1 ? 0u : plan.fine_row0;
    plan.park_pixels = (rows - plan.park_row0) * width;
    plan.park_depth = park_depth_for(spp, std::max(plan.split, plan.fine));
#ifdef LUMA_SWEEP_FINE_PCT
    if (getenv("LUMA_PARK_FULL")) { plan.park_depth = spp > 1 ? spp - 1 : 0; }   // sweep builds: a slot per frame, as in round 2
#endif
    plan.park_bytes = (plan.split > 1 || plan.fine) ? (size_t)plan.park_depth * 3u * sizeof(float) * plan.park_pixels : 0;
    return plan;
}

void apply_plan(RenderArgs& args, const SplitPlan& plan, float* parking)
{
    args.split = plan.split;
    args.fine = plan.fine;
    args.fine_row0 = plan.fine_row0;
    args.park_row0 = plan.park_row0;
    args.park_pixels = plan.park_pixels;
    args.park_depth = plan.park_depth;
    args.split_scratch = plan.park_bytes ? parking : nullptr;
}

// ---------------------------------------------------------------------------------
// Scratch of luma_hip_render_device_noalloc (include/luma_hip.h): device memory that exists from
// the moment the code object is loaded, so that entry point never calls hipMalloc.  In host code
// the name `scratch` is only a handle for hipGetSymbolAddress, never a device address: kernels get
// the pointer that static_scratch_address() returns, as they get hipMalloc'ed buffers elsewhere.
// ---------------------------------------------------------------------------------
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

hipError_t static_scratch_address(void** ptr)
{
    constexpr int kMaxDevices = 64;
    static std::mutex lock;
    static void* per_device[kMaxDevices] = {};                   // each GPU has its own copy of the array
    *ptr = nullptr;
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) { return e; }
    if (dev < 0 || dev >= kMaxDevices) { return hipErrorInvalidDevice; }
    std::lock_guard<std::mutex> guard(lock);
    if (!per_device[dev]) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));        // address on the current device
        if (e != hipSuccess) { return e; }
        if (!p) { return hipErrorInvalidSymbol; }
        per_device[dev] = p;
    }
    *ptr = per_device[dev];
    return hipSuccess;
}

// ---------------------------------------------------------------------------------
// Launchers (host).  Shapes are validated by the C-ABI layer before these are called.
// ---------------------------------------------------------------------------------

size_t scene_lds_bytes(uint32_t n_spheres) { return (size_t)n_spheres * 2u * sizeof(float4); }

hipError_t launch_prepare_records(const float* cx, const float* cy, const float* cz, const float* rad,
                                  uint32_t n, const float pos[3], float4* records, hipStream_t stream)
{
    if (n == 0) { return hipSuccess; }
    const uint32_t threads = 256, blocks = (n + threads - 1) / threads;
    hipLaunchKernelGGL(prepare_records_kernel, dim3(blocks), dim3(threads), 0, stream,
                       cx, cy, cz, rad, n, pos[0], pos[1], pos[2], records);
    return hipGetLastError();
}

hipError_t launch_render(const RenderArgs& args, hipStream_t stream)
{
    const uint32_t rows = args.rows;
    const uint32_t n = args.n_spheres;
    const bool noise = !(args.flags & FLAG_NOISE_OFF), from_hit = (args.flags & FLAG_ORIGIN_FROM_HIT) != 0;
    using kernel_t = void (*)(const RenderArgs);

    // Large scenes: sphere-parallel scan, records (padded to 64) in LDS.
    const size_t sp_bytes = (size_t)((n + 63u) & ~63u) * 2u * sizeof(float4);
    kernel_t k;
    size_t lds;
    int waves, split = 1, fine = 0;
    const bool use_grid = (args.flags & FLAG_GRID) != 0 && n > 0;
    const bool sphere_parallel = persistent_launch(n, use_grid);
    if (sphere_parallel && use_grid) {
        // wave_grid_walk: the general records, the cell offsets and the cells' sphere lists (16-bit words) all in
        // LDS when they fit beside the waves' exchange areas -- one 16-wave workgroup per CU; else 8-wave workgroups
        // with the cell offsets only (<= 26 KiB), when refresh_grid found those small enough
        const GridArgs& g = args.grid;
        const size_t table_slots = (g.n_cells + 7u) / 8u + (g.n_items + 7u) / 8u + 16u * kCoopSlotsPerWave;
        const size_t wave_slots = kCoopSlotsPerWave + kQueueSlots / 4u;   // tables in global memory: sphere indices need whole words
        const bool small_ids = n < 65536u && g.n_items < 65536u;
        const int resident = !small_ids ? 0 : (table_slots + n) * sizeof(float4) <= 156u * 1024u ? 2
                                            : table_slots * sizeof(float4) <= 156u * 1024u ? 1 : 0;
        waves = resident ? 16 : 8;
        lds = resident ? (table_slots + (resident == 2 ? n : 0u)) * sizeof(float4)
                       : ((((size_t)g.lds_cells + 7u) / 8u) + (size_t)waves * wave_slots) * sizeof(float4);
#define LUMA_COOP(W, R) { { render_kernel_sp<kTileW, W, false, false, true, R>, render_kernel_sp<kTileW, W, false, true, true, R> }, \
                          { render_kernel_sp<kTileW, W, true, false, true, R>,  render_kernel_sp<kTileW, W, true, true, true, R> } }
        static const kernel_t coop_table[3][2][2] = { LUMA_COOP(8, 0), LUMA_COOP(16, 1), LUMA_COOP(16, 2) };
#undef LUMA_COOP
        k = coop_table[resident][noise][from_hit];
    } else if (sphere_parallel) {
        // The kernel needs ~90 VGPRs, so a SIMD holds 5 waves: five 4-wave workgroups per CU
        // (20 waves) fit while each needs <= 32 KiB of the 160 KiB LDS; beyond that one 16-wave
        // workgroup per CU (a second would need 8 waves per SIMD).  640 -> 641 spheres cost 19 %
        // when the switch sat at 20 KiB (tools/scene_size_sweep.py).
        lds = sp_bytes;
        waves = lds > 32 * 1024 ? 16 : 4;
#define LUMA_SP(W) { { render_kernel_sp<kTileW, W, false, false>, render_kernel_sp<kTileW, W, false, true> }, \
                     { render_kernel_sp<kTileW, W, true, false>,  render_kernel_sp<kTileW, W, true, true> } }
        static const kernel_t sp_table[2][2][2] = { LUMA_SP(4), LUMA_SP(16) };
#undef LUMA_SP
        k = sp_table[waves == 16][noise][from_hit];
    } else {
        // Small scenes: one lane = one ray, records + materials in LDS (56 B per sphere); scenes
        // too large for LDS scan from global memory (wave-uniform addresses).
        const size_t all_bytes = (size_t)n * (2 * sizeof(float4) + sizeof(float4) + sizeof(float2));
        const int scene = all_bytes <= 20 * 1024 ? SCENE_LDS_ALL : SCENE_GLOBAL;
        lds = scene == SCENE_LDS_ALL ? all_bytes : 0;
        waves = 4;
        split = args.split;
#define LUMA_K(S, P, G) { { render_kernel<kLaneTileW, 4, S, false, false, P, G>, render_kernel<kLaneTileW, 4, S, false, true, P, G> }, \
                          { render_kernel<kLaneTileW, 4, S, true, false, P, G>,  render_kernel<kLaneTileW, 4, S, true, true, P, G> } }
        static const kernel_t table[3][2][2][2] = {
            { LUMA_K(SCENE_GLOBAL, 1, false), LUMA_K(SCENE_LDS_ALL, 1, false) },
            { LUMA_K(SCENE_GLOBAL, 2, false), LUMA_K(SCENE_LDS_ALL, 2, false) },
            { LUMA_K(SCENE_GLOBAL, 4, false), LUMA_K(SCENE_LDS_ALL, 4, false) },
        };
        static const kernel_t grid_table[3][2][2] = {
            LUMA_K(SCENE_GLOBAL, 1, true), LUMA_K(SCENE_GLOBAL, 2, true), LUMA_K(SCENE_GLOBAL, 4, true),
        };
#undef LUMA_K
        k = table[split == 4 ? 2 : split == 2 ? 1 : 0][scene == SCENE_LDS_ALL][noise][from_hit];
        // Frames taken inside waves for run-time sphere counts too, with pools of 128 slots per wave (8.4 KB per workgroup) so that
        // seven workgroups still fit a CU beside the scene's records: 1080p x 64 frames of 24 / 64 / 100 / 200 / 256 spheres
        // 3.57 / 7.94 / 12.05 / 23.4 / 29.9 -> 3.42 / 7.42 / 11.16 / 21.8 / 29.1 ms, from hit points 5.59 / 13.2 / 20.0 / 39.8 / 50.8 ->
        // 5.26 / 12.3 / 18.6 / 37.0 / 49.2; at 319 spheres it loses 4 % (and 256-slot pools lose 36 % there: four workgroups per
        // CU) -- tools/mid_scene_probe.py, profiles/r03/mid_scene_takes.txt
        if (LUMA_TAKES_GENERAL && scene == SCENE_LDS_ALL && n <= kTakesGeneralSpheresMax && noise && split == 1 && !use_grid
            && args.spp >= kStealSppMin && args.spp <= kStealSppMax) {
            static const kernel_t takes_general[2] = { render_kernel<kLaneTileW, 4, SCENE_LDS_ALL, true, false, 1, false, 0, 0, true>,
                                                       render_kernel<kLaneTileW, 4, SCENE_LDS_ALL, true, true, 1, false, 0, 0, true> };
            k = takes_general[from_hit];
        }
        // Scenes of up to 16 spheres (the reference's own has 5): instances with the sphere count as a compile-time
        // constant, their scans fully unrolled (config #3: 1.93 -> 1.84 ms, profiles/r02/ab_unrolled_scan.txt).  Round 3
        // took them from 8 to 16 spheres and to LUMA_FLAG_ORIGIN_FROM_HIT (tools/ns_reach_ab.py, profiles/r03/ns_reach_ab.txt:
        // 12 spheres 2.19 -> 2.01 ms, band of 8 0.352 -> 0.315; the default scene from hit points 2.65 -> 2.51, 0.41 -> 0.38).
        if (scene == SCENE_LDS_ALL && !use_grid && n >= 1 && n <= kCompileTimeSpheresMax) {
#define LUMA_NS(P, N) { { render_kernel<kLaneTileW, 4, SCENE_LDS_ALL, false, false, P, false, N>, render_kernel<kLaneTileW, 4, SCENE_LDS_ALL, false, true, P, false, N> }, \
                        { render_kernel<kLaneTileW, 4, SCENE_LDS_ALL, true, false, P, false, N>,  render_kernel<kLaneTileW, 4, SCENE_LDS_ALL, true, true, P, false, N> } }
#define LUMA_NS16(P) { LUMA_NS(P, 1), LUMA_NS(P, 2), LUMA_NS(P, 3), LUMA_NS(P, 4), LUMA_NS(P, 5), LUMA_NS(P, 6), LUMA_NS(P, 7), LUMA_NS(P, 8), \
                       LUMA_NS(P, 9), LUMA_NS(P, 10), LUMA_NS(P, 11), LUMA_NS(P, 12), LUMA_NS(P, 13), LUMA_NS(P, 14), LUMA_NS(P, 15), LUMA_NS(P, 16) }
            static const kernel_t ns_table[3][16][2][2] = { LUMA_NS16(1), LUMA_NS16(2), LUMA_NS16(4) };
#undef LUMA_NS16
#undef LUMA_NS
            k = ns_table[split == 4 ? 2 : split == 2 ? 1 : 0][n - 1][noise][from_hit];
            // one lane per pixel and enough frames, not too many, for a wave's lanes to even each other out: the TAKES instances
            const bool takes = noise && split == 1 && args.spp >= kStealSppMin && args.spp <= kStealSppMax;
            if (takes) {
#define LUMA_NT(N) { render_kernel<kLaneTileW, 4, SCENE_LDS_ALL, true, false, 1, false, N, 0, true>, render_kernel<kLaneTileW, 4, SCENE_LDS_ALL, true, true, 1, false, N, 0, true> }
                static const kernel_t takes_table[16][2] = { LUMA_NT(1), LUMA_NT(2), LUMA_NT(3), LUMA_NT(4), LUMA_NT(5), LUMA_NT(6), LUMA_NT(7), LUMA_NT(8),
                                                             LUMA_NT(9), LUMA_NT(10), LUMA_NT(11), LUMA_NT(12), LUMA_NT(13), LUMA_NT(14), LUMA_NT(15), LUMA_NT(16) };
#undef LUMA_NT
                k = takes_table[n - 1][from_hit];
            }
            if (args.fine) {                                     // (plan_split: camera-origin bounce rays, up to 8 spheres)
#define LUMA_NF(W, P, F, N) { render_kernel<kLaneTileW, W, SCENE_LDS_ALL, false, false, P, false, N, F>, \
                              render_kernel<kLaneTileW, W, SCENE_LDS_ALL, true, false, P, false, N, F> }
#define LUMA_NF8(W, P, F) { LUMA_NF(W, P, F, 1), LUMA_NF(W, P, F, 2), LUMA_NF(W, P, F, 3), LUMA_NF(W, P, F, 4), \
                            LUMA_NF(W, P, F, 5), LUMA_NF(W, P, F, 6), LUMA_NF(W, P, F, 7), LUMA_NF(W, P, F, 8) }
                static const kernel_t fine_table[3][8][2] = { LUMA_NF8(4, 1, 4), LUMA_NF8(4, 2, 4), LUMA_NF8(4, 4, 8) };
                k = fine_table[split == 4 ? 2 : split == 2 ? 1 : 0][n - 1][noise];
                if (takes) {
#define LUMA_NTF(N) render_kernel<kLaneTileW, 4, SCENE_LDS_ALL, true, false, 1, false, N, 4, true>
                    static const kernel_t takes_fine_table[8] = { LUMA_NTF(1), LUMA_NTF(2), LUMA_NTF(3), LUMA_NTF(4), LUMA_NTF(5), LUMA_NTF(6), LUMA_NTF(7), LUMA_NTF(8) };
#undef LUMA_NTF
                    k = takes_fine_table[n - 1];
                }
#ifdef LUMA_SWEEP_FINE_PCT
                if (args.fine == 8 && split < 4 && n == 5) {     // sweep builds: eight lanes under one or two
                    static const kernel_t f8[2][2] = { LUMA_NF(4, 1, 8, 5), LUMA_NF(4, 2, 8, 5) };
                    k = f8[split == 2][noise];
                }
                if (args.fine == 4 && split == 4 && n == 5) {    // and a fine class of the same four lanes: same kernel as one class
                    static const kernel_t f4[2] = LUMA_NF(4, 4, 4, 5);
                    k = f4[noise];
                }
#endif
                fine = (int)args.fine;
#undef LUMA_NF8
#undef LUMA_NF
            }
        }
        if (use_grid) {                                          // records and materials stay in global memory
            // the grid's cell offsets go to LDS as 16-bit words when six workgroups' copies fit a CU
            // (the grid variants run six waves per SIMD) and the offsets fit 16 bits
            lds = args.grid.lds_cells ? (((size_t)args.grid.lds_cells * sizeof(uint16_t) + 15u) & ~(size_t)15u) : 0;
            k = grid_table[split == 4 ? 2 : split == 2 ? 1 : 0][noise][from_hit];
        }
    }
    const int wpk = waves / split;                               // waves that tile pixels (the rest split frames)
    const int wgx = wg_waves_x(wpk), wgy = wpk / wgx;
    const int tile_w = sphere_parallel ? kTileW : kLaneTileW, tile_h = 64 / tile_w;
    dim3 grid((args.width + wgx * tile_w - 1) / (wgx * tile_w), (rows + wgy * tile_h - 1) / (wgy * tile_h));
    RenderArgs with_classes = args;
    if (fine) {
        // coarse workgroups first, then the fine ones, in one row of blocks (the dispatcher goes through it in order)
        // (more lanes per pixel than waves: a workgroup of the fine class is one row of one tile, see HALF)
        const bool half = fine > waves;
        const int wpkf = half ? 1 : waves / fine, wgxf = wg_waves_x(wpkf), wgyf = wpkf / wgxf;
        const int fine_tile_h = half ? 1 : tile_h;
        const uint32_t fine_rows = rows - args.fine_row0;
        with_classes.coarse_gx = grid.x;
        with_classes.coarse_wgs = grid.x * (args.fine_row0 / (uint32_t)(wgy * tile_h));   // fine_row0 is a multiple of 4 rows
        with_classes.fine_gx = (args.width + wgxf * tile_w - 1) / (wgxf * tile_w);
        grid = dim3(with_classes.coarse_wgs + with_classes.fine_gx * ((fine_rows + wgyf * fine_tile_h - 1) / (wgyf * fine_tile_h)), 1);
    }
    if (sphere_parallel) {
        // persistent waves: one resident set of workgroups, tiles handed out by a counter
        int cus = 256;
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess) { (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev); }
        const size_t by_lds = (160u * 1024u) / (lds ? lds : 1);
        const size_t per_cu = std::max<size_t>(1, std::min<size_t>(32 / waves, by_lds));
        const uint32_t n_tiles = ((args.width + tile_w - 1) / tile_w) * ((rows + tile_h - 1) / tile_h);
        // (small frames are handed out row by row by the linear-scan kernel: render_kernel_sp)
        const size_t want = ((size_t)n_tiles * (use_grid ? 1 : tile_h) + waves - 1) / waves;
        grid = dim3((unsigned)std::max<size_t>(1, std::min<size_t>(want, per_cu * (size_t)cus)), 1);
    }
#ifdef LUMA_SWEEP_FINE_PCT
    // sweep builds: unused LDS per workgroup, to hold fewer waves per SIMD than the registers allow (occupancy probe)
    if (const char* e = getenv("LUMA_LDS_PAD_KB")) { lds += (size_t)atoi(e) * 1024u; }
#endif
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { return e; }
    }
    hipLaunchKernelGGL(k, grid, dim3(waves * 64), lds, stream, with_classes);
    return hipGetLastError();
}

hipError_t launch_generate_rays(const RenderArgs& args, float* rays, hipStream_t stream)
{
    const uint32_t rows = args.rows;
    const dim3 grid((args.width + 255) / 256, rows);
    hipLaunchKernelGGL(generate_rays_kernel, grid, dim3(256), 0, stream, args, rays);
    return hipGetLastError();
}

hipError_t launch_deinterleave(const uint32_t* staging, uint32_t* frame, uint32_t width, uint32_t height,
                               uint32_t stripe, uint32_t n_ranks, hipStream_t stream)
{
    const dim3 grid((width + 255) / 256, height);
    hipLaunchKernelGGL(deinterleave_kernel, grid, dim3(256), 0, stream, staging, frame, width, height, stripe, n_ranks);
    return hipGetLastError();
}

}  // namespace luma
