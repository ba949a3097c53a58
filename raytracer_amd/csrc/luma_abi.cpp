// luma_abi.cpp -- the C-ABI of include/luma_hip.h: context, device memory, launches.
// No exceptions cross the boundary; every HIP failure becomes LUMA_E_HIP with a message.
#include "luma_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "luma_grid.h"
#include "luma_host.h"
#include "luma_kernels.h"
#include "luma_rccl.h"

namespace {

thread_local char g_last_error[512] = "";

constexpr size_t kCounterBytes = sizeof(unsigned long long) * luma::kCounterSlots * luma::kCounterStride;   // per line: rays, grid tests, grid rays
constexpr size_t kCounterAlloc = kCounterBytes + 64;   // + one line for the sphere-parallel kernel's tile counter

int fail(int code, const char* what, const char* detail)
{
    std::snprintf(g_last_error, sizeof g_last_error, "%s: %s", what, detail ? detail : "");
    return code;
}

#define LUMA_HIP_TRY(expr)                                                         \
    do {                                                                           \
        hipError_t e_ = (expr);                                                    \
        if (e_ != hipSuccess) {                                                    \
            return fail(e_ == hipErrorOutOfMemory ? LUMA_E_NOMEM : LUMA_E_HIP,     \
                        #expr, hipGetErrorString(e_));                             \
        }                                                                          \
    } while (0)

// Makes ctx's device current for the scope of one ABI call and restores the caller's.
class DeviceScope {
public:
    explicit DeviceScope(int device) : ok_(false), prev_(-1)
    {
        if (hipGetDevice(&prev_) != hipSuccess) { return; }
        if (prev_ != device && hipSetDevice(device) != hipSuccess) { return; }
        ok_ = true;
    }
    ~DeviceScope()
    {
        if (ok_ && prev_ >= 0) {
            int now = -1;
            if (hipGetDevice(&now) == hipSuccess && now != prev_) { (void)hipSetDevice(prev_); }
        }
    }
    bool ok() const { return ok_; }

private:
    bool ok_;
    int prev_;
};

// luma_hip_render_device_noalloc: every call on a GPU shares that GPU's static scratch array, whatever its context or
// stream.  One event per device, created at luma_hip_create (set-up: the call itself must allocate nothing), orders
// them: a call waits on it before it clears the counters and records it after its launch.
std::mutex g_noalloc_lock;
hipEvent_t g_noalloc_event[64] = {};
hipError_t noalloc_event_for(int device, hipEvent_t* out)      // the device is current
{
    *out = nullptr;
    if (device < 0 || device >= 64) { return hipErrorInvalidDevice; }
    std::lock_guard<std::mutex> guard(g_noalloc_lock);
    if (!g_noalloc_event[device]) {
        hipError_t e = hipEventCreateWithFlags(&g_noalloc_event[device], hipEventDisableTiming);
        if (e != hipSuccess) { g_noalloc_event[device] = nullptr; return e; }
    }
    *out = g_noalloc_event[device];
    return hipSuccess;
}

template <typename T>
void free_dev(T*& p)
{
    if (p) { (void)hipFree(p); p = nullptr; }
}

}  // namespace

struct luma_ctx {
    int device = 0;
    size_t park_limit = size_t(2) << 30;        // luma_hip_set_park_limit
    hipStream_t stream = nullptr;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;

    // scene: structure-of-arrays geometry + packed material vectors
    uint32_t n_spheres = 0;
    float* d_cx = nullptr;
    float* d_cy = nullptr;
    float* d_cz = nullptr;
    float* d_rad = nullptr;
    float4* d_mat0 = nullptr;
    float2* d_mat1 = nullptr;
    float4* d_records = nullptr;   // 2n scan records, rebuilt when scene or camera.pos changes
    bool records_stale = true;
    bool have_scene = false;

    luma_camera camera{};
    bool have_camera = false;

    // frame / band
    uint32_t width = 0, height = 0, row_begin = 0, rows = 0, stripe = 1, pitch = 1;   // see RenderArgs
    float* d_acc = nullptr;        // 3 planes of rows*width
    uint32_t* d_target = nullptr;  // rows*width
    uint32_t frame_index = 0;
    bool have_frame = false;
    hipStream_t last_stream = nullptr;   // stream of the latest render call (the planes are RMW state)
    bool rendered = false;
    float* d_split = nullptr;      // parked frames of split renders (luma_kernels.hip, SPLIT): sized by luma_hip_reserve,
    size_t split_bytes = 0;        //   else grown by the first call that needs more
    uint64_t device_allocs = 0;    // hipMalloc calls made by render calls (luma_hip_alloc_count)
    uint32_t last_split = 1;       // lanes per pixel of the latest launch, and whether it wanted more lanes
    bool last_split_refused = false;   //   than the 2 GiB cap on the parking planes allowed

    unsigned char* static_scratch = nullptr;   // this device's copy of luma_kernels.hip's `scratch` (render_device_noalloc)
    hipEvent_t noalloc_event = nullptr;        // the device's "last use of the static scratch" event (shared, not owned)

    unsigned long long* d_counters = nullptr;  // rays traced since luma_hip_create, in kCounterSlots partial sums
    unsigned long long* h_counters = nullptr;  // pinned
    // Counter totals already read back (drain_counters), and the work read but not yet handed out in
    // a luma_stats.  Sphere tests of linear-scan rays are priced at the scene size they ran against,
    // so luma_hip_set_scene drains the counters before it changes n_spheres.
    unsigned long long rays_seen = 0, grid_tests_seen = 0, grid_rays_seen = 0;
    unsigned long long rays_pending = 0, tests_pending = 0;

    // LUMA_FLAG_GRID: the spheres as given (the grid is built from them on first use) and the grid
    std::vector<luma_sphere> host_spheres;
    bool grid_stale = true;
    luma::GridArgs grid{};
    uint32_t* d_grid_cells = nullptr;
    uint32_t* d_grid_items = nullptr;
    uint32_t* d_grid_large = nullptr;
    bool view_stale = true;                    // the direction grid around camera.pos (camera-origin rays)
    uint32_t* d_view_start = nullptr;
    uint2*    d_view_items = nullptr;
    uint32_t* d_view_large = nullptr;

    size_t band_pixels() const { return static_cast<size_t>(rows) * width; }
};

namespace {

int check_params(const luma_ctx* ctx, const luma_params* p)
{
    if (!ctx || !p) { return fail(LUMA_E_ARG, "render", "null context or params"); }
    if (!ctx->have_scene) { return fail(LUMA_E_STATE, "render", "luma_hip_set_scene not called"); }
    if (!ctx->have_camera) { return fail(LUMA_E_STATE, "render", "luma_hip_set_camera not called"); }
    if (!ctx->have_frame) { return fail(LUMA_E_STATE, "render", "luma_hip_set_frame not called"); }
    if (p->width != ctx->width || p->height != ctx->height) {
        return fail(LUMA_E_ARG, "render", "params width/height differ from luma_hip_set_frame");
    }
    if (p->spp == 0) { return fail(LUMA_E_ARG, "render", "spp must be >= 1"); }
    if (p->mode != LUMA_MODE_RAYTRACE && p->mode != LUMA_MODE_PATHTRACE) {
        return fail(LUMA_E_ARG, "render", "mode must be LUMA_MODE_RAYTRACE or LUMA_MODE_PATHTRACE");
    }
    if (p->mode == LUMA_MODE_PATHTRACE && p->samples == 0) {
        return fail(LUMA_E_ARG, "render", "samples must be >= 1 in pathtrace mode (1/samples, Renderer.cpp:212)");
    }
    if (p->bounces > 4096 || p->paths > 4096) {
        return fail(LUMA_E_ARG, "render", "bounces/paths out of range (max 4096)");
    }
    if (static_cast<uint64_t>(ctx->frame_index) + p->spp > 0xFFFFFFFFull) {
        return fail(LUMA_E_ARG, "render", "frame counter would overflow; call luma_hip_reset");
    }
    return LUMA_OK;
}

luma::RenderArgs make_args(const luma_ctx* ctx, const luma_params* p, uint32_t* target_dev)
{
    luma::RenderArgs a{};
    a.width = ctx->width;
    a.height = ctx->height;
    a.row_begin = ctx->row_begin;
    a.rows = ctx->rows;
    a.stripe = ctx->stripe;
    a.pitch = ctx->pitch;
    if (p) {
        a.bounces = p->bounces;
        a.paths = p->paths;
        a.mode = p->mode;
        a.flags = p->flags;
        a.spp = p->spp;
        a.frame_index = ctx->frame_index;
        a.frame_count_last = static_cast<float>(ctx->frame_index + p->spp - 1u) + 1.f;   // renderer.h:53, Renderer.cpp:470
        a.inv_samples = p->samples ? 1.f / static_cast<float>(p->samples) : 0.f;         // Renderer.cpp:212
        luma::seed_keys(p->seed, a.key);
        a.show_depth = (ctx->camera.show_depth && !(p->flags & LUMA_FLAG_NO_SHOW_DEPTH)) ? 1u : 0u;
    }
    std::memcpy(a.cam_pos, ctx->camera.pos, sizeof a.cam_pos);
    std::memcpy(a.proj_inv, ctx->camera.proj_inv, sizeof a.proj_inv);
    std::memcpy(a.view_inv, ctx->camera.view_inv, sizeof a.view_inv);
    a.focus_depth = ctx->camera.focus_depth;
    a.n_spheres = ctx->n_spheres;
    a.records = ctx->d_records;
    a.mat0 = ctx->d_mat0;
    a.mat1 = ctx->d_mat1;
    const size_t plane = ctx->band_pixels();
    a.acc[0] = ctx->d_acc;
    a.acc[1] = ctx->d_acc ? ctx->d_acc + plane : nullptr;
    a.acc[2] = ctx->d_acc ? ctx->d_acc + 2 * plane : nullptr;
    a.target = target_dev;
    a.target_mirror = nullptr;
    a.mirror_is_frame = 0;
    a.counters = ctx->d_counters;
    a.tile_counter = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(ctx->d_counters) + kCounterBytes);
    luma::apply_plan(a, luma::SplitPlan{}, nullptr);             // one lane per pixel until a plan says otherwise
    a.fine_row0 = ctx->rows;
    a.coarse_wgs = a.coarse_gx = a.fine_gx = 0;
    a.grid = ctx->grid;
    return a;
}

int refresh_records(luma_ctx* ctx, hipStream_t stream)
{
    if (!ctx->records_stale) { return LUMA_OK; }
    LUMA_HIP_TRY(luma::launch_prepare_records(ctx->d_cx, ctx->d_cy, ctx->d_cz, ctx->d_rad, ctx->n_spheres,
                                              ctx->camera.pos, ctx->d_records, stream));
    ctx->records_stale = false;
    return LUMA_OK;
}

// LUMA_FLAG_GRID: build the uniform grid from the scene on first use and after every
// luma_hip_set_scene, and the direction grid around camera.pos whenever the scene or that position
// has changed since (luma_grid.h).
int refresh_grid(luma_ctx* ctx)
{
    if (!ctx->grid_stale && !ctx->view_stale) { return LUMA_OK; }
    auto upload = [](uint32_t*& dst, const std::vector<uint32_t>& src) -> hipError_t {
        const size_t bytes = std::max<size_t>(src.size(), 1) * sizeof(uint32_t);
        hipError_t e = hipMalloc(&dst, bytes);
        if (e == hipSuccess && !src.empty()) { e = hipMemcpy(dst, src.data(), src.size() * sizeof(uint32_t), hipMemcpyHostToDevice); }
        return e;
    };
    luma::GridArgs& a = ctx->grid;
    const uint32_t n = static_cast<uint32_t>(ctx->host_spheres.size());
    if (ctx->grid_stale) {
        luma::HostGrid g;
        try {
            luma::build_grid(ctx->host_spheres.data(), n, g);
        } catch (...) {
            return fail(LUMA_E_NOMEM, "LUMA_FLAG_GRID", "host allocation failed while building the grid");
        }
        LUMA_HIP_TRY(hipDeviceSynchronize());                    // a launch may still walk the old grid
        free_dev(ctx->d_grid_cells); free_dev(ctx->d_grid_items); free_dev(ctx->d_grid_large);
        LUMA_HIP_TRY(upload(ctx->d_grid_cells, g.cell_start));
        LUMA_HIP_TRY(upload(ctx->d_grid_items, g.items));
        LUMA_HIP_TRY(upload(ctx->d_grid_large, g.large));
        ctx->device_allocs += 3;
        for (int i = 0; i < 3; i++) { a.dims[i] = g.dims[i]; a.org[i] = g.org[i]; }
        a.h = g.h;
        a.inv_h = 1.0f / g.h;
        a.cell_start = ctx->d_grid_cells;
        a.items = ctx->d_grid_items;
        a.large = ctx->d_grid_large;
        a.n_large = static_cast<uint32_t>(g.large.size());
        a.n_cells = static_cast<uint32_t>(g.cell_start.size());
        a.n_items = static_cast<uint32_t>(g.items.size());
        // six workgroups per CU (the grid kernels' occupancy) x 26 KiB of 16-bit offsets fit the 160 KiB of LDS
        a.lds_cells = (g.cell_start.size() <= 13312 && g.items.size() < 65536) ? static_cast<uint32_t>(g.cell_start.size()) : 0u;
        ctx->grid_stale = false;
        ctx->view_stale = true;
    }
    if (ctx->view_stale) {
        luma::HostViewGrid v;
        try {
            luma::build_view_grid(ctx->host_spheres.data(), n, ctx->camera.pos, v);
        } catch (...) {
            return fail(LUMA_E_NOMEM, "LUMA_FLAG_GRID", "host allocation failed while building the direction grid");
        }
        LUMA_HIP_TRY(hipDeviceSynchronize());
        free_dev(ctx->d_view_start); free_dev(ctx->d_view_items); free_dev(ctx->d_view_large);
        a.view_res = 0;
        std::vector<uint32_t> pairs;                              // (index, nearest) interleaved: one 8-byte load per item
        try {
            pairs.resize(2 * v.items.size());
            for (size_t k = 0; k < v.items.size(); k++) {
                pairs[2 * k] = v.items[k];
                std::memcpy(&pairs[2 * k + 1], &v.nearest[k], sizeof(float));
            }
        } catch (...) {
            return fail(LUMA_E_NOMEM, "LUMA_FLAG_GRID", "host allocation failed while building the direction grid");
        }
        uint32_t* d_pairs = nullptr;
        LUMA_HIP_TRY(upload(ctx->d_view_start, v.cell_start));
        const hipError_t e_pairs = upload(d_pairs, pairs);
        ctx->d_view_items = reinterpret_cast<uint2*>(d_pairs);    // owned by the context from here on, whatever e_pairs says
        LUMA_HIP_TRY(e_pairs);
        LUMA_HIP_TRY(upload(ctx->d_view_large, v.large));
        ctx->device_allocs += 3;
        a.view_res = v.res;
        a.view_half_res = 0.5f * static_cast<float>(v.res);
        a.view_start = ctx->d_view_start;
        a.view_items = ctx->d_view_items;
        a.view_large = ctx->d_view_large;
        a.n_view_large = static_cast<uint32_t>(v.large.size());
        ctx->view_stale = false;
    }
    return LUMA_OK;
}

// The accumulation planes are read-modify-write state: work touching them on a different
// stream than the previous render call must not overtake it.
int order_after_previous(luma_ctx* ctx, hipStream_t stream)
{
    if (ctx->rendered && ctx->last_stream != stream) { LUMA_HIP_TRY(hipDeviceSynchronize()); }
    return LUMA_OK;
}

constexpr size_t kMaxSplitBytes = size_t(2) << 30;

luma::SplitPlan split_plan(const luma_ctx* ctx, uint32_t spp, bool grid, bool from_hit, size_t park_budget)
{
    return luma::plan_split(ctx->width, ctx->rows, spp, ctx->n_spheres, grid, from_hit, park_budget);
}

int grow_split_planes(luma_ctx* ctx, size_t need)
{
    LUMA_HIP_TRY(hipDeviceSynchronize());
    free_dev(ctx->d_split);
    ctx->split_bytes = 0;
    LUMA_HIP_TRY(hipMalloc(&ctx->d_split, need));
    ctx->split_bytes = need;
    return LUMA_OK;
}

// Enqueue one render call on `stream`.  With `timed`, events bracket the render kernel only.
int enqueue_render(luma_ctx* ctx, const luma_params* p, uint32_t* target_dev, hipStream_t stream, bool timed,
                   uint32_t* target_mirror = nullptr, bool mirror_is_frame = false)
{
    if (int rc = order_after_previous(ctx, stream)) { return rc; }
    if (int rc = refresh_records(ctx, stream)) { return rc; }
    const bool grid = (p->flags & LUMA_FLAG_GRID) != 0 && ctx->n_spheres > 0;
    if (grid) { if (int rc = refresh_grid(ctx)) { return rc; } }
    luma::RenderArgs args = make_args(ctx, p, target_dev);
    args.target_mirror = target_mirror;
    args.mirror_is_frame = mirror_is_frame ? 1u : 0u;
    // (more than 2 GiB of parking is not worth it: such calls run one lane per pixel, reported in luma_stats)
    const luma::SplitPlan plan = split_plan(ctx, p->spp, grid, (p->flags & LUMA_FLAG_ORIGIN_FROM_HIT) != 0, ctx->park_limit);
    ctx->last_split_refused = plan.refused;
    if (plan.park_bytes > ctx->split_bytes) {
        // only when luma_hip_reserve was not called (or named fewer spp): the first such call pays
        if (int rc = grow_split_planes(ctx, plan.park_bytes)) { return rc; }
        ctx->device_allocs++;
    }
    luma::apply_plan(args, plan, ctx->d_split);
    ctx->last_split = args.split;
    if (luma::persistent_launch(ctx->n_spheres, grid)) {
        LUMA_HIP_TRY(hipMemsetAsync(args.tile_counter, 0, sizeof(uint32_t), stream));
    }
    // the events are only recorded for calls that report a luma_stats: each is a packet of its own
    // in the queue and keeps back-to-back launches apart
    if (timed) { LUMA_HIP_TRY(hipEventRecord(ctx->ev_begin, stream)); }
    LUMA_HIP_TRY(luma::launch_render(args, stream));
    if (timed) { LUMA_HIP_TRY(hipEventRecord(ctx->ev_end, stream)); }
    ctx->frame_index += p->spp;
    ctx->last_stream = stream;
    ctx->rendered = true;
    return LUMA_OK;
}

// Reads the device counters (after the work queued on `stream`) and moves what is new into the
// pending totals, priced at the current scene size.
int drain_counters(luma_ctx* ctx, hipStream_t stream)
{
    LUMA_HIP_TRY(hipMemcpyAsync(ctx->h_counters, ctx->d_counters, kCounterBytes, hipMemcpyDeviceToHost, stream));
    LUMA_HIP_TRY(hipStreamSynchronize(stream));
    unsigned long long total = 0, grid_tests = 0, grid_rays = 0;
    for (uint32_t i = 0; i < luma::kCounterSlots; i++) {
        total += ctx->h_counters[i * luma::kCounterStride];
        grid_tests += ctx->h_counters[i * luma::kCounterStride + 1];
        grid_rays += ctx->h_counters[i * luma::kCounterStride + 2];
    }
    const unsigned long long rays = total - ctx->rays_seen;
    const unsigned long long g_rays = grid_rays - ctx->grid_rays_seen, g_tests = grid_tests - ctx->grid_tests_seen;
    ctx->rays_seen = total;
    ctx->grid_rays_seen = grid_rays;
    ctx->grid_tests_seen = grid_tests;
    ctx->rays_pending += rays;
    // a linear-scan ray tests every sphere (Renderer.cpp:242); rays of LUMA_FLAG_GRID launches count their own
    ctx->tests_pending += (rays - g_rays) * ctx->n_spheres + g_tests;
    return LUMA_OK;
}

int collect_stats(luma_ctx* ctx, hipStream_t stream, luma_stats* stats)
{
    if (int rc = drain_counters(ctx, stream)) { return rc; }
    float ms = 0.f;
    LUMA_HIP_TRY(hipEventElapsedTime(&ms, ctx->ev_begin, ctx->ev_end));
    stats->kernel_ms = ms;
    stats->lanes_per_pixel = ctx->last_split;
    stats->split_refused = ctx->last_split_refused ? 1u : 0u;
    stats->rays = ctx->rays_pending;            // since the previous luma_stats
    stats->sphere_tests = ctx->tests_pending;
    ctx->rays_pending = ctx->tests_pending = 0;
    return LUMA_OK;
}

// True for memory the runtime knows as page-locked host memory (hipHostMalloc / hipHostRegister).
bool is_page_locked(const void* p)
{
    hipPointerAttribute_t attr{};
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError();                                 // plain malloc memory: not an error here
        return false;
    }
    return attr.type == hipMemoryTypeHost;
}

// A frame qualifies for the queued (async) hand-over only when its first and last byte are both
// page-locked; a frame that is registered in part takes the blocking copies.
bool frame_is_page_locked(const void* frame, size_t bytes)
{
    return bytes > 0 && is_page_locked(frame) && is_page_locked(static_cast<const char*>(frame) + bytes - 1);
}

// The address at which the current device's kernels can write `host` if all `bytes` of it are page-locked
// (luma_hip_host_alloc), else null: such a target is written by the render kernel itself, next to the context's
// own framebuffer -- the 4 bytes per pixel cross PCIe while other pixels are still being traced, and no copy
// follows the kernel (1080p x 64 spp frame: 2.00 -> 1.84 ms; 1 spp: 0.20 -> 0.18).
uint32_t* kernel_writable(uint32_t* host, size_t bytes)
{
    if (!frame_is_page_locked(host, bytes)) { return nullptr; }
    void* alias = nullptr;
    if (hipHostGetDevicePointer(&alias, host, 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return static_cast<uint32_t*>(alias);
}

// Copies the band in ctx->d_target to its rows of a full host frame: contiguous bands in one
// copy, interleaved stripes as one strided copy (chunk q of `stripe` rows lands `pitch` rows apart).
// For frames the kernels cannot write themselves (kernel_writable); blocking.
int copy_band_to_frame(const luma_ctx* ctx, uint32_t* frame_host)
{
    const size_t row_bytes = static_cast<size_t>(ctx->width) * sizeof(uint32_t);
    uint32_t* dst = frame_host + static_cast<size_t>(ctx->row_begin) * ctx->width;
    if (ctx->stripe == ctx->rows) {
        LUMA_HIP_TRY(hipMemcpy(dst, ctx->d_target, ctx->rows * row_bytes, hipMemcpyDeviceToHost));
    } else {
        LUMA_HIP_TRY(hipMemcpy2D(dst, ctx->pitch * row_bytes, ctx->d_target, ctx->stripe * row_bytes,
                                 ctx->stripe * row_bytes, ctx->rows / ctx->stripe, hipMemcpyDeviceToHost));
    }
    return LUMA_OK;
}

}  // namespace

// What luma_hip_group_render_device needs beyond the member contexts: one RCCL communicator rank
// and one stream per DISTINCT device of the group (the root's device is rank 0), an event per member
// that says "band rendered", and on the root a staging buffer for interleaved-stripe bands plus the
// group's own frame.  Built on first use, torn down with the group.
struct GroupGather {
    std::vector<int> devices;              // distinct devices, [0] = device of member 0 = the root (GatherPlan::rank_device)
    std::vector<ncclComm_t> comms;
    std::vector<hipStream_t> streams;      // the stream each communicator rank's sends / receives run on
    std::vector<hipEvent_t> rendered;      // per member, recorded on the member's stream
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;   // around the exchange, on the root's stream
    uint32_t* staging = nullptr;           // root device; striped partitions only
    size_t staging_bytes = 0;
    uint32_t* frame = nullptr;             // root device; the group's own frame (frame_dev == NULL)
    size_t frame_bytes = 0;
    bool ready = false;
};

struct luma_group {
    std::vector<luma_ctx*> ctx;
    bool have_frame = false;
    bool striped = false;
    luma::GatherPlan plan;                 // bands, communicator ranks and receive offsets (luma_host.h); valid with have_frame
    GroupGather gather;
};

extern "C" {

int luma_hip_abi_version(void) { return LUMA_HIP_ABI_VERSION; }

const char* luma_hip_last_error(void) { return g_last_error; }

const char* luma_hip_strerror(int code)
{
    switch (code) {
        case LUMA_OK:      return "ok";
        case LUMA_E_ARG:   return "invalid argument";
        case LUMA_E_HIP:   return "HIP runtime error";
        case LUMA_E_NOMEM: return "out of memory";
        case LUMA_E_STATE: return "scene, camera or frame not set";
        case LUMA_E_RCCL:  return "RCCL error";
        default:           return "unknown error";
    }
}

int luma_hip_create(luma_ctx** out, int device_id)
{
    if (!out) { return fail(LUMA_E_ARG, "luma_hip_create", "null out pointer"); }
    *out = nullptr;
    int count = 0;
    LUMA_HIP_TRY(hipGetDeviceCount(&count));
    if (count <= 0) { return fail(LUMA_E_HIP, "luma_hip_create", "no HIP device visible"); }
    if (device_id < 0) { LUMA_HIP_TRY(hipGetDevice(&device_id)); }
    if (device_id >= count) { return fail(LUMA_E_ARG, "luma_hip_create", "device_id out of range"); }

    luma_ctx* ctx = new (std::nothrow) luma_ctx();
    if (!ctx) { return fail(LUMA_E_NOMEM, "luma_hip_create", "host allocation failed"); }
    ctx->device = device_id;
    DeviceScope scope(device_id);
    if (!scope.ok()) { delete ctx; return fail(LUMA_E_HIP, "luma_hip_create", "hipSetDevice failed"); }

    auto bail = [&](hipError_t e, const char* what) {
        int rc = fail(e == hipErrorOutOfMemory ? LUMA_E_NOMEM : LUMA_E_HIP, what, hipGetErrorString(e));
        luma_hip_destroy(ctx);
        return rc;
    };
    hipError_t e;
    if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) { return bail(e, "hipStreamCreate"); }
    if ((e = hipEventCreate(&ctx->ev_begin)) != hipSuccess) { return bail(e, "hipEventCreate"); }
    if ((e = hipEventCreate(&ctx->ev_end)) != hipSuccess) { return bail(e, "hipEventCreate"); }
    if ((e = hipMalloc(&ctx->d_counters, kCounterAlloc)) != hipSuccess) { return bail(e, "hipMalloc(counters)"); }
    if ((e = hipHostMalloc(&ctx->h_counters, kCounterBytes)) != hipSuccess) { return bail(e, "hipHostMalloc(counters)"); }
    if ((e = hipMemset(ctx->d_counters, 0, kCounterAlloc)) != hipSuccess) { return bail(e, "hipMemset(counters)"); }
    // the device is current here: this is the address of the static scratch array on THIS GPU
    void* scratch_ptr = nullptr;
    if ((e = luma::static_scratch_address(&scratch_ptr)) != hipSuccess) { return bail(e, "hipGetSymbolAddress(scratch)"); }
    ctx->static_scratch = static_cast<unsigned char*>(scratch_ptr);
    if ((e = noalloc_event_for(device_id, &ctx->noalloc_event)) != hipSuccess) { return bail(e, "hipEventCreate(noalloc)"); }
    *out = ctx;
    return LUMA_OK;
}

void luma_hip_destroy(luma_ctx* ctx)
{
    if (!ctx) { return; }
    DeviceScope scope(ctx->device);
    (void)hipDeviceSynchronize();   // launches may be in flight on the caller's streams too
    free_dev(ctx->d_cx); free_dev(ctx->d_cy); free_dev(ctx->d_cz); free_dev(ctx->d_rad);
    free_dev(ctx->d_mat0); free_dev(ctx->d_mat1); free_dev(ctx->d_records);
    free_dev(ctx->d_acc); free_dev(ctx->d_target); free_dev(ctx->d_counters); free_dev(ctx->d_split);
    free_dev(ctx->d_grid_cells); free_dev(ctx->d_grid_items); free_dev(ctx->d_grid_large);
    free_dev(ctx->d_view_start); free_dev(ctx->d_view_items); free_dev(ctx->d_view_large);
    if (ctx->h_counters) { (void)hipHostFree(ctx->h_counters); }
    if (ctx->ev_begin) { (void)hipEventDestroy(ctx->ev_begin); }
    if (ctx->ev_end) { (void)hipEventDestroy(ctx->ev_end); }
    if (ctx->stream) { (void)hipStreamDestroy(ctx->stream); }
    delete ctx;
}

int luma_hip_set_scene(luma_ctx* ctx, const luma_sphere* spheres, uint32_t n)
{
    if (!ctx) { return fail(LUMA_E_ARG, "luma_hip_set_scene", "null context"); }
    if (n > 0 && !spheres) { return fail(LUMA_E_ARG, "luma_hip_set_scene", "null sphere array"); }
    if (n > (1u << 24)) { return fail(LUMA_E_ARG, "luma_hip_set_scene", "too many spheres (max 2^24)"); }
    DeviceScope scope(ctx->device);
    if (!scope.ok()) { return fail(LUMA_E_HIP, "luma_hip_set_scene", "hipSetDevice failed"); }
    // launches given a caller's stream (luma_hip_render_device) may still read the old arrays
    LUMA_HIP_TRY(hipDeviceSynchronize());
    if (ctx->rendered) { if (int rc = drain_counters(ctx, ctx->stream)) { return rc; } }   // rays so far ran against the old n

    free_dev(ctx->d_cx); free_dev(ctx->d_cy); free_dev(ctx->d_cz); free_dev(ctx->d_rad);
    free_dev(ctx->d_mat0); free_dev(ctx->d_mat1); free_dev(ctx->d_records);
    ctx->n_spheres = 0;
    ctx->have_scene = false;          // until every array below is in place
    ctx->records_stale = true;
    ctx->grid_stale = true;
    try { ctx->host_spheres.assign(spheres, spheres + n); } catch (...) {
        return fail(LUMA_E_NOMEM, "luma_hip_set_scene", "host allocation failed");
    }
    if (n == 0) { ctx->have_scene = true; return LUMA_OK; }   // empty scene: every ray misses (Renderer.cpp:276)

    // AoS (renderer.h:23-36) -> SoA geometry + packed material vectors
    std::vector<float> cx, cy, cz, rad;
    std::vector<float4> m0;
    std::vector<float2> m1;
    try {
        cx.resize(n); cy.resize(n); cz.resize(n); rad.resize(n); m0.resize(n); m1.resize(n);
    } catch (...) {
        return fail(LUMA_E_NOMEM, "luma_hip_set_scene", "host allocation failed");
    }
    for (uint32_t i = 0; i < n; i++) {
        const luma_sphere& s = spheres[i];
        cx[i] = s.cx; cy[i] = s.cy; cz[i] = s.cz; rad[i] = s.radius;
        m0[i] = make_float4(s.dr, s.dg, s.db, s.albedo);
        m1[i] = make_float2(s.metallic, s.roughness);
    }
    const size_t fb = n * sizeof(float);
    LUMA_HIP_TRY(hipMalloc(&ctx->d_cx, fb));
    LUMA_HIP_TRY(hipMalloc(&ctx->d_cy, fb));
    LUMA_HIP_TRY(hipMalloc(&ctx->d_cz, fb));
    LUMA_HIP_TRY(hipMalloc(&ctx->d_rad, fb));
    LUMA_HIP_TRY(hipMalloc(&ctx->d_mat0, n * sizeof(float4)));
    LUMA_HIP_TRY(hipMalloc(&ctx->d_mat1, n * sizeof(float2)));
    LUMA_HIP_TRY(hipMalloc(&ctx->d_records, 2 * static_cast<size_t>(n) * sizeof(float4)));
    LUMA_HIP_TRY(hipMemcpy(ctx->d_cx, cx.data(), fb, hipMemcpyHostToDevice));
    LUMA_HIP_TRY(hipMemcpy(ctx->d_cy, cy.data(), fb, hipMemcpyHostToDevice));
    LUMA_HIP_TRY(hipMemcpy(ctx->d_cz, cz.data(), fb, hipMemcpyHostToDevice));
    LUMA_HIP_TRY(hipMemcpy(ctx->d_rad, rad.data(), fb, hipMemcpyHostToDevice));
    LUMA_HIP_TRY(hipMemcpy(ctx->d_mat0, m0.data(), n * sizeof(float4), hipMemcpyHostToDevice));
    LUMA_HIP_TRY(hipMemcpy(ctx->d_mat1, m1.data(), n * sizeof(float2), hipMemcpyHostToDevice));
    ctx->n_spheres = n;
    ctx->have_scene = true;
    return LUMA_OK;
}

int luma_hip_set_camera(luma_ctx* ctx, const luma_camera* cam)
{
    if (!ctx || !cam) { return fail(LUMA_E_ARG, "luma_hip_set_camera", "null argument"); }
    DeviceScope scope(ctx->device);
    if (!scope.ok()) { return fail(LUMA_E_HIP, "luma_hip_set_camera", "hipSetDevice failed"); }
    if (!ctx->have_camera || std::memcmp(ctx->camera.pos, cam->pos, sizeof cam->pos) != 0) {
        // the scan records embed camera.pos; launches in flight (on any stream) read the old ones
        LUMA_HIP_TRY(hipDeviceSynchronize());
        ctx->records_stale = true;
        ctx->view_stale = true;
    }
    ctx->camera = *cam;
    ctx->have_camera = true;
    return LUMA_OK;
}

namespace {

int set_frame_geometry(luma_ctx* ctx, const char* who, uint32_t width, uint32_t height,
                       uint32_t row_begin, uint32_t rows, uint32_t stripe, uint32_t pitch)
{
    if (static_cast<uint64_t>(width) * height > 0x7FFFFFFFull) { return fail(LUMA_E_ARG, who, "frame larger than 2^31 pixels"); }
    DeviceScope scope(ctx->device);
    if (!scope.ok()) { return fail(LUMA_E_HIP, who, "hipSetDevice failed"); }
    LUMA_HIP_TRY(hipDeviceSynchronize());   // also covers launches on a caller's stream
    free_dev(ctx->d_acc);
    free_dev(ctx->d_target);
    ctx->have_frame = false;
    ctx->width = width; ctx->height = height;
    ctx->row_begin = row_begin; ctx->rows = rows; ctx->stripe = stripe; ctx->pitch = pitch;
    const size_t px = ctx->band_pixels();
    LUMA_HIP_TRY(hipMalloc(&ctx->d_acc, 3 * px * sizeof(float)));
    LUMA_HIP_TRY(hipMalloc(&ctx->d_target, px * sizeof(uint32_t)));
    ctx->have_frame = true;
    return luma_hip_reset(ctx);
}

}  // namespace

int luma_hip_set_frame(luma_ctx* ctx, uint32_t width, uint32_t height, uint32_t row_begin, uint32_t row_end)
{
    if (!ctx) { return fail(LUMA_E_ARG, "luma_hip_set_frame", "null context"); }
    if (width == 0 || height == 0) { return fail(LUMA_E_ARG, "luma_hip_set_frame", "zero width or height"); }
    if (row_begin >= row_end || row_end > height) {
        return fail(LUMA_E_ARG, "luma_hip_set_frame", "need row_begin < row_end <= height");
    }
    const uint32_t rows = row_end - row_begin;
    return set_frame_geometry(ctx, "luma_hip_set_frame", width, height, row_begin, rows, rows, rows);
}

int luma_hip_set_frame_striped(luma_ctx* ctx, uint32_t width, uint32_t height, uint32_t stripe_rows,
                               uint32_t rank, uint32_t n_ranks)
{
    const char* who = "luma_hip_set_frame_striped";
    if (!ctx) { return fail(LUMA_E_ARG, who, "null context"); }
    if (width == 0 || height == 0 || stripe_rows == 0 || n_ranks == 0 || rank >= n_ranks) {
        return fail(LUMA_E_ARG, who, "zero size or rank out of range");
    }
    const uint64_t pitch = static_cast<uint64_t>(stripe_rows) * n_ranks;
    if (pitch > height || height % pitch != 0) {
        return fail(LUMA_E_ARG, who, "height must be a multiple of stripe_rows * n_ranks");
    }
    const uint32_t rows = static_cast<uint32_t>(height / pitch) * stripe_rows;
    return set_frame_geometry(ctx, who, width, height, rank * stripe_rows, rows, stripe_rows, static_cast<uint32_t>(pitch));
}

int luma_hip_reset(luma_ctx* ctx)
{
    if (!ctx) { return fail(LUMA_E_ARG, "luma_hip_reset", "null context"); }
    if (!ctx->have_frame) { return fail(LUMA_E_STATE, "luma_hip_reset", "luma_hip_set_frame not called"); }
    DeviceScope scope(ctx->device);
    if (!scope.ok()) { return fail(LUMA_E_HIP, "luma_hip_reset", "hipSetDevice failed"); }
    if (int rc = order_after_previous(ctx, ctx->stream)) { return rc; }
    const size_t px = ctx->band_pixels();
    LUMA_HIP_TRY(hipMemsetAsync(ctx->d_acc, 0, 3 * px * sizeof(float), ctx->stream));      // Renderer.cpp:381-387
    LUMA_HIP_TRY(hipMemsetAsync(ctx->d_target, 0, px * sizeof(uint32_t), ctx->stream));
    LUMA_HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->frame_index = 0;                                                                  // Renderer.cpp:379
    ctx->last_stream = ctx->stream;
    return LUMA_OK;
}

int luma_hip_set_park_limit(luma_ctx* ctx, uint64_t bytes)
{
    const char* who = "luma_hip_set_park_limit";
    if (!ctx) { return fail(LUMA_E_ARG, who, "null context"); }
    if (bytes > kMaxSplitBytes) { return fail(LUMA_E_ARG, who, "the limit is at most 2 GiB"); }
    ctx->park_limit = static_cast<size_t>(bytes);
    return LUMA_OK;
}

int luma_hip_reserve(luma_ctx* ctx, uint32_t max_spp)
{
    const char* who = "luma_hip_reserve";
    if (!ctx) { return fail(LUMA_E_ARG, who, "null context"); }
    if (!ctx->have_frame || !ctx->have_scene) { return fail(LUMA_E_STATE, who, "scene and frame must be set"); }
    if (max_spp == 0) { return fail(LUMA_E_ARG, who, "max_spp must be >= 1"); }
    DeviceScope scope(ctx->device);
    if (!scope.ok()) { return fail(LUMA_E_HIP, who, "hipSetDevice failed"); }
    // Parking grows with spp, so what max_spp needs covers every call of fewer frames -- as long as the plan at max_spp
    // is not REFUSED (wanted more than kMaxSplitBytes: it then runs one lane per pixel and parks little), in which case a
    // later call of fewer spp that does fit wants more than max_spp did: the largest such call is planned as well.
    // Every kind of launch is covered: with and without LUMA_FLAG_GRID (scenes of 320-511 spheres split only under the
    // grid), with and without LUMA_FLAG_ORIGIN_FROM_HIT (no fine class).
    size_t need = 0;
    for (int grid = 0; grid < 2; grid++) {
        for (int from_hit = 0; from_hit < 2; from_hit++) {
            const luma::SplitPlan at_max = split_plan(ctx, max_spp, grid != 0, from_hit != 0, ctx->park_limit);
            need = std::max(need, at_max.park_bytes);
            if (at_max.refused) {
                uint32_t lo = 0, hi = max_spp;                   // largest spp below max_spp whose plan is not refused
                while (hi - lo > 1) {
                    const uint32_t mid = lo + (hi - lo) / 2;
                    if (split_plan(ctx, mid, grid != 0, from_hit != 0, ctx->park_limit).refused) { hi = mid; } else { lo = mid; }
                }
                if (lo >= 1) { need = std::max(need, split_plan(ctx, lo, grid != 0, from_hit != 0, ctx->park_limit).park_bytes); }
            }
        }
    }
    if (need <= ctx->split_bytes) { return LUMA_OK; }
    return grow_split_planes(ctx, need);
}

int luma_hip_alloc_count(luma_ctx* ctx, uint64_t* count)
{
    if (!ctx || !count) { return fail(LUMA_E_ARG, "luma_hip_alloc_count", "null argument"); }
    *count = ctx->device_allocs;
    return LUMA_OK;
}

int luma_hip_render(luma_ctx* ctx, const luma_params* params, uint32_t* target_host, luma_stats* stats)
{
    if (int rc = check_params(ctx, params)) { return rc; }
    if (!target_host) { return fail(LUMA_E_ARG, "luma_hip_render", "null target"); }
    DeviceScope scope(ctx->device);
    if (!scope.ok()) { return fail(LUMA_E_HIP, "luma_hip_render", "hipSetDevice failed"); }

    // (a page-locked target is written by the kernel itself: kernel_writable)
    uint32_t* const mirror = kernel_writable(target_host, ctx->band_pixels() * sizeof(uint32_t));
    const auto t0 = std::chrono::steady_clock::now();
    if (int rc = enqueue_render(ctx, params, ctx->d_target, ctx->stream, stats != nullptr, mirror)) { return rc; }
    LUMA_HIP_TRY(hipStreamSynchronize(ctx->stream));
    const auto t1 = std::chrono::steady_clock::now();
    if (!mirror) {
        LUMA_HIP_TRY(hipMemcpy(target_host, ctx->d_target, ctx->band_pixels() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    }
    const auto t2 = std::chrono::steady_clock::now();
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        if (int rc = collect_stats(ctx, ctx->stream, stats)) { return rc; }
        stats->gather_ms = std::chrono::duration<double, std::milli>(t2 - t1).count();
        stats->total_ms = std::chrono::duration<double, std::milli>(t2 - t0).count();
    }
    return LUMA_OK;
}

int luma_hip_render_device(luma_ctx* ctx, const luma_params* params, uint32_t* target_dev, void* stream,
                           luma_stats* stats)
{
    if (int rc = check_params(ctx, params)) { return rc; }
    DeviceScope scope(ctx->device);
    if (!scope.ok()) { return fail(LUMA_E_HIP, "luma_hip_render_device", "hipSetDevice failed"); }
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : ctx->stream;
    const auto t0 = std::chrono::steady_clock::now();
    if (int rc = enqueue_render(ctx, params, target_dev ? target_dev : ctx->d_target, s, stats != nullptr)) { return rc; }
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        if (int rc = collect_stats(ctx, s, stats)) { return rc; }
        stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return LUMA_OK;
}

static_assert(luma::kStaticScratchBytes == LUMA_HIP_NOALLOC_SCRATCH_BYTES, "header and kernels disagree on the static scratch size");
static_assert(kCounterAlloc <= luma::kStaticCounterBytes, "counter block does not fit its part of the static scratch");

int luma_hip_render_device_noalloc(luma_ctx* ctx, const luma_params* params, uint32_t* target_dev, void* stream)
{
    const char* who = "luma_hip_render_device_noalloc";
    if (int rc = check_params(ctx, params)) { return rc; }
    if (!ctx->static_scratch) { return fail(LUMA_E_STATE, who, "static scratch address not resolved"); }
    if (params->flags & LUMA_FLAG_GRID) {
        // the grid tables are built (allocated) on first use, and the grid kernels spill registers,
        // for which the HIP runtime itself sets up per-queue memory at dispatch
        return fail(LUMA_E_ARG, who, "LUMA_FLAG_GRID is not available without allocation; use luma_hip_render_device");
    }
    const bool grid = false;
    // the scratch this call needs, against the fixed size of the static array
    // (a fine class whose parking does not fit is left out; frames split over the whole band must fit)
    const luma::SplitPlan plan = split_plan(ctx, params->spp, grid, (params->flags & LUMA_FLAG_ORIGIN_FROM_HIT) != 0,
                                            luma::kStaticScratchBytes - luma::kStaticCounterBytes);
    const unsigned long long parking = plan.wanted_bytes;
    if (plan.refused) {
        char msg[256];
        std::snprintf(msg, sizeof msg,
                      "spp %u x %zu band pixels needs %llu bytes of scratch, the static array holds %zu "
                      "(LUMA_HIP_NOALLOC_SCRATCH_BYTES): render the frames in several calls of fewer spp",
                      params->spp, ctx->band_pixels(), luma::kStaticCounterBytes + parking, luma::kStaticScratchBytes);
        return fail(LUMA_E_ARG, who, msg);
    }
    DeviceScope scope(ctx->device);
    if (!scope.ok()) { return fail(LUMA_E_HIP, who, "hipSetDevice failed"); }
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : ctx->stream;
    {
        // not a call to capture into a HIP graph: the frame numbers and 1/frame_count are baked into the launch
        // arguments, so a replay would add the same frames again
        hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &capturing) == hipSuccess && capturing != hipStreamCaptureStatusNone) {
            return fail(LUMA_E_STATE, who, "stream is capturing: a captured call would replay the same frames; call it outside the capture");
        }
        (void)hipGetLastError();
    }
    if (int rc = order_after_previous(ctx, s)) { return rc; }
    if (int rc = refresh_records(ctx, s)) { return rc; }          // a launch into memory luma_hip_set_scene allocated

    luma::RenderArgs args = make_args(ctx, params, target_dev ? target_dev : ctx->d_target);
    unsigned char* const scratch = ctx->static_scratch;          // the device pointer, never the array's name
    args.counters = reinterpret_cast<unsigned long long*>(scratch);
    args.tile_counter = reinterpret_cast<uint32_t*>(scratch + kCounterBytes);
    luma::apply_plan(args, plan, reinterpret_cast<float*>(scratch + luma::kStaticCounterBytes));
    // every call on this GPU shares the array and finds the previous call's contents in it: the
    // counters (the tile counter must start at zero) are cleared; the parking planes need no
    // clearing, every slot that is read was written earlier in the same launch
    {
        // after the previous use of the array on this GPU, from whatever context or stream it came (concurrent
        // host threads still have to order their calls themselves: the event is re-recorded by each)
        std::lock_guard<std::mutex> guard(g_noalloc_lock);
        LUMA_HIP_TRY(hipStreamWaitEvent(s, ctx->noalloc_event, 0));
        LUMA_HIP_TRY(hipMemsetAsync(scratch, 0, kCounterAlloc, s));
        LUMA_HIP_TRY(luma::launch_render(args, s));
        LUMA_HIP_TRY(hipEventRecord(ctx->noalloc_event, s));
    }
    ctx->frame_index += params->spp;
    ctx->last_stream = s;
    ctx->rendered = true;
    return LUMA_OK;
}

int luma_hip_render_frame(luma_ctx* ctx, const luma_params* params, uint32_t* frame_host, luma_stats* stats)
{
    if (int rc = check_params(ctx, params)) { return rc; }
    if (!frame_host) { return fail(LUMA_E_ARG, "luma_hip_render_frame", "null frame"); }
    DeviceScope scope(ctx->device);
    if (!scope.ok()) { return fail(LUMA_E_HIP, "luma_hip_render_frame", "hipSetDevice failed"); }

    uint32_t* const mirror = kernel_writable(frame_host, static_cast<size_t>(ctx->width) * ctx->height * sizeof(uint32_t));
    const auto t0 = std::chrono::steady_clock::now();
    if (int rc = enqueue_render(ctx, params, ctx->d_target, ctx->stream, stats != nullptr, mirror, true)) { return rc; }
    LUMA_HIP_TRY(hipStreamSynchronize(ctx->stream));
    const auto t1 = std::chrono::steady_clock::now();
    if (!mirror) { if (int rc = copy_band_to_frame(ctx, frame_host)) { return rc; } }
    const auto t2 = std::chrono::steady_clock::now();
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        if (int rc = collect_stats(ctx, ctx->stream, stats)) { return rc; }
        stats->gather_ms = std::chrono::duration<double, std::milli>(t2 - t1).count();
        stats->total_ms = std::chrono::duration<double, std::milli>(t2 - t0).count();
    }
    return LUMA_OK;
}

int luma_hip_device_count(int* count)
{
    if (!count) { return fail(LUMA_E_ARG, "luma_hip_device_count", "null argument"); }
    *count = 0;
    LUMA_HIP_TRY(hipGetDeviceCount(count));
    return LUMA_OK;
}

int luma_hip_group_create(luma_group** out, int n, const int* device_ids)
{
    if (!out) { return fail(LUMA_E_ARG, "luma_hip_group_create", "null out pointer"); }
    *out = nullptr;
    if (n < 1 || n > 64) { return fail(LUMA_E_ARG, "luma_hip_group_create", "need 1 <= n <= 64 devices"); }
    luma_group* g = new (std::nothrow) luma_group();
    if (!g) { return fail(LUMA_E_NOMEM, "luma_hip_group_create", "host allocation failed"); }
    try { g->ctx.reserve(static_cast<size_t>(n)); } catch (...) {
        delete g;
        return fail(LUMA_E_NOMEM, "luma_hip_group_create", "host allocation failed");
    }
    for (int i = 0; i < n; i++) {
        luma_ctx* c = nullptr;
        const int dev = device_ids ? device_ids[i] : i;
        if (dev < 0) { luma_hip_group_destroy(g); return fail(LUMA_E_ARG, "luma_hip_group_create", "negative device id"); }
        if (int rc = luma_hip_create(&c, dev)) { luma_hip_group_destroy(g); return rc; }   // message already set
        g->ctx.push_back(c);
    }
    *out = g;
    return LUMA_OK;
}

namespace {

// Communicator ranks, their streams and the events (also after a first use that failed half way).
void destroy_gather_comms(luma_group* g)
{
    GroupGather& G = g->gather;
    const luma::RcclApi* rccl = luma::rccl_api(nullptr);
    for (size_t r = 0; r < G.devices.size(); r++) {
        DeviceScope scope(G.devices[r]);
        if (!scope.ok()) { continue; }
        if (r < G.streams.size() && G.streams[r]) { (void)hipStreamSynchronize(G.streams[r]); }
        if (rccl && r < G.comms.size() && G.comms[r]) { (void)rccl->CommDestroy(G.comms[r]); }
        if (r < G.streams.size() && G.streams[r]) { (void)hipStreamDestroy(G.streams[r]); }
    }
    for (size_t i = 0; i < G.rendered.size(); i++) {
        if (!G.rendered[i]) { continue; }
        DeviceScope scope(g->ctx[i]->device);
        if (scope.ok()) { (void)hipEventDestroy(G.rendered[i]); }
    }
    if (!G.devices.empty()) {
        DeviceScope scope(G.devices[0]);
        if (scope.ok()) {
            if (G.ev_begin) { (void)hipEventDestroy(G.ev_begin); }
            if (G.ev_end) { (void)hipEventDestroy(G.ev_end); }
        }
    }
    G.devices.clear(); G.comms.clear(); G.streams.clear(); G.rendered.clear();
    G.ev_begin = G.ev_end = nullptr;
    G.ready = false;
}

void destroy_gather(luma_group* g)
{
    destroy_gather_comms(g);
    GroupGather& G = g->gather;
    if (!g->ctx.empty() && (G.staging || G.frame)) {
        DeviceScope scope(g->ctx[0]->device);
        if (scope.ok()) { free_dev(G.staging); free_dev(G.frame); }
    }
    G.staging_bytes = G.frame_bytes = 0;
}

}  // namespace

void luma_hip_group_destroy(luma_group* g)
{
    if (!g) { return; }
    destroy_gather(g);
    for (luma_ctx* c : g->ctx) { luma_hip_destroy(c); }
    delete g;
}

int luma_hip_group_size(luma_group* g, int* n)
{
    if (!g || !n) { return fail(LUMA_E_ARG, "luma_hip_group_size", "null argument"); }
    *n = static_cast<int>(g->ctx.size());
    return LUMA_OK;
}

int luma_hip_group_context(luma_group* g, int i, luma_ctx** ctx)
{
    if (!g || !ctx) { return fail(LUMA_E_ARG, "luma_hip_group_context", "null argument"); }
    if (i < 0 || static_cast<size_t>(i) >= g->ctx.size()) { return fail(LUMA_E_ARG, "luma_hip_group_context", "index out of range"); }
    *ctx = g->ctx[static_cast<size_t>(i)];
    return LUMA_OK;
}

int luma_hip_group_set_scene(luma_group* g, const luma_sphere* spheres, uint32_t n)
{
    if (!g) { return fail(LUMA_E_ARG, "luma_hip_group_set_scene", "null group"); }
    for (luma_ctx* c : g->ctx) { if (int rc = luma_hip_set_scene(c, spheres, n)) { return rc; } }
    return LUMA_OK;
}

int luma_hip_group_set_camera(luma_group* g, const luma_camera* cam)
{
    if (!g) { return fail(LUMA_E_ARG, "luma_hip_group_set_camera", "null group"); }
    for (luma_ctx* c : g->ctx) { if (int rc = luma_hip_set_camera(c, cam)) { return rc; } }
    return LUMA_OK;
}

int luma_hip_group_set_frame(luma_group* g, uint32_t width, uint32_t height)
{
    const char* who = "luma_hip_group_set_frame";
    if (!g) { return fail(LUMA_E_ARG, who, "null group"); }
    const uint32_t n = static_cast<uint32_t>(g->ctx.size());
    if (height < n) { return fail(LUMA_E_ARG, who, "fewer rows than devices: a band would be empty"); }
    g->have_frame = false;
    int ids[luma::kMaxGroupMembers];
    for (uint32_t i = 0; i < n; i++) { ids[i] = g->ctx[i]->device; }
    if (!luma::plan_gather(static_cast<int>(n), ids, width, height, &g->plan)) { return fail(LUMA_E_ARG, who, "zero width or height"); }
    const bool striped = g->plan.striped != 0;
    for (uint32_t i = 0; i < n; i++) {
        const luma::GatherMember& m = g->plan.member[i];
        const int rc = striped ? luma_hip_set_frame_striped(g->ctx[i], width, height, g->plan.stripe_rows, i, n)
                               : luma_hip_set_frame(g->ctx[i], width, height, m.row_begin, m.row_begin + m.rows);
        if (rc) { return rc; }
    }
    g->have_frame = true;
    g->striped = striped;
    return LUMA_OK;
}

int luma_hip_group_reserve(luma_group* g, uint32_t max_spp)
{
    if (!g) { return fail(LUMA_E_ARG, "luma_hip_group_reserve", "null group"); }
    for (luma_ctx* c : g->ctx) { if (int rc = luma_hip_reserve(c, max_spp)) { return rc; } }
    return LUMA_OK;
}

int luma_hip_group_reset(luma_group* g)
{
    if (!g) { return fail(LUMA_E_ARG, "luma_hip_group_reset", "null group"); }
    for (luma_ctx* c : g->ctx) { if (int rc = luma_hip_reset(c)) { return rc; } }
    return LUMA_OK;
}

namespace {

// After a failure in the middle of a group call: wait for every member stream, so that no copy
// into the caller's frame (or kernel writing a band) is still in flight when the error is
// returned and the caller frees or reuses its buffers.  Keeps the first error's message.
int drain_group_after_error(luma_group* g, int rc)
{
    char keep[sizeof g_last_error];
    std::memcpy(keep, g_last_error, sizeof keep);
    for (luma_ctx* c : g->ctx) {
        DeviceScope scope(c->device);
        if (scope.ok()) { (void)hipStreamSynchronize(c->stream); }
    }
    std::memcpy(g_last_error, keep, sizeof keep);
    return rc;
}

}  // namespace

int luma_hip_group_render(luma_group* g, const luma_params* params, uint32_t* frame_host, luma_stats* stats)
{
    const char* who = "luma_hip_group_render";
    if (!g) { return fail(LUMA_E_ARG, who, "null group"); }
    if (!frame_host) { return fail(LUMA_E_ARG, who, "null frame"); }
    if (!g->have_frame) { return fail(LUMA_E_STATE, who, "luma_hip_group_set_frame not called"); }
    for (luma_ctx* c : g->ctx) { if (int rc = check_params(c, params)) { return rc; } }   // nothing launched yet

    const auto t0 = std::chrono::steady_clock::now();
    // the body; any failure after the first launch goes through drain_group_after_error
    auto run = [&]() -> int {
        // Into a frame that is page-locked from its first to its last byte every device's kernel writes its own
        // pixels (kernel_writable), all links at once and no copy afterwards; any other frame gets one blocking
        // copy per device.
        const size_t frame_bytes = static_cast<size_t>(g->ctx[0]->width) * g->ctx[0]->height * sizeof(uint32_t);
        std::vector<uint32_t*> mirror(g->ctx.size(), nullptr);
        bool direct = true;
        for (size_t i = 0; i < g->ctx.size() && direct; i++) {
            DeviceScope scope(g->ctx[i]->device);
            if (!scope.ok()) { return fail(LUMA_E_HIP, who, "hipSetDevice failed"); }
            mirror[i] = kernel_writable(frame_host, frame_bytes);
            direct = mirror[i] != nullptr;
        }
        for (size_t i = 0; i < g->ctx.size(); i++) {               // every device starts its band ...
            luma_ctx* c = g->ctx[i];
            DeviceScope scope(c->device);
            if (!scope.ok()) { return fail(LUMA_E_HIP, who, "hipSetDevice failed"); }
            if (int rc = enqueue_render(c, params, c->d_target, c->stream, stats != nullptr, direct ? mirror[i] : nullptr, true)) { return rc; }
        }
        for (luma_ctx* c : g->ctx) {                               // ... and is waited for, or hands its band over
            DeviceScope scope(c->device);
            if (!scope.ok()) { return fail(LUMA_E_HIP, who, "hipSetDevice failed"); }
            LUMA_HIP_TRY(hipStreamSynchronize(c->stream));
            if (!direct) { if (int rc = copy_band_to_frame(c, frame_host)) { return rc; } }
        }
        return LUMA_OK;
    };
    if (int rc = run()) { return drain_group_after_error(g, rc); }
    const auto t1 = std::chrono::steady_clock::now();
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        for (luma_ctx* c : g->ctx) {
            DeviceScope scope(c->device);
            if (!scope.ok()) { return fail(LUMA_E_HIP, who, "hipSetDevice failed"); }
            luma_stats one{};
            if (int rc = collect_stats(c, c->stream, &one)) { return rc; }
            stats->kernel_ms = one.kernel_ms > stats->kernel_ms ? one.kernel_ms : stats->kernel_ms;
            stats->lanes_per_pixel = one.lanes_per_pixel > stats->lanes_per_pixel ? one.lanes_per_pixel : stats->lanes_per_pixel;
            stats->split_refused |= one.split_refused;
            stats->rays += one.rays;
            stats->sphere_tests += one.sphere_tests;
        }
        // kernels and copies of different devices overlap: the copy share is what the slowest kernel leaves
        const double all_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
        stats->gather_ms = all_ms > stats->kernel_ms ? all_ms - stats->kernel_ms : 0.0;
        stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return LUMA_OK;
}

namespace {

#define LUMA_RCCL_TRY(api, expr)                                                   \
    do {                                                                           \
        ncclResult_t r_ = (expr);                                                  \
        if (r_ != ncclSuccess) { return fail(LUMA_E_RCCL, #expr, (api)->GetErrorString(r_)); } \
    } while (0)

// First use: one communicator over the distinct devices of the group (ncclCommInitAll, SURVEY 8e:
// one process, N devices), a stream per rank, an event per member.
int ensure_gather(luma_group* g)
{
    GroupGather& G = g->gather;
    if (G.ready) { return LUMA_OK; }
    const char* why = "";
    const luma::RcclApi* rccl = luma::rccl_api(&why);
    if (!rccl) { return fail(LUMA_E_RCCL, "RCCL is not available", why); }
    destroy_gather_comms(g);                                        // leftovers of a first use that failed
    if (!g->have_frame) { return fail(LUMA_E_STATE, "luma_hip_group_render_device", "luma_hip_group_set_frame not called"); }
    try {
        G.devices.assign(g->plan.rank_device, g->plan.rank_device + g->plan.n_ranks);
        G.comms.assign(G.devices.size(), nullptr);
        G.streams.assign(G.devices.size(), nullptr);
        G.rendered.assign(g->ctx.size(), nullptr);
    } catch (...) {
        return fail(LUMA_E_NOMEM, "luma_hip_group_render_device", "host allocation failed");
    }
    LUMA_RCCL_TRY(rccl, rccl->CommInitAll(G.comms.data(), static_cast<int>(G.devices.size()), G.devices.data()));
    for (size_t r = 0; r < G.devices.size(); r++) {
        DeviceScope scope(G.devices[r]);
        if (!scope.ok()) { return fail(LUMA_E_HIP, "luma_hip_group_render_device", "hipSetDevice failed"); }
        LUMA_HIP_TRY(hipStreamCreateWithFlags(&G.streams[r], hipStreamNonBlocking));
        if (r == 0) {
            LUMA_HIP_TRY(hipEventCreate(&G.ev_begin));
            LUMA_HIP_TRY(hipEventCreate(&G.ev_end));
        }
    }
    for (size_t i = 0; i < g->ctx.size(); i++) {
        DeviceScope scope(g->ctx[i]->device);
        if (!scope.ok()) { return fail(LUMA_E_HIP, "luma_hip_group_render_device", "hipSetDevice failed"); }
        LUMA_HIP_TRY(hipEventCreateWithFlags(&G.rendered[i], hipEventDisableTiming));
    }
    G.ready = true;
    return LUMA_OK;
}

// Root-device buffer of at least `bytes`, grown on demand (set-up, not per frame).
int ensure_root_buffer(luma_group* g, uint32_t*& buf, size_t& have, size_t bytes)
{
    if (have >= bytes) { return LUMA_OK; }
    DeviceScope scope(g->ctx[0]->device);
    if (!scope.ok()) { return fail(LUMA_E_HIP, "luma_hip_group_render_device", "hipSetDevice failed"); }
    LUMA_HIP_TRY(hipDeviceSynchronize());
    free_dev(buf);
    have = 0;
    LUMA_HIP_TRY(hipMalloc(&buf, bytes));
    have = bytes;
    return LUMA_OK;
}

}  // namespace

int luma_hip_group_frame_device(luma_group* g, uint32_t** frame_dev)
{
    const char* who = "luma_hip_group_frame_device";
    if (!g || !frame_dev) { return fail(LUMA_E_ARG, who, "null argument"); }
    if (!g->have_frame) { return fail(LUMA_E_STATE, who, "luma_hip_group_set_frame not called"); }
    const size_t bytes = static_cast<size_t>(g->ctx[0]->width) * g->ctx[0]->height * sizeof(uint32_t);
    if (int rc = ensure_root_buffer(g, g->gather.frame, g->gather.frame_bytes, bytes)) { return rc; }
    *frame_dev = g->gather.frame;
    return LUMA_OK;
}

int luma_hip_group_read_frame(luma_group* g, uint32_t* frame_host)
{
    const char* who = "luma_hip_group_read_frame";
    if (!g || !frame_host) { return fail(LUMA_E_ARG, who, "null argument"); }
    if (!g->have_frame || !g->gather.frame) { return fail(LUMA_E_STATE, who, "no frame gathered yet (luma_hip_group_render_device with frame_dev == NULL)"); }
    DeviceScope scope(g->ctx[0]->device);
    if (!scope.ok()) { return fail(LUMA_E_HIP, who, "hipSetDevice failed"); }
    const size_t bytes = static_cast<size_t>(g->ctx[0]->width) * g->ctx[0]->height * sizeof(uint32_t);
    LUMA_HIP_TRY(hipMemcpy(frame_host, g->gather.frame, bytes, hipMemcpyDeviceToHost));
    return LUMA_OK;
}

int luma_hip_group_render_device(luma_group* g, const luma_params* params, uint32_t* frame_dev, luma_stats* stats)
{
    const char* who = "luma_hip_group_render_device";
    if (!g) { return fail(LUMA_E_ARG, who, "null group"); }
    if (!g->have_frame) { return fail(LUMA_E_STATE, who, "luma_hip_group_set_frame not called"); }
    for (luma_ctx* c : g->ctx) { if (int rc = check_params(c, params)) { return rc; } }   // nothing launched yet
    luma_ctx* const root = g->ctx[0];
    const size_t width = root->width, frame_bytes = width * root->height * sizeof(uint32_t);
    if (frame_dev) {
        hipPointerAttribute_t attr{};
        if (hipPointerGetAttributes(&attr, frame_dev) != hipSuccess || attr.type != hipMemoryTypeDevice || attr.device != root->device) {
            (void)hipGetLastError();
            return fail(LUMA_E_ARG, who, "frame_dev must be device memory of the root (the first member's device)");
        }
    } else {
        if (int rc = luma_hip_group_frame_device(g, &frame_dev)) { return rc; }
    }
    if (int rc = ensure_gather(g)) { return rc; }
    GroupGather& G = g->gather;
    if (g->striped) { if (int rc = ensure_root_buffer(g, G.staging, G.staging_bytes, frame_bytes)) { return rc; } }
    const luma::RcclApi* rccl = luma::rccl_api(nullptr);

    const auto t0 = std::chrono::steady_clock::now();
    auto run = [&]() -> int {
        // every device renders its band on its context's stream and marks the end with an event ...
        for (size_t i = 0; i < g->ctx.size(); i++) {
            luma_ctx* c = g->ctx[i];
            DeviceScope scope(c->device);
            if (!scope.ok()) { return fail(LUMA_E_HIP, who, "hipSetDevice failed"); }
            if (int rc = enqueue_render(c, params, c->d_target, c->stream, stats != nullptr)) { return rc; }
            LUMA_HIP_TRY(hipEventRecord(G.rendered[i], c->stream));
        }
        // ... the exchange stream of each device waits for the bands rendered on that device, the root's for
        // every band: gather_ms then starts when the last band is done and times the exchange alone, not the
        // skew between the devices' renders ...
        for (size_t i = 0; i < g->ctx.size(); i++) {
            const int r = g->plan.member[i].comm_rank;
            DeviceScope scope(G.devices[r]);
            if (!scope.ok()) { return fail(LUMA_E_HIP, who, "hipSetDevice failed"); }
            LUMA_HIP_TRY(hipStreamWaitEvent(G.streams[r], G.rendered[i], 0));
        }
        {
            DeviceScope scope(G.devices[0]);
            if (!scope.ok()) { return fail(LUMA_E_HIP, who, "hipSetDevice failed"); }
            for (size_t i = 0; i < g->ctx.size(); i++) {
                if (g->plan.member[i].comm_rank != 0) { LUMA_HIP_TRY(hipStreamWaitEvent(G.streams[0], G.rendered[i], 0)); }
            }
            LUMA_HIP_TRY(hipEventRecord(G.ev_begin, G.streams[0]));
        }
        // ... and ONE grouped exchange moves every band into the root: each member sends its band
        // from its own device (its own xGMI link into the root), the root posts the matching
        // receives, in member order (sends and receives between one pair of ranks match in order).
        // Contiguous bands land at their rows of the frame; stripe bands land back to back in the
        // staging buffer and one kernel puts the stripes in row order.  A member on the root's own
        // device goes the same way (a send to self inside the group is a local copy).
        LUMA_RCCL_TRY(rccl, rccl->GroupStart());
        int rc_group = LUMA_OK;
        for (size_t i = 0; i < g->ctx.size() && rc_group == LUMA_OK; i++) {
            luma_ctx* c = g->ctx[i];
            const luma::GatherMember& m = g->plan.member[i];              // the plan luma_gather_plan_describe shows
            const int r = m.comm_rank;
            const size_t count = static_cast<size_t>(m.pixels);
            uint32_t* dst = (g->striped ? G.staging : frame_dev) + m.recv_offset;
            ncclResult_t e;
            {
                DeviceScope scope(c->device);
                if (!scope.ok()) { rc_group = fail(LUMA_E_HIP, who, "hipSetDevice failed"); break; }
                e = rccl->Send(c->d_target, count, ncclUint32, 0, G.comms[r], G.streams[r]);
            }
            if (e == ncclSuccess) {
                DeviceScope scope(G.devices[0]);
                if (!scope.ok()) { rc_group = fail(LUMA_E_HIP, who, "hipSetDevice failed"); break; }
                e = rccl->Recv(dst, count, ncclUint32, r, G.comms[0], G.streams[0]);
            }
            if (e != ncclSuccess) { rc_group = fail(LUMA_E_RCCL, "ncclSend/ncclRecv", rccl->GetErrorString(e)); }
        }
        const ncclResult_t e_end = rccl->GroupEnd();                  // always closes the group
        if (rc_group != LUMA_OK) { return rc_group; }
        if (e_end != ncclSuccess) { return fail(LUMA_E_RCCL, "ncclGroupEnd", rccl->GetErrorString(e_end)); }
        {
            DeviceScope scope(G.devices[0]);
            if (!scope.ok()) { return fail(LUMA_E_HIP, who, "hipSetDevice failed"); }
            if (g->striped) {
                LUMA_HIP_TRY(luma::launch_deinterleave(G.staging, frame_dev, root->width, root->height, root->stripe,
                                                       static_cast<uint32_t>(g->ctx.size()), G.streams[0]));
            }
            LUMA_HIP_TRY(hipEventRecord(G.ev_end, G.streams[0]));
        }
        // synchronous: the frame is complete on return, and no band buffer is still being sent
        for (size_t r = 0; r < G.devices.size(); r++) {
            DeviceScope scope(G.devices[r]);
            if (!scope.ok()) { return fail(LUMA_E_HIP, who, "hipSetDevice failed"); }
            LUMA_HIP_TRY(hipStreamSynchronize(G.streams[r]));
        }
        return LUMA_OK;
    };
    if (int rc = run()) {
        for (size_t r = 0; r < G.streams.size(); r++) {               // nothing may still write the caller's frame
            DeviceScope scope(G.devices[r]);
            if (scope.ok() && G.streams[r]) { (void)hipStreamSynchronize(G.streams[r]); }
        }
        return drain_group_after_error(g, rc);
    }
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        for (luma_ctx* c : g->ctx) {
            DeviceScope scope(c->device);
            if (!scope.ok()) { return fail(LUMA_E_HIP, who, "hipSetDevice failed"); }
            luma_stats one{};
            if (int rc = collect_stats(c, c->stream, &one)) { return rc; }
            stats->kernel_ms = one.kernel_ms > stats->kernel_ms ? one.kernel_ms : stats->kernel_ms;
            stats->lanes_per_pixel = one.lanes_per_pixel > stats->lanes_per_pixel ? one.lanes_per_pixel : stats->lanes_per_pixel;
            stats->split_refused |= one.split_refused;
            stats->rays += one.rays;
            stats->sphere_tests += one.sphere_tests;
        }
        DeviceScope scope(G.devices[0]);
        if (!scope.ok()) { return fail(LUMA_E_HIP, who, "hipSetDevice failed"); }
        float ms = 0.f;
        LUMA_HIP_TRY(hipEventElapsedTime(&ms, G.ev_begin, G.ev_end));
        stats->gather_ms = ms;                                        // exchange (+ de-interleave) on the root
        stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return LUMA_OK;
}

int luma_hip_group_read_accum(luma_group* g, float* rgb_host)
{
    const char* who = "luma_hip_group_read_accum";
    if (!g || !rgb_host) { return fail(LUMA_E_ARG, who, "null argument"); }
    if (!g->have_frame) { return fail(LUMA_E_STATE, who, "luma_hip_group_set_frame not called"); }
    std::vector<float> band;
    for (luma_ctx* c : g->ctx) {
        try { band.resize(3 * c->band_pixels()); } catch (...) { return fail(LUMA_E_NOMEM, who, "host allocation failed"); }
        if (int rc = luma_hip_read_accum(c, band.data())) { return rc; }
        const size_t row_floats = 3 * static_cast<size_t>(c->width);
        for (uint32_t yl = 0; yl < c->rows; yl++) {             // same mapping as global_row() in the kernels
            const size_t y = c->row_begin + static_cast<size_t>(yl / c->stripe) * c->pitch + (yl % c->stripe);
            std::memcpy(rgb_host + y * row_floats, band.data() + yl * row_floats, row_floats * sizeof(float));
        }
    }
    return LUMA_OK;
}

int luma_hip_host_alloc(void** out, size_t bytes)
{
    if (!out || bytes == 0) { return fail(LUMA_E_ARG, "luma_hip_host_alloc", "null pointer or zero size"); }
    *out = nullptr;
    LUMA_HIP_TRY(hipHostMalloc(out, bytes, hipHostMallocPortable));
    return LUMA_OK;
}

void luma_hip_host_free(void* p)
{
    if (p) { (void)hipHostFree(p); }
}

int luma_hip_target_device(luma_ctx* ctx, uint32_t** target_dev)
{
    if (!ctx || !target_dev) { return fail(LUMA_E_ARG, "luma_hip_target_device", "null argument"); }
    if (!ctx->have_frame) { return fail(LUMA_E_STATE, "luma_hip_target_device", "luma_hip_set_frame not called"); }
    *target_dev = ctx->d_target;
    return LUMA_OK;
}

int luma_hip_read_target(luma_ctx* ctx, const uint32_t* target_dev, uint32_t* target_host)
{
    const char* who = "luma_hip_read_target";
    if (!ctx || !target_host) { return fail(LUMA_E_ARG, who, "null argument"); }
    if (!ctx->have_frame) { return fail(LUMA_E_STATE, who, "luma_hip_set_frame not called"); }
    DeviceScope scope(ctx->device);
    if (!scope.ok()) { return fail(LUMA_E_HIP, who, "hipSetDevice failed"); }
    LUMA_HIP_TRY(hipDeviceSynchronize());   // the latest render may have run on a caller's stream
    LUMA_HIP_TRY(hipMemcpy(target_host, target_dev ? target_dev : ctx->d_target, ctx->band_pixels() * sizeof(uint32_t),
                           hipMemcpyDeviceToHost));
    return LUMA_OK;
}

int luma_hip_read_accum(luma_ctx* ctx, float* rgb_host)
{
    if (!ctx || !rgb_host) { return fail(LUMA_E_ARG, "luma_hip_read_accum", "null argument"); }
    if (!ctx->have_frame) { return fail(LUMA_E_STATE, "luma_hip_read_accum", "luma_hip_set_frame not called"); }
    DeviceScope scope(ctx->device);
    if (!scope.ok()) { return fail(LUMA_E_HIP, "luma_hip_read_accum", "hipSetDevice failed"); }
    if (int rc = order_after_previous(ctx, ctx->stream)) { return rc; }
    LUMA_HIP_TRY(hipStreamSynchronize(ctx->stream));
    const size_t px = ctx->band_pixels();
    std::vector<float> planes;
    try { planes.resize(3 * px); } catch (...) { return fail(LUMA_E_NOMEM, "luma_hip_read_accum", "host allocation failed"); }
    LUMA_HIP_TRY(hipMemcpy(planes.data(), ctx->d_acc, 3 * px * sizeof(float), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < px; i++) {   // planar on the device, AoS rgb at the boundary (renderer.h:56)
        rgb_host[3 * i + 0] = planes[i];
        rgb_host[3 * i + 1] = planes[px + i];
        rgb_host[3 * i + 2] = planes[2 * px + i];
    }
    return LUMA_OK;
}

int luma_hip_frame_index(luma_ctx* ctx, uint32_t* frame_index)
{
    if (!ctx || !frame_index) { return fail(LUMA_E_ARG, "luma_hip_frame_index", "null argument"); }
    *frame_index = ctx->frame_index;
    return LUMA_OK;
}

int luma_hip_generate_rays(luma_ctx* ctx, float* rays_host)
{
    if (!ctx || !rays_host) { return fail(LUMA_E_ARG, "luma_hip_generate_rays", "null argument"); }
    if (!ctx->have_camera || !ctx->have_frame) {
        return fail(LUMA_E_STATE, "luma_hip_generate_rays", "camera and frame must be set");
    }
    DeviceScope scope(ctx->device);
    if (!scope.ok()) { return fail(LUMA_E_HIP, "luma_hip_generate_rays", "hipSetDevice failed"); }
    const size_t bytes = ctx->band_pixels() * 3 * sizeof(float);
    float* d_rays = nullptr;
    LUMA_HIP_TRY(hipMalloc(&d_rays, bytes));
    const luma::RenderArgs args = make_args(ctx, nullptr, nullptr);
    hipError_t e = luma::launch_generate_rays(args, d_rays, ctx->stream);
    if (e == hipSuccess) { e = hipStreamSynchronize(ctx->stream); }
    if (e == hipSuccess) { e = hipMemcpy(rays_host, d_rays, bytes, hipMemcpyDeviceToHost); }
    (void)hipFree(d_rays);
    if (e != hipSuccess) { return fail(LUMA_E_HIP, "luma_hip_generate_rays", hipGetErrorString(e)); }
    return LUMA_OK;
}

int luma_camera_default(float fov_deg, float nearclip, float farclip, uint32_t width, uint32_t height,
                        luma_camera* out)
{
    if (!out || width == 0 || height == 0) { return fail(LUMA_E_ARG, "luma_camera_default", "bad argument"); }
    luma::camera_default(fov_deg, nearclip, farclip, width, height, out);
    return LUMA_OK;
}

int luma_camera_from_pose(float fov_deg, float nearclip, float farclip, uint32_t width, uint32_t height,
                          const float pos[3], float pitch, float yaw, luma_camera* out)
{
    if (!out || !pos || width == 0 || height == 0) { return fail(LUMA_E_ARG, "luma_camera_from_pose", "bad argument"); }
    luma::camera_from_pose(fov_deg, nearclip, farclip, width, height, pos, pitch, yaw, out);
    return LUMA_OK;
}

int luma_grid_describe(const luma_sphere* spheres, uint32_t n, luma_grid_info* info,
                       uint32_t* cell_start, uint32_t* items, uint32_t* large)
{
    if (!info || (n > 0 && !spheres)) { return fail(LUMA_E_ARG, "luma_grid_describe", "null argument"); }
    luma::HostGrid g;
    try { luma::build_grid(spheres, n, g); } catch (...) {
        return fail(LUMA_E_NOMEM, "luma_grid_describe", "host allocation failed");
    }
    for (int i = 0; i < 3; i++) { info->dims[i] = g.dims[i]; info->org[i] = g.org[i]; }
    info->cell = g.h;
    info->n_items = static_cast<uint32_t>(g.items.size());
    info->n_large = static_cast<uint32_t>(g.large.size());
    if (cell_start) { std::copy(g.cell_start.begin(), g.cell_start.end(), cell_start); }
    if (items) { std::copy(g.items.begin(), g.items.end(), items); }
    if (large) { std::copy(g.large.begin(), g.large.end(), large); }
    return LUMA_OK;
}

int luma_split_plan_describe(uint32_t width, uint32_t rows, uint32_t spp, uint32_t n_spheres, uint32_t flags,
                             uint64_t park_budget, luma_split_plan* plan)
{
    if (!plan) { return fail(LUMA_E_ARG, "luma_split_plan_describe", "null argument"); }
    if (width == 0 || rows == 0 || spp == 0) { return fail(LUMA_E_ARG, "luma_split_plan_describe", "width, rows and spp must be >= 1"); }
    const bool grid = (flags & LUMA_FLAG_GRID) != 0 && n_spheres > 0;
    const luma::SplitPlan p = luma::plan_split(width, rows, spp, n_spheres, grid, (flags & LUMA_FLAG_ORIGIN_FROM_HIT) != 0,
                                               static_cast<size_t>(park_budget));
    plan->lanes = p.split;
    plan->fine_lanes = p.fine;
    plan->fine_row0 = p.fine_row0;
    plan->park_row0 = p.park_row0;
    plan->park_bytes = p.park_bytes;
    plan->park_depth = p.park_depth;
    plan->refused = p.refused ? 1u : 0u;
    return LUMA_OK;
}

int luma_gather_plan_describe(int n, const int* device_ids, uint32_t width, uint32_t height, luma_gather_plan* plan,
                              luma_gather_member* members, int* rank_devices)
{
    if (!plan) { return fail(LUMA_E_ARG, "luma_gather_plan_describe", "null argument"); }
    luma::GatherPlan P;
    if (!luma::plan_gather(n, device_ids, width, height, &P)) {
        return fail(LUMA_E_ARG, "luma_gather_plan_describe", "need 1 <= n <= 64 members, device ids >= 0, width > 0, height >= n");
    }
    plan->n_members = static_cast<uint32_t>(P.n_members);
    plan->n_ranks = static_cast<uint32_t>(P.n_ranks);
    plan->striped = P.striped;
    plan->stripe_rows = P.stripe_rows;
    for (int i = 0; members && i < P.n_members; i++) {
        const luma::GatherMember& m = P.member[i];
        members[i].device = m.device;
        members[i].comm_rank = m.comm_rank;
        members[i].row_begin = m.row_begin;
        members[i].rows = m.rows;
        members[i].pixels = m.pixels;
        members[i].recv_offset = m.recv_offset;
    }
    for (int r = 0; rank_devices && r < P.n_ranks; r++) { rank_devices[r] = P.rank_device[r]; }
    return LUMA_OK;
}

int luma_view_grid_describe(const luma_sphere* spheres, uint32_t n, const float origin[3], uint32_t* res,
                            uint32_t* n_items, uint32_t* n_large, uint32_t* cell_start, uint32_t* items, uint32_t* large)
{
    if (!origin || !res || !n_items || !n_large || (n > 0 && !spheres)) { return fail(LUMA_E_ARG, "luma_view_grid_describe", "null argument"); }
    luma::HostViewGrid g;
    try { luma::build_view_grid(spheres, n, origin, g); } catch (...) {
        return fail(LUMA_E_NOMEM, "luma_view_grid_describe", "host allocation failed");
    }
    *res = g.res;
    *n_items = static_cast<uint32_t>(g.items.size());
    *n_large = static_cast<uint32_t>(g.large.size());
    if (cell_start) { std::copy(g.cell_start.begin(), g.cell_start.end(), cell_start); }
    if (items) { std::copy(g.items.begin(), g.items.end(), items); }
    if (large) { std::copy(g.large.begin(), g.large.end(), large); }
    return LUMA_OK;
}

int luma_scene_default(luma_sphere* out)
{
    if (!out) { return fail(LUMA_E_ARG, "luma_scene_default", "null argument"); }
    luma::scene_default(out);
    return LUMA_OK;
}

int luma_scene_synthetic(luma_sphere* out, uint32_t n, uint64_t seed)
{
    if (!out && n) { return fail(LUMA_E_ARG, "luma_scene_synthetic", "null argument"); }
    luma::scene_synthetic(out, n, seed);
    return LUMA_OK;
}

}  // extern "C"
