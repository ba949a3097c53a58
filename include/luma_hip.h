/* luma_hip.h -- C-ABI of the MI355X (gfx950) render-loop backend for connorjlink/raytracer
 * ("Luma").  One shared library, libluma_hip.so, replaces the inner loops of
 * Renderer::render_to (Renderer.cpp:366-471) and Camera::recompute_rays (Camera.cpp:185-216)
 * behind the reference's own Renderer / Camera API.  Plain C types only; no exceptions cross
 * this boundary; every function returns 0 or a negative LUMA_E_* code.
 *
 * The reference has no FFI: its seam is the C++ method
 *     void Renderer::render_to(std::uint32_t* target, olc::PixelGameEngine*)   renderer.h:76
 * plus the global luma::_options (arguments.h:43).  Each entry point below names the
 * reference lines it stands in for.  INTEGRATION.md shows the reference-side glue.
 *
 * Threading: a context is used from one host thread at a time (the reference is single
 * threaded, Renderer.cpp:392).  Two calls on one context must not overlap.
 */
#ifndef LUMA_HIP_H
#define LUMA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LUMA_HIP_ABI_VERSION 3   /* 2: luma_stats gained lanes_per_pixel / split_refused; 3: luma_split_plan gained park_depth,
                                    luma_gather_plan_describe, luma_hip_set_park_limit */

enum {
    LUMA_OK       = 0,
    LUMA_E_ARG    = -1,  /* bad argument (null pointer, zero size, out-of-range value) */
    LUMA_E_HIP    = -2,  /* a HIP runtime call failed; luma_hip_last_error() has the text */
    LUMA_E_NOMEM  = -3,  /* device or host allocation failed */
    LUMA_E_STATE  = -4,  /* call order: scene / camera / frame not set */
    LUMA_E_RCCL   = -5   /* a RCCL call failed (multi-device contexts) */
};

/* luma::RenderMode, arguments.h:9-13 */
enum { LUMA_MODE_RAYTRACE = 0, LUMA_MODE_PATHTRACE = 1 };

/* Behaviour switches.  All zero == the committed reference behaviour. */
enum {
    LUMA_FLAG_NOISE_OFF       = 1u, /* no direction / roughness noise, no RNG (the reference has no such path: Renderer.cpp:302) */
    LUMA_FLAG_ORIGIN_FROM_HIT = 2u, /* bounce rays start at the hit point + .001 normal (Renderer.cpp:221-222) instead of camera.pos (:303) */
    LUMA_FLAG_CLAMP_PACK      = 4u, /* clamp channels to [0,255] before packing (Renderer.cpp:44-55 does not) */
    LUMA_FLAG_NO_SHOW_DEPTH   = 8u, /* skip the in-focus x1.2 (Renderer.cpp:448-454) regardless of camera.show_depth */
    LUMA_FLAG_PACK_RGBA8      = 16u, /* target bytes R,G,B,255 (0xFFBBGGRR) instead of ::RGB's 0x00RRGGBB (Renderer.cpp:44-55);
                                        same truncation and OR-combining, so combine with CLAMP_PACK for display use */
    LUMA_FLAG_GRID            = 32u  /* closest-hit tests only against the spheres a ray can reach instead of all of
                                        Renderer::spheres (Renderer.cpp:242): rays from camera.pos (all primary and bounce
                                        rays, :303) look their direction up in a cube-map grid around that point, rays from
                                        hit points walk a uniform grid over the scene; same tests, same winner, same frame
                                        bit for bit, far fewer tests in large scenes.  luma_stats.sphere_tests then counts
                                        the tests made.  Not the reference's scan: off by default */
};

typedef struct luma_ctx luma_ctx;

/* == luma::Sphere + luma::Material, renderer.h:23-36: 40 bytes, same field order, so a
 * std::vector<Sphere>::data() can be passed as is. */
typedef struct {
    float cx, cy, cz, radius;
    float dr, dg, db;
    float albedo, metallic, roughness;
} luma_sphere;

/* What the loop reads of luma::Camera (camera.h:23-37).  Matrices are the reference's
 * fx::mat4 m[i][j] flattened as [i*4+j]. */
typedef struct {
    float pos[3];
    float proj_inv[16];   /* Camera::projection_inverse */
    float view_inv[16];   /* Camera::view_inverse */
    float focus_depth;    /* Camera::depth */
    uint32_t show_depth;  /* Camera::show_depth */
} luma_camera;

/* luma::Options (arguments.h:35-41) snapshot + per-call additions. */
typedef struct {
    uint32_t width, height;            /* full frame, _options.width/height */
    uint32_t samples, bounces, paths;  /* _options.samples/bounces/paths */
    uint32_t mode;                     /* LUMA_MODE_* */
    uint32_t spp;                      /* frames accumulated by this call (>= 1); one render_to == 1 */
    uint32_t flags;                    /* LUMA_FLAG_* */
    uint64_t seed;                     /* counter-RNG seed */
} luma_params;

typedef struct {
    double   kernel_ms;     /* render kernel(s), HIP events on the context's stream */
    double   gather_ms;     /* device->host (and inter-device) copy of the frame */
    double   total_ms;      /* wall clock of the call */
    /* Work counters cover every render call on the context since the previous call that
     * returned a luma_stats (so a run of stats-less luma_hip_render_device calls can be
     * totalled by the last one; rays traced before a luma_hip_set_scene are priced at the scene
     * they ran against); the times are those of this call alone. */
    uint64_t rays;          /* trace_ray invocations executed (Renderer.cpp:234) */
    uint64_t sphere_tests;  /* ray-sphere tests executed (loop at Renderer.cpp:242) */
    /* How the latest launch used the GPU (no counterpart in the reference): lanes that shared one
     * pixel's frames (1, 2 or 4; small bands of small scenes use several; the last rows
     * of a launch may run more of them, its "fine class", which this number does not show), and 1 in
     * split_refused when it ran one lane per pixel only because the planes in which the extra lanes
     * park their frames (spp * 12 B * band pixels) would have passed 2 GiB. */
    uint32_t lanes_per_pixel;
    uint32_t split_refused;
} luma_stats;

/* ---- lifetime ------------------------------------------------------------------ */

/* Stands in for Renderer::Renderer (Renderer.cpp:163-170): owns the device-side
 * accumulated_data.  device_id < 0 selects the current HIP device. */
int  luma_hip_create(luma_ctx** out, int device_id);
void luma_hip_destroy(luma_ctx* ctx);

/* ---- inputs -------------------------------------------------------------------- */

/* Renderer::spheres (renderer.h:70), in vector order (ties go to the lower index,
 * Renderer.cpp:266).  AoS in, structure-of-arrays on the device. */
int luma_hip_set_scene(luma_ctx* ctx, const luma_sphere* spheres, uint32_t n);

/* camera.pos / projection_inverse / view_inverse / depth / show_depth as read at
 * Renderer.cpp:293-294,413-414,425,448 and Camera.cpp:203,212.  Does not reset the
 * accumulation: call luma_hip_reset() for the camera.moved branch. */
int luma_hip_set_camera(luma_ctx* ctx, const luma_camera* cam);

/* Frame geometry.  The context renders rows [row_begin,row_end) of a width x height frame
 * (a scanline band; the whole frame for a single device).  Allocates and zeroes the band's
 * accumulation planes (Renderer.cpp:166-169) and resets the frame counter. */
int luma_hip_set_frame(luma_ctx* ctx, uint32_t width, uint32_t height,
                       uint32_t row_begin, uint32_t row_end);

/* Same, for a band made of interleaved stripes: the context renders rows y with
 * (y / stripe_rows) % n_ranks == rank, stored compactly in that order (stripe 0 of the rank,
 * then its next stripe, ...).  height must be a multiple of stripe_rows * n_ranks.  Rows near
 * the horizon cost more than sky rows, so interleaving evens the ranks out;
 * the frame is still bit-identical to a one-device render.  No counterpart in the reference. */
int luma_hip_set_frame_striped(luma_ctx* ctx, uint32_t width, uint32_t height, uint32_t stripe_rows,
                               uint32_t rank, uint32_t n_ranks);

/* Set-up for render calls of up to max_spp frames on the current scene and frame: allocates now
 * whatever device scratch such calls use (the planes in which small bands, and the last rows of
 * large ones, park frames rendered by several lanes of one pixel: luma_split_plan.park_depth * 12 B per such
 * pixel, 73 MB for a 1080p frame of 64 spp (its last 64 rows), 172 MB for an eighth of it (every pixel)), so that luma_hip_render* itself never allocates or synchronises the
 * device for it.  Optional: without it the first render call that needs more scratch than the
 * context holds allocates it.  Call again after luma_hip_set_frame* / luma_hip_set_scene.
 * luma_hip_alloc_count reports how many device allocations render calls have made on the context
 * since luma_hip_create (0 after a luma_hip_reserve that covers them; LUMA_FLAG_GRID adds 3 per
 * scene for its tables).  No counterpart in the reference. */
int luma_hip_reserve(luma_ctx* ctx, uint32_t max_spp);
int luma_hip_alloc_count(luma_ctx* ctx, uint64_t* count);
/* The most device memory the context may hold for parked frames (default and largest value: 2 GiB).  A render call
 * whose band would want more with several lanes per pixel runs one lane per pixel instead, with as many rows in its
 * fine class as fit (luma_stats.split_refused; what luma_split_plan_describe reports for the same budget); the frame
 * is the same bit for bit.  For callers that share the GPU's memory with something larger.  Takes effect at the next
 * render or luma_hip_reserve call; memory already held is kept.  No counterpart in the reference. */
int luma_hip_set_park_limit(luma_ctx* ctx, uint64_t bytes);

/* The camera.moved branch, Renderer.cpp:377-390: frame_count = 1, accumulated_data = 0. */
int luma_hip_reset(luma_ctx* ctx);

/* ---- the render loop ----------------------------------------------------------- */

/* Renderer::render_to (Renderer.cpp:366-471) for params->spp successive frames of the
 * context's band.  target_host receives the band's pixels, rows*width
 * uint32 0x00RRGGBB, row-major (Renderer.cpp:461).  Synchronous.  stats may be NULL. */
int luma_hip_render(luma_ctx* ctx, const luma_params* params, uint32_t* target_host,
                    luma_stats* stats);

/* Same, but the band is left in device memory: target_dev is a device pointer to
 * rows*width uint32 (NULL: the context's own buffer, see
 * luma_hip_target_device).  Work is enqueued on `stream` (a hipStream_t; NULL = the
 * context's stream) and the call returns without waiting unless stats is non-NULL.  Calls on
 * one context are ordered: a call on a different stream than the previous one first waits for
 * the device (the accumulation planes are read-modify-write state). */
int luma_hip_render_device(luma_ctx* ctx, const luma_params* params, uint32_t* target_dev,
                           void* stream, luma_stats* stats);

/* luma_hip_render_device for callers that cannot allocate device memory (tight serving loops, harnesses
 * that forbid hipMalloc, code that runs next to a HIP graph capture).  No counterpart in the reference.
 * The call itself is not one to capture: frame numbers and 1/frame_count are launch arguments, a replay would add
 * the same frames again; on a capturing stream it returns LUMA_E_STATE.
 * Same inputs, same frame and same accumulation planes as luma_hip_render_device, bit for bit, but
 * the scratch of its kernels (ray counters, the tile counter of large scenes, the planes in which
 * small bands park frames that several lanes of one pixel rendered) lies in a static device array of
 * LUMA_HIP_NOALLOC_SCRATCH_BYTES that every GPU the library runs on holds from library load to
 * process end.  The call makes no hipMalloc / hipFree and calls no allocating routine; it returns
 * without waiting (no luma_stats: its rays are not added to the context's counters).
 * Limits: a call whose scratch need exceeds the array is rejected with LUMA_E_ARG and a message
 * giving both sizes (need = 32 KiB + slots * 12 B * band pixels, slots = luma_split_plan.park_depth: about (L-1)/L of spp for L lanes per pixel, for bands that run several lanes per pixel
 * (luma_split_plan.lanes > 1: scenes under 320 spheres; under 1.5 M pixels, or under ~380k pixels for calls of 8-128 frames on scenes of
 * up to 16 spheres, where one lane per pixel serves further down), 32 KiB otherwise; larger bands park only their last rows, as many as fit the
 * array): render the frames in several calls of fewer spp, which accumulates to the same bits.  LUMA_FLAG_GRID is rejected (LUMA_E_ARG): its tables are allocated
 * on first use and its kernels spill registers, which makes the HIP runtime set up memory of its
 * own at dispatch; the kernels behind every other flag use no private memory.  All calls
 * on one GPU share the array, from any context: the library orders them with one event per GPU (a call waits for
 * the previous such call on that GPU, whatever its context or stream, at the cost of one stream-wait); calls made
 * from several host threads at once still race and must be ordered by the caller.  luma_hip_set_scene / _set_camera / _set_frame
 * are set-up and do allocate; changing streams between calls on one context synchronises the device. */
#define LUMA_HIP_NOALLOC_SCRATCH_BYTES (64u << 20)
int luma_hip_render_device_noalloc(luma_ctx* ctx, const luma_params* params, uint32_t* target_dev,
                                   void* stream);

/* Device pointer of the context's own band framebuffer (valid until set_frame/destroy). */
int luma_hip_target_device(luma_ctx* ctx, uint32_t** target_dev);

/* luma_hip_render for a context that holds one band of a frame assembled in HOST memory:
 * frame_host is the whole width*height frame (Renderer::render_to's `target`, Renderer.cpp:461)
 * and the call writes only this context's rows into it, at their place in the frame (one copy for
 * a contiguous band, one strided copy for interleaved stripes).  Contexts of different devices may
 * be driven from different host threads into the same frame_host. */
int luma_hip_render_frame(luma_ctx* ctx, const luma_params* params, uint32_t* frame_host,
                          luma_stats* stats);

/* ---- several devices in one process (`--gpus N`, SURVEY.md 8b "internally N devices") ---- */

/* A group is n contexts, one per entry of device_ids (NULL: devices 0..n-1), that render one
 * frame together: the frame is cut into interleaved 15-row stripes when height divides by
 * 15*n, else into contiguous scanline bands; every device renders its part on its own stream at
 * the same time and copies it straight into the caller's host frame over its own PCIe link (the
 * reference's target is host memory, so no device-to-device gather is needed;
 * luma_hip_group_render_device below gathers the bands on one GPU over RCCL instead).  The frame is bit-identical to a
 * one-device render.  A device id may repeat (several bands on one GPU; used by the tests).
 * No counterpart in the reference, which is single-threaded CPU code. */
typedef struct luma_group luma_group;
int  luma_hip_device_count(int* count);
int  luma_hip_group_create(luma_group** out, int n, const int* device_ids);
void luma_hip_group_destroy(luma_group* g);
int  luma_hip_group_size(luma_group* g, int* n);
int  luma_hip_group_context(luma_group* g, int i, luma_ctx** ctx);   /* borrowed: read-back hooks, stats */
int  luma_hip_group_set_scene(luma_group* g, const luma_sphere* spheres, uint32_t n);
int  luma_hip_group_set_camera(luma_group* g, const luma_camera* cam);
int  luma_hip_group_set_frame(luma_group* g, uint32_t width, uint32_t height);
int  luma_hip_group_reset(luma_group* g);
int  luma_hip_group_reserve(luma_group* g, uint32_t max_spp);        /* luma_hip_reserve on every member */
/* Renderer::render_to over the whole frame: frame_host receives width*height pixels.  kernel_ms is
 * the slowest device's, rays / sphere_tests are summed.  After a failure call luma_hip_group_reset
 * (devices may have advanced unevenly). */
int  luma_hip_group_render(luma_group* g, const luma_params* params, uint32_t* frame_host,
                           luma_stats* stats);
/* Renderer::render_to over the whole frame with the frame assembled in DEVICE memory of the root (the
 * first member's device), for consumers that keep it on the GPU (the reference's own
 * target is host memory, Renderer.cpp:461, which luma_hip_group_render serves without any gather).
 * Every device renders its band; then one grouped RCCL exchange over one communicator for the group's
 * devices (ncclCommInitAll at first use) moves the bands: each member ncclSend's its band from its own
 * device, the root ncclRecv's them, contiguous bands straight into their rows of the frame, stripe
 * bands into a staging buffer that one kernel then puts in row order.  frame_dev: width*height uint32
 * on the root's device, or NULL for the group's own frame (luma_hip_group_frame_device).  Synchronous:
 * the frame is complete on return.  Bit-identical to a one-device luma_hip_render_device.  gather_ms is
 * the exchange (and de-interleave) on the root, from HIP events.  RCCL is bound at first use
 * (librccl.so.1); LUMA_E_RCCL if that fails or a RCCL call does.  A device id may repeat in the group:
 * its members share that device's communicator rank. */
int  luma_hip_group_render_device(luma_group* g, const luma_params* params, uint32_t* frame_dev,
                                  luma_stats* stats);
/* The group's own frame on the root device (allocated on first use, valid until the group is destroyed
 * or its frame grows), and a copy of it to host memory. */
int  luma_hip_group_frame_device(luma_group* g, uint32_t** frame_dev);
int  luma_hip_group_read_frame(luma_group* g, uint32_t* frame_host);
/* Renderer::accumulated_data of the whole frame (width*height*3 floats, AoS rgb, renderer.h:56),
 * put together from the members' bands. */
int  luma_hip_group_read_accum(luma_group* g, float* rgb_host);

/* ---- host frames ---------------------------------------------------------------- */

/* Page-locked host memory for a frame that luma_hip_render / _render_frame / _group_render fill
 * (stands in for `new std::uint32_t[width * height]`, main.cpp:63,:259): the render kernels then
 * write their pixels into it themselves, over PCIe while the rest of the frame is still being
 * traced, and no copy follows the kernel (the context's own framebuffer receives the band as well).
 * Any host pointer is accepted by the render calls; this one is only faster (1080p x 64 spp:
 * 1.84 ms per frame against 2.0 with a copy and 2.3 into pageable memory).  Visible to every device. */
int  luma_hip_host_alloc(void** out, size_t bytes);
void luma_hip_host_free(void* p);

/* ---- read-back hooks ----------------------------------------------------------- */

/* Renderer::accumulated_data for the band as the reference lays it out: W*rows*3 floats,
 * AoS rgb (renderer.h:56). */
int luma_hip_read_accum(luma_ctx* ctx, float* rgb_host);

/* The band framebuffer after a luma_hip_render_device call: copies rows*width uint32 from
 * target_dev (NULL: the context's own buffer) to target_host once the device is idle. */
int luma_hip_read_target(luma_ctx* ctx, const uint32_t* target_dev, uint32_t* target_host);

/* Frames accumulated since the last reset (== frame_count - 1, renderer.h:53). */
int luma_hip_frame_index(luma_ctx* ctx, uint32_t* frame_index);

/* Camera::recompute_rays (Camera.cpp:185-216) for the band, run as its own kernel:
 * rays_host receives rows*W*3 floats.  The render kernel computes the same directions
 * in registers and never reads such a table. */
int luma_hip_generate_rays(luma_ctx* ctx, float* rays_host);

/* ---- host-side camera and scene helpers (no GPU work) -------------------------- */

/* Camera::Camera (Camera.cpp:77-90) + recompute_projection/recompute_view (:169-183):
 * pos (0,0,0), dir (0,0,-1), depth 10, show_depth on (camera.h:33-35). */
int luma_camera_default(float fov_deg, float nearclip, float farclip,
                        uint32_t width, uint32_t height, luma_camera* out);

/* Camera::update's pose arithmetic (Camera.cpp:135-152) + recompute_view (:176-183). */
int luma_camera_from_pose(float fov_deg, float nearclip, float farclip,
                          uint32_t width, uint32_t height,
                          const float pos[3], float pitch, float yaw, luma_camera* out);

/* The reference's hard-coded scene, renderer.h:62-70, order s,a,q,r,t.  out has room for 5. */
int luma_scene_default(luma_sphere* out);

/* Build-defined large scene: sphere 0 = ground, the rest from
 * the counter RNG.  No counterpart in the reference. */
int luma_scene_synthetic(luma_sphere* out, uint32_t n, uint64_t seed);

/* Host-only view of the uniform grid LUMA_FLAG_GRID builds for a scene (no GPU work; for tests and
 * tuning).  Call with the three arrays NULL to get the sizes, then again with
 * cell_start[dims[0]*dims[1]*dims[2] + 1], items[n_items] and large[n_large].  Cell (x,y,z) is entry
 * (z*dims[1] + y)*dims[0] + x and spans org + (x,y,z)*cell .. org + (x+1,y+1,z+1)*cell. */
typedef struct {
    uint32_t dims[3];
    float    org[3];
    float    cell;
    uint32_t n_items, n_large;
} luma_grid_info;
int luma_grid_describe(const luma_sphere* spheres, uint32_t n, luma_grid_info* info,
                       uint32_t* cell_start, uint32_t* items, uint32_t* large);

/* The same for the direction grid LUMA_FLAG_GRID builds around camera.pos for the rays that start
 * there (every primary and bounce ray of the reference, Renderer.cpp:303,:416): `res` bins per edge
 * of each of six cube-map faces, bin (face*res + iv)*res + iu with face = 2*major axis + (negative),
 * u and v the other two direction components (axis order) over |major|, mapped from [-1,1] to
 * [0,res).  Call with the three arrays NULL for the sizes, then with cell_start[6*res*res + 1],
 * items[n_items], large[n_large]. */
int luma_view_grid_describe(const luma_sphere* spheres, uint32_t n, const float origin[3], uint32_t* res,
                            uint32_t* n_items, uint32_t* n_large, uint32_t* cell_start, uint32_t* items,
                            uint32_t* large);

/* Host-only view of how a render call of the ray-per-lane kernel would spread the frames of a pixel over lanes
 * (no GPU work; for tests and tuning; no counterpart in the reference): for a band of `rows` x `width`
 * pixels, `spp` frames per call, a scene of `n_spheres`, with or without LUMA_FLAG_GRID, and at most
 * `park_budget` bytes for the planes in which lanes park frames.  lanes / fine_lanes: lanes per pixel in rows
 * [0, fine_row0) and [fine_row0, rows) (fine_lanes 0: one class, fine_row0 == rows); rows [park_row0, rows)
 * park frames, park_bytes = park_depth * 12 B per such pixel; refused: more lanes were wanted than the budget allows. */
typedef struct {
    uint32_t lanes, fine_lanes, fine_row0, park_row0;
    uint64_t park_bytes;
    uint32_t refused;
    uint32_t park_depth;   /* slots per parked pixel: only frames claimed from the back are parked, frame spp-1-b in slot b */
} luma_split_plan;
int luma_split_plan_describe(uint32_t width, uint32_t rows, uint32_t spp, uint32_t n_spheres, uint32_t flags,
                             uint64_t park_budget, luma_split_plan* plan);

/* Host-only view of how a group cuts a frame and how luma_hip_group_render_device moves the bands to the root (no GPU
 * work; for tests; no counterpart in the reference, which renders in one thread, Renderer.cpp:393): for n members with
 * the given device ids (NULL: 0..n-1; ids may repeat) and a width x height frame.  The communicator
 * (ncclCommInitAll) has one rank per DISTINCT device, in order of first appearance: rank_devices[n_ranks], rank 0 is the
 * root.  Member i renders rows [row_begin, row_begin + rows) (contiguous bands), or with `striped` the stripes
 * i, i + n, i + 2n, ... of stripe_rows rows each; its communicator rank comm_rank ncclSend's `pixels` uint32 and the root
 * ncclRecv's them from that rank, in member order, at recv_offset pixels into the frame (contiguous) or into the staging
 * buffer that the de-interleave kernel then puts in row order (striped).  members[n] and rank_devices[n] may be NULL. */
typedef struct {
    uint32_t n_members, n_ranks, striped, stripe_rows;
} luma_gather_plan;
typedef struct {
    int32_t  device, comm_rank;
    uint32_t row_begin, rows;
    uint64_t pixels, recv_offset;
} luma_gather_member;
int luma_gather_plan_describe(int n, const int* device_ids, uint32_t width, uint32_t height, luma_gather_plan* plan,
                              luma_gather_member* members, int* rank_devices);

/* ---- diagnostics --------------------------------------------------------------- */
const char* luma_hip_strerror(int code);
const char* luma_hip_last_error(void);   /* thread-local text of the last failure */
int         luma_hip_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
