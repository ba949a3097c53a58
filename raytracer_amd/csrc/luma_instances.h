// luma_instances.h -- which instances of the render kernel templates (luma_device.hpp) the library holds, as X-macro
// lists: luma_kernels.hip declares them `extern template`, each luma_inst_*.hip defines one group, so that the groups
// compile in parallel (make -j) instead of as one 100-second translation unit.  launch_render's tables name exactly these.
#pragma once

#define LUMA_KERNEL_EXTERN(...) extern template __global__ void __VA_ARGS__(const RenderArgs);
#define LUMA_KERNEL_DEFINE(...) template __global__ void __VA_ARGS__(const RenderArgs);

// ray per lane, run-time sphere count
#define LUMA_LIST_GENERAL(X) \
    X(render_kernel<kLaneTileW, 4, SCENE_GLOBAL, false, false, 1, false>) \
    X(render_kernel<kLaneTileW, 4, SCENE_GLOBAL, false, true, 1, false>) \
    X(render_kernel<kLaneTileW, 4, SCENE_GLOBAL, true, false, 1, false>) \
    X(render_kernel<kLaneTileW, 4, SCENE_GLOBAL, true, true, 1, false>) \
    X(render_kernel<kLaneTileW, 4, SCENE_GLOBAL, false, false, 2, false>) \
    X(render_kernel<kLaneTileW, 4, SCENE_GLOBAL, false, true, 2, false>) \
    X(render_kernel<kLaneTileW, 4, SCENE_GLOBAL, true, false, 2, false>) \
    X(render_kernel<kLaneTileW, 4, SCENE_GLOBAL, true, true, 2, false>) \
    X(render_kernel<kLaneTileW, 4, SCENE_GLOBAL, false, false, 4, false>) \
    X(render_kernel<kLaneTileW, 4, SCENE_GLOBAL, false, true, 4, false>) \
    X(render_kernel<kLaneTileW, 4, SCENE_GLOBAL, true, false, 4, false>) \
    X(render_kernel<kLaneTileW, 4, SCENE_GLOBAL, true, true, 4, false>) \
    X(render_kernel<kLaneTileW, 4, SCENE_LDS_ALL, false, false, 1, false>) \
    X(render_kernel<kLaneTileW, 4, SCENE_LDS_ALL, false, true, 1, false>) \
    X(render_kernel<kLaneTileW, 4, SCENE_LDS_ALL, true, false, 1, false>) \
    X(render_kernel<kLaneTileW, 4, SCENE_LDS_ALL, true, true, 1, false>) \
    X(render_kernel<kLaneTileW, 4, SCENE_LDS_ALL, false, false, 2, false>) \
    X(render_kernel<kLaneTileW, 4, SCENE_LDS_ALL, false, true, 2, false>) \
    X(render_kernel<kLaneTileW, 4, SCENE_LDS_ALL, true, false, 2, false>) \
    X(render_kernel<kLaneTileW, 4, SCENE_LDS_ALL, true, true, 2, false>) \
    X(render_kernel<kLaneTileW, 4, SCENE_LDS_ALL, false, false, 4, false>) \
    X(render_kernel<kLaneTileW, 4, SCENE_LDS_ALL, false, true, 4, false>) \
    X(render_kernel<kLaneTileW, 4, SCENE_LDS_ALL, true, false, 4, false>) \
    X(render_kernel<kLaneTileW, 4, SCENE_LDS_ALL, true, true, 4, false>)

// ray per lane under LUMA_FLAG_GRID
#define LUMA_LIST_GRID(X) \
    X(render_kernel<kLaneTileW, 4, SCENE_GLOBAL, false, false, 1, true>) \
    X(render_kernel<kLaneTileW, 4, SCENE_GLOBAL, false, true, 1, true>) \
    X(render_kernel<kLaneTileW, 4, SCENE_GLOBAL, true, false, 1, true>) \
    X(render_kernel<kLaneTileW, 4, SCENE_GLOBAL, true, true, 1, true>) \
    X(render_kernel<kLaneTileW, 4, SCENE_GLOBAL, false, false, 2, true>) \
    X(render_kernel<kLaneTileW, 4, SCENE_GLOBAL, false, true, 2, true>) \
    X(render_kernel<kLaneTileW, 4, SCENE_GLOBAL, true, false, 2, true>) \
    X(render_kernel<kLaneTileW, 4, SCENE_GLOBAL, true, true, 2, true>) \
    X(render_kernel<kLaneTileW, 4, SCENE_GLOBAL, false, false, 4, true>) \
    X(render_kernel<kLaneTileW, 4, SCENE_GLOBAL, false, true, 4, true>) \
    X(render_kernel<kLaneTileW, 4, SCENE_GLOBAL, true, false, 4, true>) \
    X(render_kernel<kLaneTileW, 4, SCENE_GLOBAL, true, true, 4, true>)

// ray per lane, compile-time sphere count N (1..16), camera-origin or hit-point bounce rays
#define LUMA_LIST_NS_N(X, N) \
    X(render_kernel<kLaneTileW, 4, SCENE_LDS_ALL, false, false, 1, false, N>) \
    X(render_kernel<kLaneTileW, 4, SCENE_LDS_ALL, false, true, 1, false, N>) \
    X(render_kernel<kLaneTileW, 4, SCENE_LDS_ALL, true, false, 1, false, N>) \
    X(render_kernel<kLaneTileW, 4, SCENE_LDS_ALL, true, true, 1, false, N>) \
    X(render_kernel<kLaneTileW, 4, SCENE_LDS_ALL, false, false, 2, false, N>) \
    X(render_kernel<kLaneTileW, 4, SCENE_LDS_ALL, false, true, 2, false, N>) \
    X(render_kernel<kLaneTileW, 4, SCENE_LDS_ALL, true, false, 2, false, N>) \
    X(render_kernel<kLaneTileW, 4, SCENE_LDS_ALL, true, true, 2, false, N>) \
    X(render_kernel<kLaneTileW, 4, SCENE_LDS_ALL, false, false, 4, false, N>) \
    X(render_kernel<kLaneTileW, 4, SCENE_LDS_ALL, false, true, 4, false, N>) \
    X(render_kernel<kLaneTileW, 4, SCENE_LDS_ALL, true, false, 4, false, N>) \
    X(render_kernel<kLaneTileW, 4, SCENE_LDS_ALL, true, true, 4, false, N>)

#define LUMA_LIST_NS_1_4(X) LUMA_LIST_NS_N(X, 1) LUMA_LIST_NS_N(X, 2) LUMA_LIST_NS_N(X, 3) LUMA_LIST_NS_N(X, 4)

#define LUMA_LIST_NS_5_8(X) LUMA_LIST_NS_N(X, 5) LUMA_LIST_NS_N(X, 6) LUMA_LIST_NS_N(X, 7) LUMA_LIST_NS_N(X, 8)

#define LUMA_LIST_NS_9_12(X) LUMA_LIST_NS_N(X, 9) LUMA_LIST_NS_N(X, 10) LUMA_LIST_NS_N(X, 11) LUMA_LIST_NS_N(X, 12)

#define LUMA_LIST_NS_13_16(X) LUMA_LIST_NS_N(X, 13) LUMA_LIST_NS_N(X, 14) LUMA_LIST_NS_N(X, 15) LUMA_LIST_NS_N(X, 16)

// the same with a second class of workgroups (FINE lanes per pixel in the last rows), N = 1..8
#define LUMA_LIST_FINE_N(X, N) \
    X(render_kernel<kLaneTileW, 4, SCENE_LDS_ALL, false, false, 1, false, N, 4>) \
    X(render_kernel<kLaneTileW, 4, SCENE_LDS_ALL, true, false, 1, false, N, 4>) \
    X(render_kernel<kLaneTileW, 4, SCENE_LDS_ALL, false, false, 2, false, N, 4>) \
    X(render_kernel<kLaneTileW, 4, SCENE_LDS_ALL, true, false, 2, false, N, 4>) \
    X(render_kernel<kLaneTileW, 4, SCENE_LDS_ALL, false, false, 4, false, N, 8>) \
    X(render_kernel<kLaneTileW, 4, SCENE_LDS_ALL, true, false, 4, false, N, 8>)

#define LUMA_LIST_FINE_1_4(X) LUMA_LIST_FINE_N(X, 1) LUMA_LIST_FINE_N(X, 2) LUMA_LIST_FINE_N(X, 3) LUMA_LIST_FINE_N(X, 4)

#define LUMA_LIST_FINE_5_8(X) LUMA_LIST_FINE_N(X, 5) LUMA_LIST_FINE_N(X, 6) LUMA_LIST_FINE_N(X, 7) LUMA_LIST_FINE_N(X, 8)

// TAKES: one lane per pixel, lanes out of frames take frames of their wave's other pixels (with and without the fine class)
#define LUMA_LIST_TAKES_N(X, N) \
    X(render_kernel<kLaneTileW, 4, SCENE_LDS_ALL, true, false, 1, false, N, 0, true>) \
    X(render_kernel<kLaneTileW, 4, SCENE_LDS_ALL, true, true, 1, false, N, 0, true>)
#define LUMA_LIST_TAKES_FINE_N(X, N) \
    X(render_kernel<kLaneTileW, 4, SCENE_LDS_ALL, true, false, 1, false, N, 4, true>)

#define LUMA_LIST_TAKES_1_8(X) LUMA_LIST_TAKES_N(X, 1) LUMA_LIST_TAKES_N(X, 2) LUMA_LIST_TAKES_N(X, 3) LUMA_LIST_TAKES_N(X, 4) \
    LUMA_LIST_TAKES_N(X, 5) LUMA_LIST_TAKES_N(X, 6) LUMA_LIST_TAKES_N(X, 7) LUMA_LIST_TAKES_N(X, 8)

#define LUMA_LIST_TAKES_FINE(X) LUMA_LIST_TAKES_FINE_N(X, 1) LUMA_LIST_TAKES_FINE_N(X, 2) LUMA_LIST_TAKES_FINE_N(X, 3) LUMA_LIST_TAKES_FINE_N(X, 4) \
    LUMA_LIST_TAKES_FINE_N(X, 5) LUMA_LIST_TAKES_FINE_N(X, 6) LUMA_LIST_TAKES_FINE_N(X, 7) LUMA_LIST_TAKES_FINE_N(X, 8)

#define LUMA_LIST_TAKES_9_16(X) LUMA_LIST_TAKES_N(X, 9) LUMA_LIST_TAKES_N(X, 10) LUMA_LIST_TAKES_N(X, 11) LUMA_LIST_TAKES_N(X, 12) \
    LUMA_LIST_TAKES_N(X, 13) LUMA_LIST_TAKES_N(X, 14) LUMA_LIST_TAKES_N(X, 15) LUMA_LIST_TAKES_N(X, 16)

// ... and the run-time-sphere-count instances with frames taken (17 to kTakesGeneralSpheresMax spheres)
#define LUMA_LIST_TAKES_GENERAL(X) \
    X(render_kernel<kLaneTileW, 4, SCENE_LDS_ALL, true, false, 1, false, 0, 0, true>) \
    X(render_kernel<kLaneTileW, 4, SCENE_LDS_ALL, true, true, 1, false, 0, 0, true>)

// sphere-parallel scan
#define LUMA_LIST_SP(X) \
    X(render_kernel_sp<kTileW, 4, false, false>) \
    X(render_kernel_sp<kTileW, 4, false, true>) \
    X(render_kernel_sp<kTileW, 4, true, false>) \
    X(render_kernel_sp<kTileW, 4, true, true>) \
    X(render_kernel_sp<kTileW, 16, false, false>) \
    X(render_kernel_sp<kTileW, 16, false, true>) \
    X(render_kernel_sp<kTileW, 16, true, false>) \
    X(render_kernel_sp<kTileW, 16, true, true>)

// wave-cooperative grid walk (LUMA_FLAG_GRID, large scenes)
#define LUMA_LIST_COOP(X) \
    X(render_kernel_sp<kTileW, 8, false, false, true, 0>) \
    X(render_kernel_sp<kTileW, 8, false, true, true, 0>) \
    X(render_kernel_sp<kTileW, 8, true, false, true, 0>) \
    X(render_kernel_sp<kTileW, 8, true, true, true, 0>) \
    X(render_kernel_sp<kTileW, 16, false, false, true, 1>) \
    X(render_kernel_sp<kTileW, 16, false, true, true, 1>) \
    X(render_kernel_sp<kTileW, 16, true, false, true, 1>) \
    X(render_kernel_sp<kTileW, 16, true, true, true, 1>) \
    X(render_kernel_sp<kTileW, 16, false, false, true, 2>) \
    X(render_kernel_sp<kTileW, 16, false, true, true, 2>) \
    X(render_kernel_sp<kTileW, 16, true, false, true, 2>) \
    X(render_kernel_sp<kTileW, 16, true, true, true, 2>)

#define LUMA_LIST_ALL(X) LUMA_LIST_GENERAL(X) LUMA_LIST_GRID(X) LUMA_LIST_NS_1_4(X) LUMA_LIST_NS_5_8(X) LUMA_LIST_NS_9_12(X) LUMA_LIST_NS_13_16(X) LUMA_LIST_FINE_1_4(X) LUMA_LIST_FINE_5_8(X) LUMA_LIST_TAKES_1_8(X) LUMA_LIST_TAKES_FINE(X) LUMA_LIST_TAKES_9_16(X) LUMA_LIST_TAKES_GENERAL(X) LUMA_LIST_SP(X) LUMA_LIST_COOP(X)
