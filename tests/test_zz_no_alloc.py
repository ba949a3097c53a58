"""luma_hip_render_device_noalloc (include/luma_hip.h): the entry point that allocates nothing.

Called through the C-ABI and compared, bit for bit (packed band and accumulation planes), with
luma_hip_render_device on a second context given the same inputs -- on two calls in a row, the
second with a smaller input, so that the shared static scratch array is reused with stale
contents of a larger call in it.  Cases cover the three users of the scratch: the parking planes
of SPLIT renders (small bands of small scenes), the tile counter of the sphere-parallel kernel
(>= 320 spheres), and the ray counters (every launch)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 0x4C554D41


def _pair(abi, scene, W, H):
    """Two contexts with the same scene, camera and frame: the existing entry point and the new one."""
    ctxs = []
    for _ in range(2):
        c = abi.Context(0)
        c.set_scene(scene)
        c.set_camera(abi.default_camera(W, H))
        c.set_frame(W, H)
        ctxs.append(c)
    return ctxs


def _device_free_bytes():
    """hipMemGetInfo through the HIP runtime this process has already loaded (found in
    /proc/self/maps, so no soname is spelled out here)."""
    path = None
    with open("/proc/self/maps") as f:
        for line in f:
            if "libamdhip64" in line:
                path = line.split()[-1]
                break
    assert path, "the HIP runtime is not loaded"
    hip = C.CDLL(path)
    free, total = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def _check_call(abi, ref, new, W, H, what, check_mem=True, **kw):
    """check_mem: free device memory must not drop across the call.  Not asked of a stream's first
    call: the HIP runtime itself sets up per-stream resources (2 MiB seen) at a stream's first
    memset / launch, whichever entry point issues it."""
    p = abi.make_params(W, H, **kw)
    ref.render_device(p)
    before = _device_free_bytes()
    new.render_device_noalloc(p)
    got = new.read_target()                       # waits for the device
    after = _device_free_bytes()
    want = ref.read_target()
    assert int((got != want).sum()) == 0, f"{what}: packed band differs"
    assert np.array_equal(new.read_accum().view(np.uint32), ref.read_accum().view(np.uint32)), f"{what}: accumulation planes"
    assert new.frame_index() == ref.frame_index()
    assert after >= before or not check_mem, f"{what}: device memory shrank by {before - after} bytes across the call"


def test_noalloc_equals_render_device_two_calls_second_smaller(abi):
    scene = abi.default_scene()
    # call 1: 640x360 x 16 spp -> 4 lanes per pixel, 44 MB of parking planes in the static array
    W, H = 640, 360
    ref, new = _pair(abi, scene, W, H)
    try:
        kw = dict(bounces=5, paths=1, samples=1, mode=1, flags=0, seed=SEED)
        _check_call(abi, ref, new, W, H, "640x360 x16", check_mem=False, spp=16, **kw)
        # call 2, smaller input on the same contexts' GPU: fewer frames, accumulating on call 1
        _check_call(abi, ref, new, W, H, "640x360 x5 (accumulating)", spp=5, **kw)
    finally:
        ref.close(); new.close()
    # call 3: a smaller frame in fresh contexts; the array still holds call 1's parked frames
    W, H = 200, 120
    ref, new = _pair(abi, scene, W, H)
    try:
        _check_call(abi, ref, new, W, H, "200x120 x8", check_mem=False, spp=8, **kw)
        _check_call(abi, ref, new, W, H, "200x120 x1", spp=1, **dict(kw, flags=abi.FLAG_NOISE_OFF))
    finally:
        ref.close(); new.close()


def test_noalloc_large_band_parks_only_its_last_rows(abi):
    # 1920x1080: one lane per pixel, and the last rows of the launch four (the fine class), whose
    # frames park in the static array: as many rows as fit it at 64 spp (45 of the 108 the other entry point uses),
    # all 108 at 16 spp.  Same bits as luma_hip_render_device either way.
    W, H = 1920, 1080
    ref, new = _pair(abi, abi.default_scene(), W, H)
    try:
        kw = dict(bounces=5, paths=1, samples=1, mode=1, flags=0, seed=SEED)
        _check_call(abi, ref, new, W, H, "1080p x64", check_mem=False, spp=64, **kw)
        _check_call(abi, ref, new, W, H, "1080p x16 (accumulating)", spp=16, **kw)
    finally:
        ref.close(); new.close()


def test_noalloc_large_scene_tile_counter(abi):
    # 512 spheres: the sphere-parallel kernel pulls tiles from the tile counter in the static array
    scene = abi.synthetic_scene(512, SEED)
    W, H = 256, 96
    ref, new = _pair(abi, scene, W, H)
    try:
        kw = dict(bounces=3, paths=1, samples=1, mode=1, flags=0, seed=SEED)
        _check_call(abi, ref, new, W, H, "512 spheres x2", check_mem=False, spp=2, **kw)
        _check_call(abi, ref, new, W, H, "512 spheres x1", spp=1, **kw)
        # LUMA_FLAG_GRID needs allocations (tables, and runtime memory for its kernels' spills): refused
        with pytest.raises(abi.LumaError) as e:
            new.render_device_noalloc(abi.make_params(W, H, spp=1, **dict(kw, flags=abi.FLAG_GRID)))
        assert e.value.code == abi.E_ARG and "LUMA_FLAG_GRID" in str(e.value)
        assert new.frame_index() == ref.frame_index()
        _check_call(abi, ref, new, W, H, "512 spheres x3 after a refused call", spp=3, **kw)
    finally:
        ref.close(); new.close()


def test_noalloc_rejects_what_does_not_fit(abi):
    W, H = 640, 360
    with abi.Context(0) as ctx:
        ctx.set_scene(abi.default_scene())
        ctx.set_camera(abi.default_camera(W, H))
        ctx.set_frame(W, H)
        kw = dict(bounces=5, paths=1, samples=1, mode=1, flags=0, seed=SEED)
        need = 32 * 1024 + abi.split_plan(W, H, 64, 5).park_depth * 12 * W * H     # slots for the frames claimed from the back
        assert need > abi.NOALLOC_SCRATCH_BYTES
        with pytest.raises(abi.LumaError) as e:
            ctx.render_device_noalloc(abi.make_params(W, H, spp=64, **kw))
        assert e.value.code == abi.E_ARG
        assert str(need) in str(e.value) and str(abi.NOALLOC_SCRATCH_BYTES) in str(e.value)
        assert ctx.frame_index() == 0
        # ... and the advice in the message works: 4 x 16 spp == 64 spp of luma_hip_render_device
        for _ in range(4):
            ctx.render_device_noalloc(abi.make_params(W, H, spp=16, **kw))
        got, acc = ctx.read_target(), ctx.read_accum()
    with abi.Context(0) as ref:
        ref.set_scene(abi.default_scene())
        ref.set_camera(abi.default_camera(W, H))
        ref.set_frame(W, H)
        ref.render_device(abi.make_params(W, H, spp=64, **kw))
        assert int((got != ref.read_target()).sum()) == 0
        assert np.array_equal(acc.view(np.uint32), ref.read_accum().view(np.uint32))
