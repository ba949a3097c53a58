"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle on the same inputs.

Bar (BASELINE.json north_star): <= 1 LSB per channel for deterministic primary-ray shading,
and a stated MAE/PSNR bound for multi-bounce accumulation.  Because kernels and oracle share
one arithmetic contract, the tests below demand the stronger result -- bit-exact
frames and bit-exact accumulation buffers -- and say so where they check it.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 0x4C554D41


def channels(img):
    img = np.asarray(img, dtype=np.uint32)
    return np.stack([(img >> 16) & 0xFF, (img >> 8) & 0xFF, img & 0xFF], axis=-1).astype(np.int32)


def assert_frames_equal(got, want, what):
    nd = int((got != want).sum())
    if nd:
        d = np.abs(channels(got) - channels(want))
        raise AssertionError(f"{what}: {nd} pixels differ, max channel delta {int(d.max())} LSB, "
                             f"MAE {float(d.mean()):.5f} LSB")


def hip_render(abi, spheres, cam, W, H, rows=None, **kw):
    p = abi.make_params(W, H, **kw)
    with abi.Context(0) as ctx:
        ctx.set_scene(spheres)
        ctx.set_camera(cam)
        ctx.set_frame(W, H, *(rows or (0, H)))
        img, st = ctx.render(p)
        acc = ctx.read_accum()
    return img, acc, st


def to_abi_scene(abi, osph):
    arr = (abi.Sphere * len(osph))()
    for i, s in enumerate(osph):
        arr[i] = abi.Sphere(s.pos[0], s.pos[1], s.pos[2], s.radius, s.diffuse[0], s.diffuse[1], s.diffuse[2],
                            s.albedo, s.metallic, s.roughness)
    return arr


def to_abi_camera(abi, ocam):
    cam = abi.Camera()
    cam.pos[:] = ocam.pos[:]
    cam.proj_inv[:] = ocam.proj_inv[:]
    cam.view_inv[:] = ocam.view_inv[:]
    cam.focus_depth = ocam.focus_depth
    cam.show_depth = ocam.show_depth
    return cam


# ---- K1: camera ray generation (Camera.cpp:185-216) ----------------------------------

@pytest.mark.parametrize("W,H", [(64, 36), (161, 91), (1, 1), (1920, 1080)])
def test_ray_table_bit_exact(abi, oracle, W, H):
    with abi.Context(0) as ctx:
        ctx.set_camera(abi.default_camera(W, H))
        ctx.set_frame(W, H)
        got = ctx.generate_rays()
    want = oracle.camera_rays(oracle.camera(W, H), W, H)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_ray_table_posed_camera(abi, oracle):
    W, H = 96, 54
    pos, pitch, yaw = (1.5, -0.25, 3.0), 0.3, -0.7
    with abi.Context(0) as ctx:
        ctx.set_camera(abi.camera_from_pose(W, H, pos, pitch, yaw))
        ctx.set_frame(W, H)
        got = ctx.generate_rays()
    want = oracle.camera_rays(oracle.camera_pose(W, H, pos, pitch, yaw), W, H)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


# ---- config #1 / #2: deterministic primary-ray shading -------------------------------

@pytest.mark.parametrize("W,H", [(64, 36), (160, 90), (1280, 720), (1920, 1080)])
def test_primary_rays_noise_off(abi, oracle, W, H):
    """BASELINE configs #1 (1280x720) and #2 (1920x1080): 1 spp, bounces=1, NOISE_OFF.
    Contract bar <= 1 LSB/channel; checked bit-exact."""
    kw = dict(spp=1, bounces=1, mode=0, flags=abi.FLAG_NOISE_OFF, seed=SEED)
    got, acc, st = hip_render(abi, abi.default_scene(), abi.default_camera(W, H), W, H, **kw)
    want, wacc, ost = oracle.render(oracle.default_scene(), oracle.camera(W, H), oracle.make_params(W, H, **kw))
    d = np.abs(channels(got) - channels(want))
    assert d.max() <= 1, f"max delta {d.max()} LSB"
    assert_frames_equal(got, want, f"primary {W}x{H}")
    assert np.array_equal(acc.view(np.uint32), wacc.view(np.uint32))
    assert st.rays == ost.rays == 2 * W * H          # DOF pre-pass + one bounce per pixel
    assert st.sphere_tests == ost.sphere_tests == 10 * W * H


def test_config1_reference_noise_on(abi, oracle):
    """Config #1 as the committed reference behaves (noise on, raytrace mode, 1 spp)."""
    W, H = 1280, 720
    kw = dict(spp=1, bounces=1, mode=0, flags=0, seed=SEED)
    got, acc, _ = hip_render(abi, abi.default_scene(), abi.default_camera(W, H), W, H, **kw)
    want, wacc, _ = oracle.render(oracle.default_scene(), oracle.camera(W, H), oracle.make_params(W, H, **kw))
    assert_frames_equal(got, want, "config #1 noise on")
    assert np.array_equal(acc.view(np.uint32), wacc.view(np.uint32))


# ---- config #3: multi-bounce accumulation --------------------------------------------

@pytest.mark.parametrize("W,H,spp,bounces,paths,samples", [
    (64, 36, 4, 5, 1, 1),
    (160, 90, 16, 5, 1, 1),
    (97, 55, 3, 4, 3, 2),
    (33, 7, 2, 8, 0, 1),
])
def test_pathtrace_multibounce(abi, oracle, W, H, spp, bounces, paths, samples):
    """Stated bound for multi-bounce accumulation: MAE <= 0.5 LSB and PSNR >= 45 dB
    (BASELINE.md section 4).  Checked bit-exact, frame and accumulation buffer."""
    kw = dict(spp=spp, bounces=bounces, paths=paths, samples=samples, mode=1, flags=0, seed=SEED)
    got, acc, st = hip_render(abi, abi.default_scene(), abi.default_camera(W, H), W, H, **kw)
    want, wacc, ost = oracle.render(oracle.default_scene(), oracle.camera(W, H), oracle.make_params(W, H, **kw))
    d = (channels(got) - channels(want)).astype(np.float64)
    mae, mse = np.abs(d).mean(), (d * d).mean()
    psnr = np.inf if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)
    assert mae <= 0.5 and psnr >= 45.0, f"MAE {mae} LSB, PSNR {psnr} dB"
    assert_frames_equal(got, want, "pathtrace")
    assert np.array_equal(acc.view(np.uint32), wacc.view(np.uint32))
    # the kernel traces the noise-free DOF ray once per call, the reference once per frame
    assert st.rays == ost.rays - (spp - 1) * W * H


@pytest.mark.parametrize("W,H,spp,n,flags,first", [
    (64, 2, 8, 5, 0, 0), (97, 13, 16, 5, 0, 0), (160, 90, 33, 5, 0, 7), (200, 40, 64, 5, 0, 0), (64, 8, 128, 5, 0, 3),
    (33, 5, 129, 5, 0, 0),                       # one frame past kStealSppMax: the twin instance, nothing taken
    (96, 31, 24, 5, 2, 0),                       # bounce rays from hit points
    (70, 33, 20, 1, 0, 0), (70, 33, 20, 12, 0, 0), (70, 33, 20, 16, 2, 1),   # other compile-time sphere counts
    (1920, 9, 48, 5, 0, 0),                      # a fine class under the one-lane rows
])
def test_frames_taken_inside_waves_vs_oracle(abi, oracle, W, H, spp, n, flags, first):
    """One lane per pixel, 8 to 128 frames per call: a lane that has no frame of its own left renders frames from the
    back of another pixel of its wave, and that pixel's lane adds them in frame order (TAKES instances).
    Frames and accumulators are the oracle's bit for bit, also on a second call that continues the accumulation."""
    kw = dict(spp=spp, bounces=5, paths=1, samples=1, mode=1, flags=flags, seed=SEED)
    scene_g = abi.default_scene() if n == 5 else abi.synthetic_scene(n)
    scene_o = oracle.default_scene() if n == 5 else oracle.synthetic_scene(n)
    r = oracle.Renderer(scene_o, oracle.camera(W, H), W, H)
    with abi.Context(0) as ctx:
        ctx.set_scene(scene_g); ctx.set_camera(abi.default_camera(W, H)); ctx.set_frame(W, H, 0, H)
        for call in range(2 if first else 1):
            p = abi.make_params(W, H, **kw)
            got, st = ctx.render(p, want_stats=True)
            r.render(oracle.make_params(W, H, **kw))
            assert_frames_equal(got, r.target, f"frames taken, call {call}")
            assert np.array_equal(ctx.read_accum().view(np.uint32), r.accum.view(np.uint32))
            kw["spp"] = first or spp                 # the second call adds `first` more frames to the same accumulators


@pytest.mark.parametrize("W,H,spp,n,flags,limit", [(640, 360, 64, 5, 0, 8 << 20), (512, 300, 100, 5, 0, 0), (640, 360, 48, 3, 2, 1 << 20),
                                                   (97, 55, 128, 16, 0, 0),
                                                   (200, 150, 3, 5, 0, 0), (160, 90, 7, 5, 2, 0), (97, 55, 2, 8, 0, 0),    # 2 to 7 frames: no fine class
                                                   (160, 90, 24, 40, 0, 0), (128, 72, 9, 100, 2, 0), (96, 54, 16, 17, 0, 0),   # run-time sphere counts (17-256)
                                                   (96, 54, 16, 256, 0, 0), (96, 54, 64, 200, 2, 0), (96, 54, 16, 257, 0, 0)])   # ... to their end and just past it
def test_frames_taken_on_a_small_frame_vs_the_whole_oracle_frame(abi, oracle, W, H, spp, n, flags, limit):
    """The TAKES instances run where a band has one lane per pixel: from 1.5 M pixels, or where four lanes' parking does
    not fit its budget.  luma_hip_set_park_limit makes small frames such bands (with a fine class as large as the limit
    holds, or none), so the oracle can render all of them: every pixel and accumulator equal, and again after a second
    call that continues the accumulation."""
    plan = abi.split_plan(W, H, spp, n, flags=flags, park_budget=limit)
    assert plan.lanes == 1 and plan.refused == 1 and plan.park_bytes <= limit
    kw = dict(spp=spp, bounces=5, paths=1, samples=1, mode=1, flags=flags, seed=SEED)
    scene_g = abi.default_scene() if n == 5 else abi.synthetic_scene(n)
    scene_o = oracle.default_scene() if n == 5 else oracle.synthetic_scene(n)
    r = oracle.Renderer(scene_o, oracle.camera(W, H), W, H)
    with abi.Context(0) as ctx:
        ctx.set_scene(scene_g); ctx.set_camera(abi.default_camera(W, H)); ctx.set_frame(W, H, 0, H)
        ctx.set_park_limit(limit)
        for spp_call in (spp, 9):
            kw["spp"] = spp_call
            got, st = ctx.render(abi.make_params(W, H, **kw), want_stats=True)
            r.render(oracle.make_params(W, H, **kw))
            assert st.lanes_per_pixel == 1
            assert_frames_equal(got, r.target, f"frames taken, small frame, {spp_call} spp")
            assert np.array_equal(ctx.read_accum().view(np.uint32), r.accum.view(np.uint32))
        with pytest.raises(abi.LumaError):
            ctx.set_park_limit(3 << 30)


def test_config3_full_size_band_vs_oracle(abi, oracle):
    """Config #3 at full size (1920x1080, 64 spp, 5 bounces, pathtrace) on the GPU; the
    oracle renders three 4-row bands of the same frame (it needs ~40 s for all of it)."""
    W, H = 1920, 1080
    kw = dict(spp=64, bounces=5, paths=1, samples=1, mode=1, flags=0, seed=SEED)
    got, acc, st = hip_render(abi, abi.default_scene(), abi.default_camera(W, H), W, H, **kw)
    ocam, osc, op = oracle.camera(W, H), oracle.default_scene(), oracle.make_params(W, H, **kw)
    for y0 in (0, 538, 1076):
        r = oracle.Renderer(osc, ocam, W, H)
        r.render(op, rows=(y0, y0 + 4))
        assert_frames_equal(got[y0:y0 + 4], r.target[y0:y0 + 4], f"config #3 rows {y0}..{y0 + 4}")
        assert np.array_equal(acc[y0:y0 + 4].view(np.uint32), r.accum[y0:y0 + 4].view(np.uint32))
    assert st.rays > 2 * W * H * 64 // 2


def test_config4_full_size_rows_vs_oracle(abi, oracle):
    """Config #4's frame (3840x2160, 256 spp, 5 bounces, pathtrace) on one GPU as two of the eight
    interleaved-stripe bands; the oracle renders one row of each (256 spp x 3840 px ~ 0.3 s)."""
    from raytracer_amd import bands
    W, H = 3840, 2160
    kw = dict(spp=256, bounces=5, paths=1, samples=1, mode=1, flags=0, seed=SEED)
    ocam, osc, op = oracle.camera(W, H), oracle.default_scene(), oracle.make_params(W, H, **kw)
    for rank in (0, 5):
        rows = bands.striped_rows(H, 8, rank, 15)
        with abi.Context(0) as ctx:
            ctx.set_scene(abi.default_scene())
            ctx.set_camera(abi.default_camera(W, H))
            ctx.set_frame_striped(W, H, 15, rank, 8)
            got, st = ctx.render(abi.make_params(W, H, **kw))
        for li in (0, 137, len(rows) - 1):
            y = rows[li]
            r = oracle.Renderer(osc, ocam, W, H)
            r.render(op, rows=(y, y + 1))
            assert_frames_equal(got[li:li + 1], r.target[y:y + 1], f"config #4 row {y}")


def test_config5_full_size_row_vs_oracle(abi, oracle):
    """Config #5 (synthetic 4096-sphere scene, 7680x4320, 16 spp, 5 bounces): the GPU renders one
    15-row stripe of the frame with the sphere-parallel kernel; the oracle checks part of one
    row (a full 7680-px row is ~1.3e9 ray-sphere tests on the CPU)."""
    W, H = 7680, 4320
    kw = dict(spp=16, bounces=5, paths=1, samples=1, mode=1, flags=0, seed=SEED)
    y0 = 2160
    with abi.Context(0) as ctx:
        ctx.set_scene(abi.synthetic_scene(4096))
        ctx.set_camera(abi.default_camera(W, H))
        ctx.set_frame(W, H, y0, y0 + 15)
        got, st = ctx.render(abi.make_params(W, H, **kw))
    assert st.sphere_tests == st.rays * 4096
    # oracle: same frame geometry, one row, compared on its first 1536 pixels
    r = oracle.Renderer(oracle.synthetic_scene(4096), oracle.camera(W, H), W, H)
    # a full-width row at 16 spp is ~1.3e9 ray-sphere tests on one core; check the row at 4 spp
    kw4 = dict(kw, spp=4)
    r.render(oracle.make_params(W, H, **kw4), rows=(y0 + 7, y0 + 8))
    with abi.Context(0) as ctx:
        ctx.set_scene(abi.synthetic_scene(4096))
        ctx.set_camera(abi.default_camera(W, H))
        ctx.set_frame(W, H, y0 + 7, y0 + 8)
        got4, _ = ctx.render(abi.make_params(W, H, **kw4))
    assert_frames_equal(got4, r.target[y0 + 7:y0 + 8], "config #5 row, 4 spp")
    # the 16-spp stripe agrees with a 16-spp single-row render of the same row (band independence)
    with abi.Context(0) as ctx:
        ctx.set_scene(abi.synthetic_scene(4096))
        ctx.set_camera(abi.default_camera(W, H))
        ctx.set_frame(W, H, y0 + 7, y0 + 8)
        got16, _ = ctx.render(abi.make_params(W, H, **kw))
    assert_frames_equal(got[7:8], got16, "config #5 stripe vs single row")


def test_progressive_frames_equal_in_kernel_spp(abi, oracle):
    """64 render_to calls (Renderer.cpp:457-470) == one call with spp=64."""
    W, H = 80, 45
    kw = dict(bounces=5, paths=1, samples=1, mode=1, flags=0, seed=SEED)
    one, acc_one, _ = hip_render(abi, abi.default_scene(), abi.default_camera(W, H), W, H, spp=12, **kw)
    with abi.Context(0) as ctx:
        ctx.set_scene(abi.default_scene())
        ctx.set_camera(abi.default_camera(W, H))
        ctx.set_frame(W, H)
        for i in range(12):
            img, _ = ctx.render(abi.make_params(W, H, spp=1, **kw))
            assert ctx.frame_index() == i + 1
        acc = ctx.read_accum()
        assert_frames_equal(img, one, "progressive vs in-kernel spp")
        assert np.array_equal(acc.view(np.uint32), acc_one.view(np.uint32))
        # camera.moved branch (Renderer.cpp:377-390): reset, then the first frame again
        ctx.reset()
        assert ctx.frame_index() == 0
        first, _ = ctx.render(abi.make_params(W, H, spp=1, **kw))
    want, _, _ = oracle.render(oracle.default_scene(), oracle.camera(W, H), oracle.make_params(W, H, spp=1, **kw))
    assert_frames_equal(first, want, "first frame after reset")


def test_against_committed_golden_frames(abi):
    """HIP frames against tests/golden/oracle_frames.npz (made by tests/golden/make_golden.py):
    parity that does not depend on rebuilding the oracle on the GPU box."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_frames.npz"))
    cases = {
        "primary_64x36": dict(W=64, H=36, spp=1, bounces=1, mode=0, flags=abi.FLAG_NOISE_OFF),
        "primary_160x90": dict(W=160, H=90, spp=1, bounces=1, mode=0, flags=abi.FLAG_NOISE_OFF),
        "direct_noise_64x36": dict(W=64, H=36, spp=1, bounces=1, mode=0, flags=0),
        "path_64x36_4spp": dict(W=64, H=36, spp=4, bounces=5, mode=1, flags=0),
        "path_160x90_4spp": dict(W=160, H=90, spp=4, bounces=5, mode=1, flags=0),
        "path_fromhit_64x36": dict(W=64, H=36, spp=2, bounces=5, mode=1, flags=abi.FLAG_ORIGIN_FROM_HIT, paths=2),
    }
    for name, c in cases.items():
        W, H = c.pop("W"), c.pop("H")
        got, acc, _ = hip_render(abi, abi.default_scene(), abi.default_camera(W, H), W, H, seed=SEED, **c)
        assert_frames_equal(got, g[name + "_target"], name)
        assert np.array_equal(acc.view(np.uint32), g[name + "_accum"].view(np.uint32)), name
    got, _, _ = hip_render(abi, abi.synthetic_scene(300), abi.default_camera(48, 27), 48, 27,
                           spp=2, bounces=5, mode=1, seed=SEED)
    assert_frames_equal(got, g["synthetic300_48x27_target"], "synthetic300")


# ---- flags, scenes, edge cases -------------------------------------------------------

@pytest.mark.parametrize("flags", [2, 4, 8, 2 | 4 | 8, 1 | 2, 16, 4 | 16])
def test_flags(abi, oracle, flags):
    W, H = 120, 68
    kw = dict(spp=3, bounces=5, paths=2, samples=1, mode=1, flags=flags, seed=7)
    got, acc, _ = hip_render(abi, abi.default_scene(), abi.default_camera(W, H), W, H, **kw)
    want, wacc, _ = oracle.render(oracle.default_scene(), oracle.camera(W, H), oracle.make_params(W, H, **kw))
    assert_frames_equal(got, want, f"flags={flags}")
    assert np.array_equal(acc.view(np.uint32), wacc.view(np.uint32))


def test_rgba8_pack_is_a_byte_permutation(abi):
    """LUMA_FLAG_PACK_RGBA8 with CLAMP_PACK: bytes R,G,B,255 of the same channels as ::RGB's
    0x00RRGGBB (Renderer.cpp:44-55), sphere-parallel kernel included."""
    for scene, W, H in [(abi.default_scene(), 96, 54), (abi.synthetic_scene(384), 40, 24)]:
        kw = dict(spp=2, bounces=5, paths=1, samples=1, mode=1, seed=SEED)
        ref, _, _ = hip_render(abi, scene, abi.default_camera(W, H), W, H, flags=abi.FLAG_CLAMP_PACK, **kw)
        got, _, _ = hip_render(abi, scene, abi.default_camera(W, H), W, H,
                               flags=abi.FLAG_CLAMP_PACK | abi.FLAG_PACK_RGBA8, **kw)
        b = got.view(np.uint8).reshape(H, W, 4)
        assert np.array_equal(b[..., 0], (ref >> 16) & 0xFF)
        assert np.array_equal(b[..., 1], (ref >> 8) & 0xFF)
        assert np.array_equal(b[..., 2], ref & 0xFF)
        assert (b[..., 3] == 255).all()


def test_posed_camera_render(abi, oracle):
    W, H = 128, 72
    pos, pitch, yaw = (2.0, -1.0, 4.0), 0.2, 0.35
    kw = dict(spp=2, bounces=5, paths=1, samples=1, mode=1, flags=0, seed=SEED)
    got, acc, _ = hip_render(abi, abi.default_scene(), abi.camera_from_pose(W, H, pos, pitch, yaw), W, H, **kw)
    want, wacc, _ = oracle.render(oracle.default_scene(), oracle.camera_pose(W, H, pos, pitch, yaw),
                                  oracle.make_params(W, H, **kw))
    assert_frames_equal(got, want, "posed camera")
    assert np.array_equal(acc.view(np.uint32), wacc.view(np.uint32))


@pytest.mark.parametrize("n,W,H,spp,flags,paths", [
    (300, 96, 54, 2, 0, 1), (4096, 64, 36, 1, 0, 1), (6000, 48, 27, 1, 0, 1),
    (191, 40, 24, 2, 0, 1), (257, 33, 19, 2, 0, 2), (319, 40, 24, 2, 0, 1), (320, 40, 24, 2, 0, 1),
    (321, 33, 19, 1, 0, 2), (449, 33, 19, 1, 2, 1), (1000, 50, 30, 2, 2, 1),
    (640, 37, 21, 1, 1, 1), (1024, 37, 21, 2, 1 | 2, 0), (1025, 37, 21, 1, 0, 1), (4864, 20, 12, 1, 0, 1), (4865, 20, 12, 1, 0, 1),
    (5056, 20, 12, 1, 0, 1), (5057, 20, 12, 1, 0, 1),
])
def test_synthetic_scene(abi, oracle, n, W, H, spp, flags, paths):
    """Config #5's scene generator.  n < 320: ray-per-lane scan from LDS; 320 <= n <= 5056:
    sphere-parallel scan over records padded to batches of 64 (four batches per trip, then up to
    three single ones; 4-wave workgroups up to 1024 spheres, 16-wave beyond; n=4096 fills 128 KiB
    of LDS, n=5056 all 158); larger n: ray-per-lane scan from global memory."""
    kw = dict(spp=spp, bounces=5, paths=paths, samples=1, mode=1, flags=flags, seed=SEED)
    got, acc, st = hip_render(abi, abi.synthetic_scene(n), abi.default_camera(W, H), W, H, **kw)
    want, wacc, ost = oracle.render(oracle.synthetic_scene(n), oracle.camera(W, H), oracle.make_params(W, H, **kw))
    assert_frames_equal(got, want, f"synthetic n={n}")
    assert np.array_equal(acc.view(np.uint32), wacc.view(np.uint32))
    assert st.sphere_tests == st.rays * n


@pytest.mark.parametrize("n,W,H,spp,flags,posed", [
    (5, 160, 90, 3, 0, False), (2, 64, 36, 2, 0, False), (64, 96, 54, 2, 2, False), (300, 96, 54, 2, 0, True),
    (1000, 64, 36, 2, 1, False), (4096, 96, 54, 1, 0, False), (4096, 48, 27, 2, 2, True), (20000, 64, 36, 1, 0, False),
    # the wave-cooperative walk (512 spheres and up): at its threshold, LDS resident with rays from hit points as
    # bounce rays (flag 2) and without noise (flag 1), and too large for LDS (tables in global memory) likewise
    (511, 64, 36, 2, 0, False), (512, 64, 36, 2, 0, False), (2500, 80, 45, 3, 3, True), (4500, 64, 36, 2, 2, False),
    (20000, 48, 27, 2, 2, True), (9000, 64, 36, 2, 1, True), (70000, 48, 27, 1, 2, False),
    # ... and with the grid's tables in LDS but the records in global memory (about 5100 to 11000 spheres)
    (6000, 64, 36, 2, 2, False), (7000, 48, 27, 2, 1, True), (8000, 64, 36, 1, 0, False), (5200, 48, 27, 1, 0, False),
])
def test_grid_gives_the_linear_scan_frame(abi, oracle, n, W, H, spp, flags, posed):
    """LUMA_FLAG_GRID: rays are tested only against the spheres of the grid cells they cross (plus
    the few spheres too large for the grid); frame and accumulators equal the oracle's linear scan
    bit for bit, with a fraction of the sphere tests."""
    kw = dict(spp=spp, bounces=5, paths=1, samples=1, mode=1, seed=SEED)
    scene = abi.default_scene() if n == 5 else abi.synthetic_scene(n)
    oscene = oracle.default_scene() if n == 5 else oracle.synthetic_scene(n)
    if posed:      # in the middle of the synthetic cloud, looking sideways
        cam, ocam = abi.camera_from_pose(W, H, (5.0, -4.0, -40.0), 0.3, 1.2), oracle.camera_pose(W, H, (5.0, -4.0, -40.0), 0.3, 1.2)
    else:
        cam, ocam = abi.default_camera(W, H), oracle.camera(W, H)
    got, acc, st = hip_render(abi, scene, cam, W, H, flags=flags | abi.FLAG_GRID, **kw)
    want, wacc, ost = oracle.render(oscene, ocam, oracle.make_params(W, H, flags=flags, **kw))
    assert_frames_equal(got, want, f"grid n={n}")
    assert np.array_equal(acc.view(np.uint32), wacc.view(np.uint32))
    assert st.rays == ost.rays - (spp - 1) * W * H
    if n >= 1000:
        assert st.sphere_tests * 10 < st.rays * n, "the grid should skip most of a large scene"


def test_grid_equals_linear_scan_on_a_config5_band(abi):
    """Config #5 at full size, one of its eight stripe bands (7680 x 540 pixels, 16 spp, 4096
    spheres): the grid path and the sphere-parallel linear scan give the same 4.1 million pixels and
    accumulators (the linear scan itself is checked against the oracle row by row above)."""
    W, H = 7680, 4320
    frames = {}
    for name, flags in (("linear", 0), ("grid", abi.FLAG_GRID)):
        with abi.Context(0) as ctx:
            ctx.set_scene(abi.synthetic_scene(4096))
            ctx.set_camera(abi.default_camera(W, H))
            ctx.set_frame_striped(W, H, 15, 3, 8)
            img, st = ctx.render(abi.make_params(W, H, spp=16, bounces=5, paths=1, samples=1, mode=1, flags=flags, seed=SEED))
            frames[name] = (img, ctx.read_accum(), st)
    assert_frames_equal(frames["grid"][0], frames["linear"][0], "config #5 band, grid against linear")
    assert np.array_equal(frames["grid"][1].view(np.uint32), frames["linear"][1].view(np.uint32))
    assert frames["grid"][2].rays == frames["linear"][2].rays
    assert frames["grid"][2].sphere_tests * 100 < frames["linear"][2].sphere_tests


@pytest.mark.parametrize("n,flags", [(2000, 0), (8000, 2), (30000, 0), (70000, 2)])
def test_grid_equals_linear_scan_at_every_residency_tier(abi, n, flags):
    """640x360 x 3 spp: the wave-cooperative walk with everything in LDS (2000 spheres), the tables only (8000), nothing
    (30000) and sphere indices beyond 16 bits (70000) against the linear scan of the same scene on the GPU (itself
    checked against the oracle elsewhere): frames, accumulators and ray counts equal."""
    W, H = 640, 360
    frames = {}
    for name, extra in (("linear", 0), ("grid", abi.FLAG_GRID)):
        with abi.Context(0) as ctx:
            ctx.set_scene(abi.synthetic_scene(n)); ctx.set_camera(abi.default_camera(W, H)); ctx.set_frame(W, H)
            img, st = ctx.render(abi.make_params(W, H, spp=3, bounces=5, paths=1, samples=1, mode=1, flags=flags | extra, seed=SEED))
            frames[name] = (img, ctx.read_accum(), st)
    assert_frames_equal(frames["grid"][0], frames["linear"][0], f"grid against linear, n={n}")
    assert np.array_equal(frames["grid"][1].view(np.uint32), frames["linear"][1].view(np.uint32))
    assert frames["grid"][2].rays == frames["linear"][2].rays
    assert frames["grid"][2].sphere_tests * 50 < frames["linear"][2].sphere_tests


@pytest.mark.parametrize("kind,n", [("cluster_and_outliers", 600), ("line", 600), ("coincident", 600), ("shells", 600), ("flat", 600),
                                    # the wave-cooperative walk's other tiers: tables in LDS / everything in global memory
                                    ("cluster_and_outliers", 6000), ("coincident", 6000), ("flat", 7000),
                                    ("cluster_and_outliers", 20000), ("shells", 20000), ("line", 20000)])
def test_grid_on_awkward_distributions(abi, oracle, kind, n):
    """Scenes a uniform grid handles badly (everything in one cell, a line of cells, hundreds or thousands
    of identical or concentric spheres, a plane): still the linear scan's frame."""
    rng = np.random.default_rng({"cluster_and_outliers": 1, "line": 2, "coincident": 3, "shells": 4, "flat": 5}[kind])
    pos = np.zeros((n, 3), np.float32)
    rad = np.full(n, 0.3, np.float32)
    if kind == "cluster_and_outliers":
        pos[:] = rng.normal([0, 0, -12], 0.8, size=(n, 3))
        pos[:8] = rng.uniform(-4000, 4000, size=(8, 3))
    elif kind == "line":
        pos[:, 2] = np.linspace(-4, -300, n)
        pos[:, 0] = 0.01 * np.arange(n)
    elif kind == "coincident":
        pos[:] = [0.5, 0.2, -9]
        pos[n // 2:] = [-1.5, 0.1, -7]
        rad[:] = 1.0
    elif kind == "shells":
        pos[:] = [0, 0, -15]
        rad[:] = np.linspace(0.05, 6.0, n)
    else:
        pos[:, 0] = rng.uniform(-30, 30, n); pos[:, 2] = rng.uniform(-80, -5, n); pos[:, 1] = 1.5
    a_scene, o_scene = (abi.Sphere * n)(), (oracle.Sphere * n)()
    for i in range(n):
        d = [float(v) for v in rng.uniform(0.1, 1, 3)]
        a_scene[i] = abi.Sphere(float(pos[i, 0]), float(pos[i, 1]), float(pos[i, 2]), float(rad[i]), d[0], d[1], d[2], 1.0, [.001, .1, .4][i % 3], 0.3)
        o = o_scene[i]
        o.pos[:] = [float(v) for v in pos[i]]; o.radius = float(rad[i]); o.diffuse[:] = d
        o.albedo, o.metallic, o.roughness = 1.0, [.001, .1, .4][i % 3], 0.3
    assert bytes(a_scene) == bytes(o_scene)
    W, H = (80, 45) if n <= 1000 else (48, 27)
    for flags in (1, 0, 2):
        kw = dict(spp=2 if n <= 1000 else 1, bounces=4 if n <= 1000 else 3, paths=1, samples=1, mode=1, seed=SEED)
        got, acc, st = hip_render(abi, a_scene, abi.default_camera(W, H), W, H, flags=flags | abi.FLAG_GRID, **kw)
        want, wacc, _ = oracle.render(o_scene, oracle.camera(W, H), oracle.make_params(W, H, flags=flags, **kw))
        assert_frames_equal(got, want, f"grid, {kind}, flags={flags}")
        assert np.array_equal(acc.view(np.uint32), wacc.view(np.uint32))


def test_scene_generators_match_oracle(abi, oracle):
    a, o = abi.synthetic_scene(1000), oracle.synthetic_scene(1000)
    assert bytes(a) == bytes(o)
    assert bytes(abi.default_scene()) == bytes(oracle.default_scene())


def test_empty_scene_and_tiny_frames(abi, oracle):
    kw = dict(spp=2, bounces=3, paths=1, samples=1, mode=1, flags=0, seed=3)
    for W, H in [(1, 1), (3, 2), (17, 1), (1, 9)]:
        got, _, st = hip_render(abi, (abi.Sphere * 0)(), abi.default_camera(W, H), W, H, **kw)
        want, _, _ = oracle.render((oracle.Sphere * 0)(), oracle.camera(W, H), oracle.make_params(W, H, **kw))
        assert_frames_equal(got, want, f"empty scene {W}x{H}")
        assert st.sphere_tests == 0
        got, _, _ = hip_render(abi, abi.default_scene(), abi.default_camera(W, H), W, H, **kw)
        want, _, _ = oracle.render(oracle.default_scene(), oracle.camera(W, H), oracle.make_params(W, H, **kw))
        assert_frames_equal(got, want, f"default scene {W}x{H}")


def test_zero_bounces(abi, oracle):
    """bounces=0: the loop at Renderer.cpp:300 never runs; tonemap(0) = 0 everywhere."""
    W, H = 40, 30
    kw = dict(spp=1, bounces=0, mode=0, flags=0, seed=1)
    got, _, _ = hip_render(abi, abi.default_scene(), abi.default_camera(W, H), W, H, **kw)
    want, _, _ = oracle.render(oracle.default_scene(), oracle.camera(W, H), oracle.make_params(W, H, **kw))
    assert_frames_equal(got, want, "bounces=0")
    assert not got.any()


# ---- multi-GPU partition: bands are bit-identical to the whole frame -----------------

@pytest.mark.parametrize("nbands", [2, 8])
def test_bands_equal_full_frame(abi, oracle, nbands):
    W, H = 192, 104   # 104/8 = 13 rows per band: not a multiple of the workgroup tile
    kw = dict(spp=3, bounces=5, paths=1, samples=1, mode=1, flags=0, seed=SEED)
    full, facc, fst = hip_render(abi, abi.default_scene(), abi.default_camera(W, H), W, H, **kw)
    rays = 0
    for b in range(nbands):
        y0, y1 = b * H // nbands, (b + 1) * H // nbands
        band, bacc, st = hip_render(abi, abi.default_scene(), abi.default_camera(W, H), W, H, rows=(y0, y1), **kw)
        assert band.shape == (y1 - y0, W)
        assert_frames_equal(band, full[y0:y1], f"band {b}/{nbands}")
        assert np.array_equal(bacc.view(np.uint32), facc[y0:y1].view(np.uint32))
        rays += st.rays
    assert rays == fst.rays


# (two lanes per pixel under the class only where no frames are taken inside waves: more than 128 frames per call)
@pytest.mark.parametrize("W,H,spp,lanes", [(1603, 1001, 9, 1), (1201, 701, 13, 1), (1201, 701, 130, 2), (1537, 977, 8, 1)])
def test_two_classes_of_workgroups_equal_small_bands(abi, W, H, spp, lanes):
    """Frames large enough for the fine class above one or two lanes per pixel (luma_kernels.hip plan_split: the last
    rows of the launch run four lanes per pixel, the rows above them `lanes`), with sides that are no multiple of
    the 32x2 wave tile or of the four rows a coarse workgroup covers, against the same frame rendered as 11 uneven
    bands (four lanes per pixel and their own fine class of eight: test_fine_class_of_eight_lanes_vs_oracle)."""
    kw = dict(spp=spp, bounces=5, paths=1, samples=1, mode=1, flags=0, seed=SEED)
    full, facc, fst = hip_render(abi, abi.default_scene(), abi.default_camera(W, H), W, H, **kw)
    assert fst.lanes_per_pixel == lanes
    rays = 0
    for b in range(11):
        y0, y1 = b * H // 11, (b + 1) * H // 11
        band, bacc, st = hip_render(abi, abi.default_scene(), abi.default_camera(W, H), W, H, rows=(y0, y1), **kw)
        assert st.lanes_per_pixel == 4
        assert_frames_equal(band, full[y0:y1], f"rows {y0}..{y1}")
        assert np.array_equal(bacc.view(np.uint32), facc[y0:y1].view(np.uint32))
        rays += st.rays
    assert rays == fst.rays


@pytest.mark.parametrize("W,H,spp,noise_off", [(333, 77, 9, 0), (96, 64, 8, 0), (257, 130, 17, 1), (31, 9, 64, 0)])
def test_fine_class_of_eight_lanes_vs_oracle(abi, oracle, W, H, spp, noise_off):
    """Small frames run four lanes per pixel and, from 8 spp, eight in their last rows (two lanes of each pixel side
    by side in a wave, 32x1 pixels per workgroup): whole frames against the oracle, accumulator included."""
    kw = dict(spp=spp, bounces=5, paths=1, samples=1, mode=1, flags=noise_off, seed=SEED)
    got, acc, st = hip_render(abi, abi.default_scene(), abi.default_camera(W, H), W, H, **kw)
    assert st.lanes_per_pixel == 4
    r = oracle.Renderer(oracle.default_scene(), oracle.camera(W, H), W, H)
    r.render(oracle.make_params(W, H, **kw))
    assert_frames_equal(got, r.target, "fine class of eight lanes")
    assert np.array_equal(acc.view(np.uint32), r.accum.view(np.uint32))


@pytest.mark.parametrize("W,H,stripe,n,scene_n", [(96, 48, 3, 4, 0), (64, 120, 15, 8, 0), (50, 32, 4, 2, 384)])
def test_interleaved_stripes_equal_full_frame(abi, oracle, W, H, stripe, n, scene_n):
    """luma_hip_set_frame_striped: rank r renders rows with (y / stripe) % n == r; the stripes put
    back in order are bit-identical to the one-device frame (and to the oracle)."""
    from raytracer_amd import bands
    kw = dict(spp=3, bounces=5, paths=1, samples=1, mode=1, flags=0, seed=SEED)
    scene = abi.synthetic_scene(scene_n) if scene_n else abi.default_scene()
    oscene = oracle.synthetic_scene(scene_n) if scene_n else oracle.default_scene()
    want, wacc, ost = oracle.render(oscene, oracle.camera(W, H), oracle.make_params(W, H, **kw))
    frame = np.zeros((H, W), dtype=np.uint32)
    for r in range(n):
        with abi.Context(0) as ctx:
            ctx.set_scene(scene)
            ctx.set_camera(abi.default_camera(W, H))
            ctx.set_frame_striped(W, H, stripe, r, n)
            band, _ = ctx.render(abi.make_params(W, H, **kw))
            acc = ctx.read_accum()
            rays = ctx.generate_rays()
        rows = bands.striped_rows(H, n, r, stripe)
        assert band.shape == (H // n, W)
        frame[rows] = band
        assert np.array_equal(acc.view(np.uint32), wacc[rows].view(np.uint32))
        assert np.array_equal(rays, oracle.camera_rays(oracle.camera(W, H), W, H)[rows])
    assert_frames_equal(frame, want, "striped bands")
    with abi.Context(0) as ctx:
        with pytest.raises(abi.LumaError) as e:
            ctx.set_frame_striped(64, 100, 15, 0, 8)
        assert e.value.code == abi.E_ARG


@pytest.mark.parametrize("W,H,ids,scene_n", [
    (96, 60, [0, 0], 0),          # 60 % (15*2) == 0: interleaved stripes
    (96, 60, [0, 0, 0], 0),       # 60 % 45 != 0: contiguous bands of 20 rows
    (70, 53, [0, 0, 0, 0], 0),    # ragged bands 13,13,13,14
    (48, 120, [0] * 8, 384),      # eight stripes-of-15 bands, sphere-parallel kernel
    (64, 36, [0], 0),             # a group of one
])
def test_group_renders_whole_frame(abi, oracle, W, H, ids, scene_n):
    """luma_hip_group_*: n contexts in one process render one host frame together (here all on
    device 0); frame, accumulators and ray count equal the one-context render and the oracle,
    call after call (progressive accumulation)."""
    kw = dict(spp=2, bounces=5, paths=1, samples=1, mode=1, flags=0, seed=SEED)
    scene = abi.synthetic_scene(scene_n) if scene_n else abi.default_scene()
    oscene = oracle.synthetic_scene(scene_n) if scene_n else oracle.default_scene()
    params = abi.make_params(W, H, **kw)
    r = oracle.Renderer(oscene, oracle.camera(W, H), W, H)
    with abi.Group(ids) as g:
        assert len(g) == len(ids)
        g.set_scene(scene)
        g.set_camera(abi.default_camera(W, H))
        g.set_frame(W, H)
        rays = 0
        for call in range(2):
            got, st = g.render(params)
            ost = r.render(oracle.make_params(W, H, **kw))
            assert_frames_equal(got, r.target, f"group of {len(ids)}, call {call}")
            assert np.array_equal(g.read_accum().view(np.uint32), r.accum.view(np.uint32))
            rays += st.rays
            assert st.kernel_ms > 0 and st.sphere_tests == st.rays * len(scene)
        g.reset()
        again, _ = g.render(params)
    first = oracle.Renderer(oscene, oracle.camera(W, H), W, H)
    first.render(oracle.make_params(W, H, **kw))
    assert_frames_equal(again, first.target, "after group reset")


@pytest.mark.parametrize("n", [2, 8])
def test_config3_frame_cut_for_n_gpus_equals_one_gpu(abi, n):
    """Config #3 at full size (1920x1080, 64 spp) as n interleaved-stripe bands (here all on one
    device): the bands run the frame-splitting kernel (2 lanes per pixel at n = 2, 4 at n = 8, frames
    claimed at run time), the whole frame runs one lane per pixel; same 2 million pixels."""
    W, H = 1920, 1080
    params = abi.make_params(W, H, spp=64, bounces=5, paths=1, samples=1, mode=1, flags=0, seed=SEED)
    with abi.Context(0) as ctx:
        ctx.set_scene(abi.default_scene()); ctx.set_camera(abi.default_camera(W, H)); ctx.set_frame(W, H)
        want, wst = ctx.render(params)
        wacc = ctx.read_accum()
    with abi.Group([0] * n) as g:
        g.set_scene(abi.default_scene()); g.set_camera(abi.default_camera(W, H)); g.set_frame(W, H)
        got, st = g.render(params)
        acc = g.read_accum()
    assert_frames_equal(got, want, f"config #3 in {n} bands")
    assert np.array_equal(acc.view(np.uint32), wacc.view(np.uint32))
    assert st.rays == wst.rays


@pytest.mark.parametrize("W,H,spp,scene_n,flags,what", [
    (3840, 2160, 256, 0, 0, "config #4"),
    (7680, 4320, 16, 4096, 0, "config #5"),
])
def test_config4_and_config5_as_eight_bands_equal_one_device(abi, W, H, spp, scene_n, flags, what):
    """BASELINE configs[3] and [4] at full size, cut for 8 GPUs (interleaved 15-row stripes, here all
    eight bands on one device): the assembled frame and the accumulation buffer equal the one-context
    render bit for bit, as test_config3_frame_cut_for_n_gpus_equals_one_gpu shows for config #3."""
    scene = abi.synthetic_scene(scene_n) if scene_n else abi.default_scene()
    params = abi.make_params(W, H, spp=spp, bounces=5, paths=1, samples=1, mode=1, flags=flags, seed=SEED)
    with abi.Context(0) as ctx:
        ctx.set_scene(scene); ctx.set_camera(abi.default_camera(W, H)); ctx.set_frame(W, H)
        want, wst = ctx.render(params)
        wacc = ctx.read_accum()
    with abi.Group([0] * 8) as g:
        g.set_scene(scene); g.set_camera(abi.default_camera(W, H)); g.set_frame(W, H)
        got, st = g.render(params)
        assert_frames_equal(got, want, f"{what} in 8 bands")
        del got
        acc = g.read_accum()
    assert np.array_equal(acc.view(np.uint32), wacc.view(np.uint32))
    assert st.rays == wst.rays


def test_config5_band_through_the_grid_equals_the_linear_band(abi):
    """One of config #5's eight stripe bands (7680x540, 16 spp, 4096 spheres) with LUMA_FLAG_GRID
    against the same band scanned linearly."""
    W, H = 7680, 4320
    scene = abi.synthetic_scene(4096)
    out = []
    for flags in (0, abi.FLAG_GRID):
        with abi.Context(0) as ctx:
            ctx.set_scene(scene); ctx.set_camera(abi.default_camera(W, H)); ctx.set_frame_striped(W, H, 15, 3, 8)
            img, st = ctx.render(abi.make_params(W, H, spp=16, bounces=5, paths=1, samples=1, mode=1, flags=flags, seed=SEED))
            out.append((img, ctx.read_accum(), st.rays))
    assert_frames_equal(out[1][0], out[0][0], "config #5 band, grid against linear")
    assert np.array_equal(out[1][1].view(np.uint32), out[0][1].view(np.uint32))
    assert out[1][2] == out[0][2]


def test_reserve_takes_allocation_out_of_the_render_call(abi):
    """One of config #3's eight bands (4 lanes per pixel, 199 MB of parking planes at 64 spp).
    Without luma_hip_reserve the first render call allocates the planes and the second does not;
    with it no render call allocates at all.  Same frames either way.  A call whose planes would
    pass 2 GiB runs one lane per pixel and says so in luma_stats."""
    W, H = 1920, 1080
    params = abi.make_params(W, H, spp=64, bounces=5, paths=1, samples=1, mode=1, flags=0, seed=SEED)
    frames = []
    for reserve in (False, True):
        with abi.Context(0) as ctx:
            ctx.set_scene(abi.default_scene()); ctx.set_camera(abi.default_camera(W, H))
            ctx.set_frame_striped(W, H, 15, 2, 8)
            if reserve:
                ctx.reserve(64)
            assert ctx.alloc_count() == 0
            st = ctx.render_device(params, want_stats=True)
            assert st.lanes_per_pixel == 4 and st.split_refused == 0
            first = ctx.alloc_count()
            assert first == (0 if reserve else 1)
            ctx.render_device(params)
            ctx.render_device(abi.make_params(W, H, spp=16, bounces=5, paths=1, samples=1, mode=1, flags=0, seed=SEED))
            assert ctx.alloc_count() == first                       # context-warm calls never allocate
            frames.append((ctx.read_target(), ctx.read_accum()))
    assert_frames_equal(frames[1][0], frames[0][0], "reserved against grown on demand")
    assert np.array_equal(frames[1][1].view(np.uint32), frames[0][1].view(np.uint32))
    # 1.04 M pixels x 256 spp would park 3.2 GB: refused, reported, still the right frame (config #4's band)
    W, H = 3840, 2160
    big = abi.make_params(W, H, spp=256, bounces=1, paths=1, samples=1, mode=0, flags=0, seed=SEED)
    with abi.Context(0) as ctx:
        ctx.set_scene(abi.default_scene()); ctx.set_camera(abi.default_camera(W, H))
        ctx.set_frame_striped(W, H, 15, 0, 8)
        ctx.reserve(256)
        st = ctx.render_device(big, want_stats=True)
        assert st.lanes_per_pixel == 1 and st.split_refused == 1 and ctx.alloc_count() == 0


def test_reserve_covers_grid_launches_and_smaller_calls_after_a_refused_plan(abi):
    """luma_hip_reserve plans every kind of launch the context can make, not only the linear scan at max_spp:
    (a) 400 spheres run the sphere-parallel kernel (no parking) without LUMA_FLAG_GRID but the ray-per-lane kernel with
    four lanes per pixel under it; (b) when max_spp itself is refused (parking past 2 GiB: one lane per pixel), a later
    call of fewer spp that does fit wants MORE parking than max_spp did.  Neither may allocate in the render call."""
    W, H = 1920, 1080
    with abi.Context(0) as ctx:
        ctx.set_scene(abi.synthetic_scene(400, SEED)); ctx.set_camera(abi.default_camera(W, H))
        ctx.set_frame_striped(W, H, 15, 0, 8)
        assert abi.split_plan(W, 135, 16, 400, flags=abi.FLAG_GRID).park_bytes > 0 == abi.split_plan(W, 135, 16, 400).park_bytes
        ctx.reserve(16)
        p = abi.make_params(W, H, spp=16, bounces=2, paths=1, samples=1, mode=1, flags=abi.FLAG_GRID, seed=SEED)
        st = ctx.render_device(p, want_stats=True)
        assert st.lanes_per_pixel == 4
        assert ctx.alloc_count() == 6            # the grid's own tables (3 + 3, built on first use), no parking planes
    W, H = 3840, 2160
    with abi.Context(0) as ctx:
        ctx.set_scene(abi.default_scene()); ctx.set_camera(abi.default_camera(W, H))
        ctx.set_frame_striped(W, H, 15, 0, 8)
        # (160 frames: past kStealSppMax, so two lanes per pixel and the whole band parks; up to 128 it would be one lane)
        assert abi.split_plan(W, 270, 256, 5).refused == 1 and abi.split_plan(W, 270, 160, 5).refused == 0
        ctx.reserve(256)
        st = ctx.render_device(abi.make_params(W, H, spp=160, bounces=1, paths=1, samples=1, mode=0, flags=0, seed=SEED), want_stats=True)
        assert st.lanes_per_pixel > 1 and st.split_refused == 0 and ctx.alloc_count() == 0


# ---- bands gathered on the root GPU over RCCL -----------------------------

@pytest.mark.parametrize("W,H,ids,scene_n", [
    (64, 36, [0], 0),             # one device, one band: a 1-rank communicator, send to self
    (96, 60, [0, 0], 0),          # interleaved stripes -> staging buffer + de-interleave kernel
    (70, 53, [0, 0, 0], 0),       # ragged contiguous bands received straight into the frame
    (1920, 1080, [0] * 8, 0),     # config #3's frame cut for 8 GPUs (8.3 MB through the exchange)
    (48, 120, [0] * 8, 384),      # sphere-parallel kernel, stripes
])
def test_group_render_device_gathers_over_rccl(abi, W, H, ids, scene_n):
    """luma_hip_group_render_device: the members' bands reach the root GPU through ONE grouped
    ncclSend/ncclRecv exchange on a communicator over the group's distinct devices (here device 0
    alone, so every band is a send to self), into the caller's device frame or the group's own.
    Frame, accumulators and ray count equal the one-context luma_hip_render_device, two calls in a
    row (progressive accumulation), and gather_ms is reported."""
    kw = dict(spp=3, bounces=5, paths=1, samples=1, mode=1, flags=0, seed=SEED)
    scene = abi.synthetic_scene(scene_n) if scene_n else abi.default_scene()
    params = abi.make_params(W, H, **kw)
    with abi.Context(0) as ctx, abi.Group(ids) as g:
        ctx.set_scene(scene); ctx.set_camera(abi.default_camera(W, H)); ctx.set_frame(W, H)
        g.set_scene(scene); g.set_camera(abi.default_camera(W, H)); g.set_frame(W, H)
        with abi.Context(0) as holder:                      # a caller-owned device frame: another context's buffer
            holder.set_scene(scene); holder.set_camera(abi.default_camera(W, H)); holder.set_frame(W, H)
            for call in range(2):
                wst = ctx.render_device(params, want_stats=True)
                want, wacc = ctx.read_target(), ctx.read_accum()
                if call == 0:
                    st = g.render_device(params)            # the group's own frame
                    got = g.read_frame()
                    assert g.frame_device()
                else:
                    st = g.render_device(params, holder.target_device())
                    got = holder.read_target()
                assert_frames_equal(got, want, f"RCCL-gathered frame, {len(ids)} band(s), call {call}")
                assert np.array_equal(g.read_accum().view(np.uint32), wacc.view(np.uint32))
                assert st.rays == wst.rays and st.kernel_ms > 0 and st.gather_ms > 0
        with pytest.raises(abi.LumaError) as e:             # host memory is not a device frame
            g.render_device(params, np.zeros((H, W), np.uint32).ctypes.data)
        assert e.value.code == abi.E_ARG


def test_render_frame_scatters_bands_into_one_frame(abi):
    """luma_hip_render_frame: each context writes only its own rows of the shared host frame."""
    W, H, n = 80, 90, 3
    kw = dict(spp=2, bounces=5, paths=1, samples=1, mode=1, flags=0, seed=SEED)
    want, _, _ = hip_render(abi, abi.default_scene(), abi.default_camera(W, H), W, H, **kw)
    for striped in (True, False):
        frame = np.full((H, W), 0xDEADBEEF, dtype=np.uint32)
        for rnk in range(n):
            with abi.Context(0) as ctx:
                ctx.set_scene(abi.default_scene())
                ctx.set_camera(abi.default_camera(W, H))
                if striped:
                    ctx.set_frame_striped(W, H, 15, rnk, n)
                else:
                    ctx.set_frame(W, H, rnk * H // n, (rnk + 1) * H // n)
                before = frame.copy()
                ctx.render_frame(abi.make_params(W, H, **kw), frame)
                touched = (frame != before).any(axis=1).sum()
                assert touched <= H // n
        assert_frames_equal(frame, want, f"render_frame striped={striped}")


def test_contexts_driven_from_their_own_threads(abi):
    """The thread-per-GPU host shape of luma_hip_render_frame: four contexts (here on one device),
    each driven by its own host thread, fill one frame at the same time, several frames in a row."""
    import threading
    W, H, n = 96, 64, 4
    kw = dict(spp=3, bounces=5, paths=1, samples=1, mode=1, flags=0, seed=SEED)
    params = abi.make_params(W, H, **kw)
    frames = []
    with abi.Context(0) as ctx:
        ctx.set_scene(abi.default_scene())
        ctx.set_camera(abi.default_camera(W, H))
        ctx.set_frame(W, H)
        for _ in range(3):
            frames.append(ctx.render(params)[0])
    frame = np.zeros((H, W), dtype=np.uint32)
    got, errors = [], []
    start = threading.Barrier(n + 1)
    step = threading.Barrier(n + 1)

    def worker(rank):
        try:
            with abi.Context(0) as c:
                c.set_scene(abi.default_scene())
                c.set_camera(abi.default_camera(W, H))
                c.set_frame(W, H, rank * H // n, (rank + 1) * H // n)
                start.wait()
                for _ in range(3):
                    c.render_frame(params, frame)
                    step.wait()          # frame complete: the main thread copies it
                    step.wait()
        except Exception as e:           # surfaces in the main thread
            errors.append(e)
            start.abort(); step.abort()

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(n)]
    for t in threads:
        t.start()
    try:
        start.wait()
        for _ in range(3):
            step.wait()
            got.append(frame.copy())
            step.wait()
    except threading.BrokenBarrierError:
        pass
    for t in threads:
        t.join(timeout=60)
    assert not errors, errors
    for i in range(3):
        assert_frames_equal(got[i], frames[i], f"threaded frame {i}")


def test_page_locked_host_frame(abi):
    """luma_hip_host_alloc: the same pixels arrive in a page-locked frame as in a pageable one."""
    W, H = 128, 72
    kw = dict(spp=2, bounces=5, paths=1, samples=1, mode=1, flags=0, seed=SEED)
    want, _, _ = hip_render(abi, abi.default_scene(), abi.default_camera(W, H), W, H, **kw)
    with abi.HostFrame(H, W) as hf, abi.Group([0, 0]) as g:
        hf.array[:] = 0
        g.set_scene(abi.default_scene())
        g.set_camera(abi.default_camera(W, H))
        g.set_frame(W, H)
        g.render(abi.make_params(W, H, **kw), out=hf.array)
        assert_frames_equal(hf.array.copy(), want, "page-locked frame")
    with pytest.raises(abi.LumaError) as e:
        abi.HostFrame(0, 16)
    assert e.value.code == abi.E_ARG


@pytest.mark.parametrize("n,flags", [(5, 0), (5, 16), (700, 0), (3000, 32)])
def test_kernel_written_host_target(abi, n, flags):
    """luma_hip_render into a page-locked target: the kernel writes the host band itself (no copy after it).
    The band equals the one a pageable target gets, the context's own framebuffer holds it too, and a band
    that is not the whole frame lands in a page-locked buffer of just its size -- for the ray-per-lane, the
    sphere-parallel and the grid kernels."""
    W, H = 192, 108
    scene = abi.default_scene() if n == 5 else abi.synthetic_scene(n)
    p = abi.make_params(W, H, spp=3, bounces=5, paths=1, samples=1, mode=1, flags=flags, seed=SEED)
    for rows in ((0, H), (27, 81)):
        with abi.Context(0) as a, abi.Context(0) as b, abi.HostFrame(rows[1] - rows[0], W) as hf:
            for ctx in (a, b):
                ctx.set_scene(scene); ctx.set_camera(abi.default_camera(W, H)); ctx.set_frame(W, H, *rows)
            want, _ = a.render(p)                            # pageable numpy target: hipMemcpy after the kernel
            hf.array[:] = 0xDEADBEEF
            got, _ = b.render(p, out=hf.array)
            assert_frames_equal(got.copy(), want, f"kernel-written host target, rows {rows}")
            assert_frames_equal(b.read_target(), want, "the context's own framebuffer")
            got2, _ = b.render(p, out=hf.array)              # accumulates on: frames 4..6
            want2, _ = a.render(p)
            assert_frames_equal(got2.copy(), want2, "second call")


def test_kernel_written_host_frame(abi):
    """luma_hip_render_frame and luma_hip_group_render into a page-locked FRAME: every context's kernel writes its
    own rows of it (contiguous bands and interleaved stripes); same frame as through a pageable one."""
    W, H = 160, 120
    p = abi.make_params(W, H, spp=2, bounces=5, paths=1, samples=1, mode=1, flags=0, seed=SEED)
    want, _, _ = hip_render(abi, abi.default_scene(), abi.default_camera(W, H), W, H, spp=2, bounces=5, paths=1, samples=1, mode=1, flags=0, seed=SEED)
    with abi.HostFrame(H, W) as hf:
        for striped in (False, True):
            hf.array[:] = 0
            for r in range(4):
                with abi.Context(0) as ctx:
                    ctx.set_scene(abi.default_scene()); ctx.set_camera(abi.default_camera(W, H))
                    if striped: ctx.set_frame_striped(W, H, 15, r, 4)
                    else: ctx.set_frame(W, H, r * 30, (r + 1) * 30)
                    ctx.render_frame(p, hf.array)
            assert_frames_equal(hf.array.copy(), want, f"render_frame, striped={striped}")
        for ids in ([0], [0, 0, 0], [0] * 8):
            hf.array[:] = 0
            with abi.Group(ids) as g:
                g.set_scene(abi.default_scene()); g.set_camera(abi.default_camera(W, H)); g.set_frame(W, H)
                g.render(p, out=hf.array)
            assert_frames_equal(hf.array.copy(), want, f"group of {len(ids)}")


def test_group_errors(abi):
    with pytest.raises(abi.LumaError) as e:
        abi.Group([])
    assert e.value.code == abi.E_ARG
    with pytest.raises(abi.LumaError) as e:
        abi.Group([abi.device_count()])          # one past the last device
    assert e.value.code == abi.E_ARG
    with abi.Group([0, 0, 0]) as g:
        g.set_scene(abi.default_scene())
        g.set_camera(abi.default_camera(16, 2))
        with pytest.raises(abi.LumaError) as e:
            g.render(abi.make_params(16, 2))
        assert e.value.code == abi.E_STATE
        with pytest.raises(abi.LumaError) as e:
            g.set_frame(16, 2)                   # fewer rows than devices
        assert e.value.code == abi.E_ARG
        g.set_frame(16, 3)
        img, _ = g.render(abi.make_params(16, 3))
        assert img.shape == (3, 16)


def test_one_context_through_scene_camera_and_frame_changes(abi, oracle):
    """One context reused the way an interactive host would: scenes of every kernel class in turn
    (ray-per-lane from LDS, sphere-parallel with 4- and 16-wave workgroups, ray-per-lane from global
    memory), camera moves with and without a reset, frame-size changes, spp changes (the scratch
    planes of the frame-splitting kernel grow and are reused).  Every frame equals the oracle's."""
    with abi.Context(0) as ctx:
        W, H = 72, 40
        ctx.set_frame(W, H)
        cam_a, ocam_a = abi.default_camera(W, H), oracle.camera(W, H)
        ctx.set_camera(cam_a)
        ref = None
        steps = [("scene", 0), ("render", 3), ("scene", 400), ("render", 2), ("render", 5), ("scene", 5), ("render", 9),
                 ("move", (1.0, -0.5, 2.0, 0.1, -0.4)), ("render", 4), ("scene", 1500), ("render", 1),
                 ("frame", (33, 21)), ("render", 2), ("scene", 5200), ("render", 1), ("scene", 5), ("render", 17),
                 ("move_noreset", (0.0, 0.0, 1.0, 0.0, 0.2)), ("render", 2)]
        n, ocam = 0, ocam_a
        for what, arg in steps:
            if what == "scene":
                n = arg
                scene = abi.default_scene() if n == 5 else abi.synthetic_scene(n) if n else (abi.Sphere * 0)()
                oscene = oracle.default_scene() if n == 5 else oracle.synthetic_scene(n) if n else (oracle.Sphere * 0)()
                ctx.set_scene(scene)
                ctx.reset()
                ref = oracle.Renderer(oscene, ocam, W, H)
            elif what in ("move", "move_noreset"):
                px, py, pz, pitch, yaw = arg
                ctx.set_camera(abi.camera_from_pose(W, H, (px, py, pz), pitch, yaw))
                ocam = oracle.camera_pose(W, H, (px, py, pz), pitch, yaw)
                if what == "move":
                    ctx.reset()
                    ref = oracle.Renderer(oscene, ocam, W, H)
                else:                                   # the accumulation goes on under the new camera
                    ref.cam = ocam
            elif what == "frame":
                W, H = arg
                ctx.set_frame(W, H)
                ctx.set_camera(abi.default_camera(W, H))
                ocam = oracle.camera(W, H)
                ref = oracle.Renderer(oscene, ocam, W, H)
            else:
                kw = dict(spp=arg, bounces=5, paths=1, samples=1, mode=1, flags=0, seed=SEED)
                got, _ = ctx.render(abi.make_params(W, H, **kw))
                ref.render(oracle.make_params(W, H, **kw))
                assert_frames_equal(got, ref.target, f"after {what} {arg}, n={n}")
                assert np.array_equal(ctx.read_accum().view(np.uint32), ref.accum.view(np.uint32))
                assert ctx.frame_index() == ref.frame_index


# ---- device-resident target (what bench.py times) ------------------------------------

def test_render_device_matches_host_target(abi):
    W, H = 100, 60
    kw = dict(spp=2, bounces=5, paths=1, samples=1, mode=1, flags=0, seed=SEED)
    host, _, _ = hip_render(abi, abi.default_scene(), abi.default_camera(W, H), W, H, **kw)
    with abi.Context(0) as ctx:
        ctx.set_scene(abi.default_scene())
        ctx.set_camera(abi.default_camera(W, H))
        ctx.set_frame(W, H)
        st = ctx.render_device(abi.make_params(W, H, **kw), want_stats=True)
        assert st.rays > 0 and st.kernel_ms > 0
        ptr = ctx.target_device()
        assert ptr
        out = ctx.read_target(ptr)
        assert np.array_equal(out, ctx.read_target())
    assert_frames_equal(out, host, "device target")


def test_stats_priced_at_the_scene_the_rays_ran_against(abi):
    """A stats-less render, then luma_hip_set_scene, then a render with stats: the first call's rays
    are counted against the first scene's size (the loop at Renderer.cpp:242), not the second's."""
    W, H = 64, 32
    kw = dict(spp=1, bounces=1, paths=0, samples=1, mode=0, flags=abi.FLAG_NOISE_OFF, seed=SEED)
    with abi.Context(0) as ctx:
        ctx.set_scene(abi.default_scene())                # 5 spheres
        ctx.set_camera(abi.default_camera(W, H))
        ctx.set_frame(W, H)
        ctx.render_device(abi.make_params(W, H, **kw))    # no stats: W*H*2 rays against 5 spheres
        ctx.set_scene(abi.synthetic_scene(12))
        st = ctx.render_device(abi.make_params(W, H, **kw), want_stats=True)
    rays_each = 2 * W * H                                 # DOF ray + one bounce ray per pixel
    assert st.rays == 2 * rays_each
    assert st.sphere_tests == rays_each * 5 + rays_each * 12


# ---- error behaviour -----------------------------------------------------------------

def test_error_codes(abi):
    with abi.Context(0) as ctx:
        p = abi.make_params(32, 32)
        with pytest.raises(abi.LumaError) as e:
            ctx.render(p)
        assert e.value.code == abi.E_ARG or e.value.code == abi.E_STATE
        ctx.set_scene(abi.default_scene())
        ctx.set_camera(abi.default_camera(32, 32))
        with pytest.raises(abi.LumaError) as e:
            ctx.set_frame(32, 32, 8, 8)
        assert e.value.code == abi.E_ARG
        with pytest.raises(abi.LumaError) as e:
            ctx.set_frame(0, 32)
        assert e.value.code == abi.E_ARG
        ctx.set_frame(32, 32)
        for bad in (abi.make_params(32, 32, spp=0), abi.make_params(16, 32), abi.make_params(32, 32, mode=5),
                    abi.make_params(32, 32, mode=1, samples=0)):
            with pytest.raises(abi.LumaError) as e:
                ctx.render(bad)
            assert e.value.code == abi.E_ARG
        img, _ = ctx.render(p)          # the context still works after rejected calls
        assert img.shape == (32, 32)
