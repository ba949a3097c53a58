"""Randomised parity: seeded random scenes, cameras and options through the C-ABI against the
oracle, bit for bit (frame and accumulation buffer).  The scenes are built to reach the corners
the default scene does not: duplicate spheres (exact ties: the lower index wins,
Renderer.cpp:266), concentric and overlapping spheres, the camera inside a sphere (only the far
root exists, which trace_ray ignores, :255-258), zero and tiny radii, metallic == 0 (the path ends
at the first hit, :353-357), albedo == 0 (no indirect rays, :337), large roughness, every flag
combination, paths 0..3, bounces 1..8, and scene sizes on both sides of every kernel hand-over."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def random_scene(rng, n, kind):
    out = []
    for i in range(n):
        if kind == "cluster":          # overlapping spheres in front of the camera
            pos = rng.uniform([-4, -3, -14], [4, 3, -4])
            radius = rng.choice([0.0, 1e-6, 0.05, 0.5, 1.0, 2.5])
        elif kind == "around":         # all around the camera, which sits inside some of them
            pos = rng.uniform(-6, 6, size=3)
            radius = rng.uniform(0.1, 7.0)
        elif kind == "mirror":         # +-x pairs: rays in the x = 0 plane hit both at the same t (ties)
            pos = rng.uniform([0.5, -2, -12], [3, 2, -5])
            radius = rng.choice([0.5, 1.0, 1.5])
            if i % 2 and out:
                prev = out[-1]
                pos = prev[0] * np.array([-1, 1, 1], np.float32)
                radius = prev[1]
        else:                          # "wide": many small ones, far away, plus a ground
            pos = rng.uniform([-40, -20, -90], [40, 1.5, -10])
            radius = rng.uniform(0.2, 0.8)
        metallic = float(rng.choice([0.0, 0.001, 0.1, 0.4, 1.0]))
        albedo = float(rng.choice([0.0, 0.5, 1.0]))
        roughness = float(rng.choice([0.0, 0.4, 0.5, 3.0]))
        out.append((pos.astype(np.float32), np.float32(radius), rng.uniform(0, 1, size=3).astype(np.float32),
                    albedo, metallic, roughness))
    if kind == "wide" and n:
        out[0] = (np.array([0, 102, 0], np.float32), np.float32(100), np.array([.6, .6, .6], np.float32), 1.0, .1, .5)
    # exact duplicates and concentric pairs: ties and nested hits
    for _ in range(0 if kind == "mirror" else min(3, n // 2)):
        a, b = rng.integers(0, n, size=2)
        pos, radius, *rest = out[a]
        out[b] = (pos.copy(), radius if rng.random() < 0.5 else np.float32(radius * 0.5), *out[b][2:])
    return out


def rescale(spheres, scale):
    """The same scene at another length scale (millimetres to kilometres): exercises the range
    guards of the lean sqrt and of the near-tie test."""
    return [(pos * np.float32(scale), np.float32(radius * np.float32(scale)), *rest) for pos, radius, *rest in spheres]


def fill(struct_array, spheres, abi_layout):
    for i, (pos, radius, diffuse, albedo, metallic, roughness) in enumerate(spheres):
        s = struct_array[i]
        if abi_layout:
            s.cx, s.cy, s.cz = map(float, pos)
            s.dr, s.dg, s.db = map(float, diffuse)
        else:
            s.pos[:] = [float(v) for v in pos]
            s.diffuse[:] = [float(v) for v in diffuse]
        s.radius, s.albedo, s.metallic, s.roughness = float(radius), albedo, metallic, roughness
    return struct_array


CASES = []
_rng = np.random.default_rng(20240611)
for case in range(96):
    kind = ["cluster", "around", "wide", "mirror"][case % 4]
    n = int(_rng.choice([0, 1, 2, 3, 7, 16, 33, 64]) if case < 80 else _rng.choice([319, 320, 400, 700, 1100]))
    scale = float(_rng.choice([1.0, 1.0, 1.0, 1e-3, 1e3, 3e4]))
    CASES.append(dict(
        id=f"{case}-{kind}-n{n}-x{scale:g}", kind=kind, n=n, scale=scale, seed=int(_rng.integers(1, 2**31)),
        W=int(_rng.integers(1, 40)), H=int(_rng.integers(1, 24)),
        spp=int(_rng.integers(1, 6)), bounces=int(_rng.integers(1, 9)), paths=int(_rng.integers(0, 4)),
        samples=int(_rng.integers(1, 4)), mode=int(_rng.integers(0, 2)), flags=int(_rng.integers(0, 32)),
        rng_seed=int(_rng.integers(0, 2**63)), posed=bool(_rng.random() < 0.6),
        pos=[float(v) for v in _rng.uniform(-3, 3, size=3)], pitch=float(_rng.uniform(-0.7, 0.7)),
        yaw=float(_rng.uniform(-3.1, 3.1)), calls=int(_rng.integers(1, 3))))
# larger frames of the tie-prone kinds, half of them noise free (rays of a whole column share dir.x == 0)
for case in range(12):
    kind = ["mirror", "cluster", "around"][case % 3]
    n = int(_rng.choice([2, 6, 16, 40]))
    CASES.append(dict(
        id=f"big{case}-{kind}-n{n}", kind=kind, n=n, scale=1.0, seed=int(_rng.integers(1, 2**31)),
        W=int(_rng.choice([160, 161, 192])), H=int(_rng.choice([90, 91])), spp=3, bounces=int(_rng.integers(1, 7)),
        paths=int(_rng.integers(0, 3)), samples=1, mode=1, flags=int(_rng.choice([0, 1, 2, 3])),
        rng_seed=int(_rng.integers(0, 2**63)), posed=False, pos=[0.0, 0.0, 0.0], pitch=0.0, yaw=0.0, calls=1))


# every sphere count that has an instance with the count as a compile-time constant (1..8), with and
# without noise, one and several lanes per pixel (spp 1 forces one lane, larger spp on these small
# frames takes four)
for n in range(1, 9):
    for flags in (0, 1):
        kind = ["cluster", "around", "mirror", "wide"][n % 4]
        CASES.append(dict(
            id=f"fixed{n}-{kind}-f{flags}", kind=kind, n=n, scale=1.0, seed=int(_rng.integers(1, 2**31)),
            W=int(_rng.integers(20, 70)), H=int(_rng.integers(10, 40)), spp=int(_rng.choice([1, 7])),
            bounces=int(_rng.integers(1, 7)), paths=int(_rng.integers(0, 3)), samples=1, mode=1, flags=flags,
            rng_seed=int(_rng.integers(0, 2**63)), posed=bool(n % 2), pos=[float(v) for v in _rng.uniform(-2, 2, size=3)],
            pitch=float(_rng.uniform(-0.5, 0.5)), yaw=float(_rng.uniform(-1, 1)), calls=2))


# the same sphere counts at 8 spp and more: the last rows of these launches form the fine class (eight lanes per
# pixel, two of each pixel side by side in a wave; luma_kernels.hip plan_split, HALF), frames accumulating over two calls
for n in range(1, 9):
    kind = ["mirror", "cluster", "wide", "around"][n % 4]
    CASES.append(dict(
        id=f"fine{n}-{kind}", kind=kind, n=n, scale=1.0, seed=int(_rng.integers(1, 2**31)),
        W=int(_rng.integers(20, 70)), H=int(_rng.integers(21, 46)), spp=int(_rng.choice([8, 9, 13, 20])),
        bounces=int(_rng.integers(1, 7)), paths=int(_rng.integers(0, 3)), samples=1, mode=1, flags=int(n % 2),
        rng_seed=int(_rng.integers(0, 2**63)), posed=bool(n % 3 == 0), pos=[float(v) for v in _rng.uniform(-2, 2, size=3)],
        pitch=float(_rng.uniform(-0.5, 0.5)), yaw=float(_rng.uniform(-1, 1)), calls=2))


# one lane per pixel with frames taken inside the waves (TAKES instances: 2 to 128 frames per call, 1 to 16 spheres, noise
# on): small frames are made such bands with luma_hip_set_park_limit (0: no fine class; some MiB: a partial or whole one);
# scales that leave the lean sqrt / division windows, tie-prone kinds, hit-point origins, two accumulating calls
for case in range(24):
    kind = ["mirror", "cluster", "wide", "around"][case % 4]
    n = int(_rng.choice([1, 2, 5, 8, 9, 16, 24, 40]))
    CASES.append(dict(
        id=f"takes{case}-{kind}-n{n}", kind=kind, n=n, scale=float(_rng.choice([1.0, 1.0, 1e-3, 3e4])), seed=int(_rng.integers(1, 2**31)),
        W=int(_rng.integers(30, 200)), H=int(_rng.integers(3, 60)), spp=int(_rng.choice([2, 3, 5, 8, 9, 17, 40, 64, 128])),
        bounces=int(_rng.integers(1, 7)), paths=int(_rng.integers(0, 3)), samples=int(_rng.integers(1, 3)), mode=int(case % 5 != 0),
        flags=int(_rng.choice([0, 0, 2, 4, 8, 2 | 16])), park_limit=int(_rng.choice([0, 0, 1 << 20, 64 << 20])),
        rng_seed=int(_rng.integers(0, 2**63)), posed=bool(case % 3 == 0), pos=[float(v) for v in _rng.uniform(-2, 2, size=3)],
        pitch=float(_rng.uniform(-0.5, 0.5)), yaw=float(_rng.uniform(-1, 1)), calls=2))


@pytest.mark.parametrize("c", CASES, ids=[c["id"] for c in CASES])
def test_random_scene_matches_oracle(abi, oracle, c):
    rng = np.random.default_rng(c["seed"])
    spheres = rescale(random_scene(rng, c["n"], c["kind"]), c["scale"])
    a_scene = fill((abi.Sphere * c["n"])(), spheres, True)
    o_scene = fill((oracle.Sphere * c["n"])(), spheres, False)
    assert bytes(a_scene) == bytes(o_scene)
    W, H = c["W"], c["H"]
    kw = dict(spp=c["spp"], bounces=c["bounces"], paths=c["paths"], samples=c["samples"], mode=c["mode"],
              flags=c["flags"], seed=c["rng_seed"])
    if c["posed"]:
        pos = [v * c["scale"] for v in c["pos"]]
        a_cam = abi.camera_from_pose(W, H, pos, c["pitch"], c["yaw"])
        o_cam = oracle.camera_pose(W, H, pos, c["pitch"], c["yaw"])
    else:
        a_cam, o_cam = abi.default_camera(W, H), oracle.camera(W, H)
    assert bytes(a_cam) == bytes(o_cam)

    ref = oracle.Renderer(o_scene, o_cam, W, H)
    with abi.Context(0) as ctx:
        ctx.set_scene(a_scene)
        ctx.set_camera(a_cam)
        ctx.set_frame(W, H)
        if "park_limit" in c:
            ctx.set_park_limit(c["park_limit"])
        for call in range(c["calls"]):                      # the second call accumulates on the first
            got, st = ctx.render(abi.make_params(W, H, **kw))
            assert "park_limit" not in c or st.lanes_per_pixel == 1 or c["park_limit"] > 0
            ref.render(oracle.make_params(W, H, **kw))
            acc = ctx.read_accum()
            diff = int((got != ref.target).sum())
            assert diff == 0, f"{c['id']} call {call}: {diff} of {W * H} pixels differ"
            assert np.array_equal(acc.view(np.uint32), ref.accum.view(np.uint32)), f"{c['id']} call {call}: accumulators"
        # the kernels trace the DOF ray once per call, the reference once per frame
        assert st.sphere_tests == st.rays * c["n"]

        # the same calls with the uniform grid choosing the spheres to test (LUMA_FLAG_GRID)
        ctx.reset()
        ref = oracle.Renderer(o_scene, o_cam, W, H)
        for call in range(c["calls"]):
            got, gst = ctx.render(abi.make_params(W, H, **dict(kw, flags=kw["flags"] | abi.FLAG_GRID)))
            ref.render(oracle.make_params(W, H, **kw))
            diff = int((got != ref.target).sum())
            assert diff == 0, f"{c['id']} grid, call {call}: {diff} of {W * H} pixels differ"
            assert np.array_equal(ctx.read_accum().view(np.uint32), ref.accum.view(np.uint32)), f"{c['id']} grid: accumulators"
        assert gst.rays == st.rays          # (a sphere listed in several cells may be tested more than once)


# Degenerate caller matrices: view_inverse scaled so that |dir| is ~1e-17 .. 1e15 and
# a = dot(dir, dir) leaves the window in which the ray-per-lane scan orders candidates by the
# numerator alone (luma_kernels.hip scan_spheres: outside it both quotients are formed, as the
# reference does at Renderer.cpp:255-266).  The linear, sphere-parallel and grid scans must all
# still give the oracle's frame.
@pytest.mark.parametrize("n", [7, 400])
@pytest.mark.parametrize("dir_scale", [1e-17, 1e-10, 1e-7, 1e7, 1e10, 1e15])
def test_extreme_direction_scale_all_scan_paths(abi, oracle, n, dir_scale):
    rng = np.random.default_rng(1234 + n)
    spheres = random_scene(rng, n, "cluster")
    a_scene = fill((abi.Sphere * n)(), spheres, True)
    o_scene = fill((oracle.Sphere * n)(), spheres, False)
    W, H = 48, 20
    a_cam, o_cam = abi.default_camera(W, H), oracle.camera(W, H)
    for i in range(16):
        a_cam.view_inv[i] = np.float32(a_cam.view_inv[i] * np.float32(dir_scale))
        o_cam.view_inv[i] = a_cam.view_inv[i]
    assert bytes(a_cam) == bytes(o_cam)
    # noise off keeps |dir| at the scale under test (the +-blur noise would swamp a tiny dir)
    for flags in (abi.FLAG_NOISE_OFF, 0):
        kw = dict(spp=2, bounces=4, paths=1, samples=1, mode=1, flags=flags, seed=99)
        ref = oracle.Renderer(o_scene, o_cam, W, H)
        ref.render(oracle.make_params(W, H, **kw))
        with abi.Context(0) as ctx:
            ctx.set_scene(a_scene)
            ctx.set_camera(a_cam)
            ctx.set_frame(W, H)
            for extra in (0, abi.FLAG_GRID):
                ctx.reset()
                got, _ = ctx.render(abi.make_params(W, H, **dict(kw, flags=flags | extra)))
                diff = int((got != ref.target).sum())
                assert diff == 0, f"n={n} scale={dir_scale:g} flags={flags | extra}: {diff} pixels differ"
                assert np.array_equal(ctx.read_accum().view(np.uint32), ref.accum.view(np.uint32))
