"""CPU tests that pin the oracle: the reference's own fixtures (output.ppm, luma.ppm layout),
the committed golden vectors, and independent restatements of its primitives in numpy."""
import ctypes as C
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SEED = 0x4C554D41


def f32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


# ---- the reference's own fixtures ----------------------------------------------------

def test_output_ppm_tonemap_pack_kat(oracle, tmp_path):
    """/root/reference/output.ppm (committed copy: golden/ref_output.ppm): 50x50, every pixel
    '115 128 156' = RGB(tonemap((.6,.7,.95))) -- Renderer.cpp:57-70, :44-55, :174."""
    L = oracle.lib()
    c = np.array([.6, .7, .95], dtype=np.float32)
    t = np.zeros(3, dtype=np.float32)
    L.lo_tonemap(f32p(c), f32p(t))
    px = L.lo_pack_rgb(f32p(t), 0)
    assert ((px >> 16) & 255, (px >> 8) & 255, px & 255) == (115, 128, 156)
    frame = np.full((50, 50), px, dtype=np.uint32)
    out = tmp_path / "o.ppm"
    assert oracle.export_ppm(frame, str(out)) == 0
    want = open(os.path.join(GOLD, "ref_output.ppm"), "rb").read().replace(b"\r\n", b"\n")
    assert out.read_bytes() == want


def test_pack_variants(oracle):
    """::RGB (Renderer.cpp:44-55) truncates, shifts and ORs without clamping, so a channel above
    1.0 bleeds into its neighbour; CLAMP_PACK and PACK_RGBA8 are build-defined variants."""
    L = oracle.lib()
    c = np.array([1.1, 0.5, 0.25], dtype=np.float32)          # 280, 127, 63
    assert L.lo_pack_rgb(f32p(c), 0) == ((280 << 16) | (127 << 8) | 63)
    assert L.lo_pack_rgb(f32p(c), 4) == ((255 << 16) | (127 << 8) | 63)
    assert L.lo_pack_rgb(f32p(c), 4 | 16) == (0xFF000000 | (63 << 16) | (127 << 8) | 255)
    assert L.lo_pack_rgb(f32p(c), 16) == (0xFF000000 | (63 << 16) | (127 << 8) | 280)


def test_luma_ppm_layout(oracle, tmp_path):
    """/root/reference/luma.ppm pins export_to_ppm's column-major emission (image.cpp:35-39):
    W text lines of H triples under a 'W H' header."""
    facts = json.load(open(os.path.join(GOLD, "ref_luma_ppm_layout.json")))
    W, H = 256, 240
    frame = (np.arange(W * H, dtype=np.uint32).reshape(H, W) * 2654435761 & 0xFFFFFF).astype(np.uint32)
    out = tmp_path / "l.ppm"
    assert oracle.export_ppm(frame, str(out)) == 0
    lines = out.read_text().split("\n")
    assert lines[:3] == facts["header"] == ["P3", f"{W} {H}", "255"]
    body = [ln for ln in lines[3:] if ln]
    assert len(body) == facts["body_lines"] == W
    assert {len(ln.split()) // 3 for ln in body} == set(facts["triples_per_line"]) == {H}
    # x outer, y inner: line x, entry y is pixel (x, y)
    x, y = 17, 201
    tok = body[x].split()[3 * y:3 * y + 3]
    d = int(frame[y, x])
    assert [int(v) for v in tok] == [(d >> 16) & 255, (d >> 8) & 255, d & 255]
    assert body[0].endswith("  ")


def test_bmp_writer(oracle, tmp_path):
    """Image::export_to_bmp, image.cpp:66-153: 54-byte header, BGR24, bottom-up, 4-byte row padding."""
    W, H = 5, 3
    frame = np.array([[0x010203 + 0x101010 * (y * W + x) for x in range(W)] for y in range(H)], dtype=np.uint32)
    out = tmp_path / "o.bmp"
    assert oracle.export_bmp(frame, str(out)) == 0
    b = out.read_bytes()
    pad = (4 - (W * 3) % 4) % 4
    assert len(b) == 54 + (W * 3 + pad) * H
    assert b[:2] == b"BM" and int.from_bytes(b[2:6], "little") == len(b) and int.from_bytes(b[10:14], "little") == 54
    assert int.from_bytes(b[14:18], "little") == 40
    assert int.from_bytes(b[18:22], "little") == W and int.from_bytes(b[22:26], "little") == H
    assert b[26:30] == bytes([1, 0, 24, 0])
    first = b[54:57]          # bottom row first, x = 0, stored B G R
    d = int(frame[H - 1, 0])
    assert list(first) == [d & 255, (d >> 8) & 255, (d >> 16) & 255]


# ---- committed golden vectors (oracle drift) -----------------------------------------

def test_oracle_matches_golden_frames(oracle):
    g = np.load(os.path.join(GOLD, "oracle_frames.npz"))
    cases = {
        "primary_64x36": dict(W=64, H=36, spp=1, bounces=1, mode=0, flags=oracle.FLAG_NOISE_OFF),
        "primary_160x90": dict(W=160, H=90, spp=1, bounces=1, mode=0, flags=oracle.FLAG_NOISE_OFF),
        "direct_noise_64x36": dict(W=64, H=36, spp=1, bounces=1, mode=0, flags=0),
        "path_64x36_4spp": dict(W=64, H=36, spp=4, bounces=5, mode=1, flags=0),
        "path_160x90_4spp": dict(W=160, H=90, spp=4, bounces=5, mode=1, flags=0),
        "path_fromhit_64x36": dict(W=64, H=36, spp=2, bounces=5, mode=1, flags=oracle.FLAG_ORIGIN_FROM_HIT, paths=2),
    }
    for name, c in cases.items():
        W, H = c.pop("W"), c.pop("H")
        t, a, st = oracle.render(oracle.default_scene(), oracle.camera(W, H), oracle.make_params(W, H, seed=SEED, **c))
        assert np.array_equal(t, g[name + "_target"]), name
        assert np.array_equal(a.view(np.uint32), g[name + "_accum"].view(np.uint32)), name
        assert [st.rays, st.sphere_tests] == list(g[name + "_rays"]), name
    t, a, _ = oracle.render(oracle.synthetic_scene(300), oracle.camera(48, 27),
                            oracle.make_params(48, 27, spp=2, bounces=5, mode=1, seed=SEED))
    assert np.array_equal(t, g["synthetic300_48x27_target"])


def test_oracle_matches_golden_kats(oracle):
    import hashlib
    L = oracle.lib()
    k = np.load(os.path.join(GOLD, "oracle_kats.npz"))
    y = np.array([L.lo_expf(float(x)) for x in k["expf_x"]], dtype=np.float32)
    assert np.array_equal(y.view(np.uint32), k["expf_y"].view(np.uint32))
    for (x, yy, z), want in zip(k["pcg3d_in"], k["pcg3d_out"]):
        o = (C.c_uint32 * 3)()
        L.lo_pcg3d(int(x), int(yy), int(z), o)
        assert list(o) == list(want)
    keys = (C.c_uint32 * 3)()
    L.lo_seed_keys(SEED, keys)
    assert list(keys) == list(k["seed_keys"])
    cam = oracle.camera(1920, 1080)
    assert np.array_equal(np.array(cam.proj_inv[:], dtype=np.float32), k["cam1080_proj_inv"])
    assert np.array_equal(np.array(cam.view_inv[:], dtype=np.float32), k["cam1080_view_inv"])
    assert np.array_equal(oracle.camera_rays(oracle.camera(64, 36), 64, 36), k["rays_64x36"])
    sc = oracle.synthetic_scene(4096)
    assert hashlib.sha256(bytes(sc)).digest() == bytes(k["synthetic4096_sha256"])


# ---- independent restatements --------------------------------------------------------

def pcg3d_np(x, y, z):
    M = np.uint64(0xFFFFFFFF)
    x, y, z = (np.uint64(v) for v in (x, y, z))
    lcg = lambda v: (v * np.uint64(1664525) + np.uint64(1013904223)) & M
    x, y, z = lcg(x), lcg(y), lcg(z)
    x = (x + y * z) & M; y = (y + z * x) & M; z = (z + x * y) & M
    x ^= x >> np.uint64(16); y ^= y >> np.uint64(16); z ^= z >> np.uint64(16)
    x = (x + y * z) & M; y = (y + z * x) & M; z = (z + x * y) & M
    return int(x), int(y), int(z)


def test_pcg3d_against_numpy_restatement(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(1)
    for x, y, z in rng.integers(0, 2 ** 32, size=(200, 3), dtype=np.uint64):
        o = (C.c_uint32 * 3)()
        L.lo_pcg3d(int(x), int(y), int(z), o)
        assert tuple(o) == pcg3d_np(x, y, z)


def test_rng_uniformity(oracle):
    """Sanity of the counter RNG as the render loop indexes it: consecutive pixels / samples /
    slots give uniform, uncorrelated draws in [lo, hi)."""
    L = oracle.lib()
    keys = (C.c_uint32 * 3)()
    L.lo_seed_keys(SEED, keys)
    n = 20000
    v = np.zeros((n, 3), dtype=np.float32)
    o = (C.c_float * 3)()
    for i in range(n):
        L.lo_rng_vec3(keys, i % 1920 + 1920 * (i // 4000), (i // 1920) % 64, i % 7, -1.0, 1.0, o)
        v[i] = list(o)
    assert v.min() >= -1.0 and v.max() < 1.0
    assert abs(v.mean()) < 0.02 and abs(v.var() - 1 / 3) < 0.02
    hist, _ = np.histogram(v.ravel(), bins=16, range=(-1, 1))
    expected = v.size / 16
    chi2 = ((hist - expected) ** 2 / expected).sum()
    assert chi2 < 45            # 15 dof, p ~ 1e-4
    c = np.corrcoef(v[:-1, 0], v[1:, 0])[0, 1]
    assert abs(c) < 0.03
    assert abs(np.corrcoef(v[:, 0], v[:, 1])[0, 1]) < 0.03


def test_expf_accuracy(oracle):
    """lo_expf stands in for std::exp (Renderer.cpp:61, :427): <= 2 ulp on [-87, 0]."""
    L = oracle.lib()
    xs = np.concatenate([np.linspace(-87, 0, 20001), -np.logspace(-10, 1.9, 2000)]).astype(np.float32)
    got = np.array([L.lo_expf(float(x)) for x in xs], dtype=np.float32)
    want = np.exp(xs.astype(np.float64))
    ulp = np.spacing(want.astype(np.float32)).astype(np.float64)
    assert np.max(np.abs(got.astype(np.float64) - want) / ulp) <= 2.0
    assert L.lo_expf(-0.0) == 1.0 and L.lo_expf(-100.0) == 0.0 and L.lo_expf(float("-inf")) == 0.0


def test_focus_float_restatement_vs_double(oracle):
    """The reference evaluates focus/blur in double (Renderer.cpp:427-432); the contract uses
    float.  Bound the deviation: |blur_f32 - blur_f64| <= 4e-6, and the focus < .5 decision
    (Renderer.cpp:450) agrees except within 1e-6 of the threshold |dd| = 0.964."""
    L = oracle.lib()
    depths = np.concatenate([np.linspace(7.5, 12.5, 50001), [3.4028235e38, 0.1, 50.0]]).astype(np.float32)
    b = C.c_float()
    worst = 0.0
    for d in depths:
        f = L.lo_focus(float(d), 10.0, C.byref(b))
        dd = float(np.float32(d) - np.float32(10.0))
        with np.errstate(over="ignore"):
            f64 = 1.0 - np.exp(-np.float64(dd) ** 10)
        b64 = 0.5 + f64 * 4.5
        worst = max(worst, abs(b.value - b64))
        if abs(abs(dd) - 0.964) > 1e-4:
            assert (f < 0.5) == (f64 < 0.5)
    assert worst <= 4e-6


def test_camera_closed_form(oracle):
    """SURVEY section 7 cross-check: with the default pose the ray table is
    normalize(cx*tan(fov/2)*W/H, cy*tan(fov/2), -1), cx = 2x/W-1, cy = 2y/H-1."""
    W, H = 160, 90
    rays = oracle.camera_rays(oracle.camera(W, H), W, H).astype(np.float64)
    th = np.tan(np.radians(70.0) / 2)
    xs = 2 * np.arange(W) / W - 1
    ys = 2 * np.arange(H) / H - 1
    want = np.stack(np.broadcast_arrays(xs[None, :] * th * W / H, ys[:, None] * th, -np.ones((H, W))), axis=-1)
    want /= np.linalg.norm(want, axis=-1, keepdims=True)
    assert np.max(np.abs(rays - want)) < 2e-6


def test_mat4_inverse_roundtrip(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(3)
    for _ in range(50):
        m = (rng.standard_normal((4, 4)) + 2 * np.eye(4)).astype(np.float32)
        inv = np.zeros((4, 4), dtype=np.float32)
        L.lo_mat4_inverse(f32p(m), f32p(inv))
        assert np.allclose(m.astype(np.float64) @ inv.astype(np.float64), np.eye(4), atol=2e-4)


def test_default_scene_values(oracle):
    """renderer.h:62-70, order s, a, q, r, t."""
    sc = oracle.default_scene()
    rows = [(list(s.pos), s.radius, list(s.diffuse), s.albedo, s.metallic, s.roughness) for s in sc]
    f = np.float32
    assert rows[0] == ([0, f(.5), -10], 1, [0, 0, 1], 1, f(.001), f(.4))
    assert rows[1] == ([0, 102, 0], 100, [f(.6)] * 3, 1, f(.1), f(.5))
    assert rows[2][0] == [3, f(.5), -10] and rows[2][2] == [0, 1, 0]
    assert rows[3][0] == [6, f(.5), -10] and rows[3][2] == [1, 0, 0]
    assert rows[4] == ([3, f(.5), -23], 10, [1, 1, 1], 1, f(.4), 0)


def test_render_semantics_small(oracle):
    """Hand-checkable facts of the loop: an empty scene is the sky gradient of Renderer.cpp:308-316
    (tonemapped, x1.2 never applies because a miss has focus 1), bands leave other rows untouched,
    and frame-by-frame accumulation equals one multi-frame call."""
    W, H = 8, 6
    cam = oracle.camera(W, H)
    p = oracle.make_params(W, H, spp=1, bounces=1, mode=0, flags=oracle.FLAG_NOISE_OFF)
    t, a, st = oracle.render((oracle.Sphere * 0)(), cam, p)
    rays = oracle.camera_rays(cam, W, H)
    adj = (np.clip(rays[..., 1], -1, 1) + 1) * .5
    top, bot = np.array([.529, .808, .922]), np.array([.106, .275, .711])
    sky = top * (1 - adj[..., None]) + bot * adj[..., None]
    want = 1 - np.exp(-sky)
    assert np.allclose(a, want, atol=1e-6)
    assert st.rays == 2 * W * H and st.sphere_tests == 0

    r = oracle.Renderer(oracle.default_scene(), cam, W, H)
    r.render(p, rows=(2, 4))
    assert not r.target[:2].any() and not r.target[4:].any() and r.target[2:4].any()

    kw = dict(bounces=5, paths=1, samples=1, mode=1, flags=0, seed=9)
    one, acc1, _ = oracle.render(oracle.default_scene(), cam, oracle.make_params(W, H, spp=5, **kw))
    r = oracle.Renderer(oracle.default_scene(), cam, W, H)
    for _ in range(5):
        r.render(oracle.make_params(W, H, spp=1, **kw))
    assert np.array_equal(r.target, one) and np.array_equal(r.accum, acc1)
