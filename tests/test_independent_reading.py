"""The oracle against a SECOND reading of the reference's shading code (tests/reading_f64.py: numpy float64, written
from Renderer.cpp / Camera.cpp directly, sharing nothing with oracle/ but the counter RNG's draws).  Oracle and HIP
kernels come from one reading of Renderer.cpp:234-364; the reference holds no vector for any of it, so this is the
only independent weight on trace_ray / render_pixel / indirect_illumination / reflect_intersection (VERDICT r2,
"missing 3").  Two readings in different precisions agree except where a discontinuity (d > 0, t > 0, cos_theta >= 0,
focus < .5, the int cast of the pack) is decided the other way by a rounding: those pixels are counted and bounded.
CPU only."""
import ctypes as C

import numpy as np
import pytest

import reading_f64 as rd

SEED = 0x4C554D41


def seed_keys(oracle):
    keys = (C.c_uint32 * 3)()
    oracle.lib().lo_seed_keys(SEED, keys)
    return [int(k) for k in keys]


def channels(target):
    return np.stack([(target >> 16) & 0xFF, (target >> 8) & 0xFF, target & 0xFF], axis=-1).astype(np.int64)


def test_the_reading_draws_what_the_oracle_draws(oracle):
    """The one shared input: uniforms of the counter RNG.  The numpy pcg3d + [lo, hi] map here against lo_rng_vec3."""
    L, keys = oracle.lib(), seed_keys(oracle)
    ck = (C.c_uint32 * 3)(*keys)
    out = (C.c_float * 3)()
    rng = np.random.default_rng(3)
    for pixel, sample, slot in rng.integers(0, 2 ** 31, size=(200, 3)):
        lo, hi = -2.5, 2.5
        L.lo_rng_vec3(ck, int(pixel), int(sample), int(slot), lo, hi, out)
        mine = rd.random_vec3(keys, np.array([pixel], dtype=np.uint32), int(sample), np.array([slot], dtype=np.uint32), lo, hi)[0]
        assert np.allclose(np.array(out[:], dtype=np.float64), mine, rtol=0, atol=4e-7)


def test_ray_table_reading(oracle):
    """Camera::recompute_rays read as glm perspectiveFov / lookAt / inverse in double against the oracle's float table."""
    for W, H in ((64, 36), (160, 90)):
        mine = rd.camera_rays(W, H)
        theirs = oracle.camera_rays(oracle.camera(W, H), W, H).astype(np.float64)
        assert np.abs(mine - theirs).max() < 5e-7
        assert mine[0, 0, 1] < 0 < mine[H - 1, 0, 1] and mine[0, 0, 0] < 0       # row 0 looks up (y-down world), column 0 left


@pytest.mark.parametrize("W,H", [(64, 36), (160, 90)])
def test_primary_rays_noise_off(oracle, W, H):
    """BASELINE config #2's shape (1 frame, bounces = 1, NOISE_OFF): at most 1 LSB per channel, except silhouette
    pixels whose d > 0 / t > 0 / focus < .5 the two precisions decide differently -- counted, a handful at most."""
    flags = oracle.FLAG_NOISE_OFF if hasattr(oracle, "FLAG_NOISE_OFF") else 1
    theirs, acc, _ = oracle.render(oracle.default_scene(), oracle.camera(W, H),
                                   oracle.make_params(W, H, spp=1, bounces=1, paths=1, samples=1, mode=0, flags=flags, seed=SEED))
    mine, acc64 = rd.render_frames(rd.default_scene(), rd.camera_rays(W, H), (0, 0, 0), W, H, spp=1, bounces=1, paths=1,
                                   samples=1, pathtrace=False, keys=seed_keys(oracle), flags=rd.FLAG_NOISE_OFF)
    diff = np.abs(channels(mine) - channels(theirs)).max(axis=-1)
    off = int((diff > 1).sum())
    assert off <= 3, f"{off} pixels beyond 1 LSB: {np.argwhere(diff > 1)[:8].tolist()}"
    assert np.abs(acc64 - acc.astype(np.float64))[diff <= 1].max() < 2e-5
    assert (diff > 0).mean() < 0.02                                          # truncation of 255 * c flips a unit here and there


@pytest.mark.parametrize("W,H", [(64, 36), (160, 90)])
def test_four_frames_five_bounces_pathtrace(oracle, W, H):
    """The reference's own mode (noise on every bounce, camera-origin bounce rays, indirect sampling): 4 spp x 5 bounces.
    Stated bound between the two readings, on the accumulated radiance relative to max(1, |value|) -- the
    reference's un-normalised cos_theta lets `direct` go far below zero and 1 - exp(-direct) then reaches -1e6 on a few
    pixels, where float32 cannot hold more than five digits --: mean <= 2e-5, at most 0.1 % of the pixels beyond 1e-3
    (each a path that took the other side of a discontinuity: measured 0 of 2304 and 1 of 14400); 8-bit MAE <= 0.01 LSB,
    at most 0.1 % of the pixels beyond 2 LSB."""
    kw = dict(spp=4, bounces=5, paths=1, samples=1)
    theirs, acc, _ = oracle.render(oracle.default_scene(), oracle.camera(W, H), oracle.make_params(W, H, mode=1, flags=0, seed=SEED, **kw))
    mine, acc64 = rd.render_frames(rd.default_scene(), rd.camera_rays(W, H), (0, 0, 0), W, H, pathtrace=True,
                                   keys=seed_keys(oracle), **kw)
    rel = np.abs(acc64 - acc.astype(np.float64)) / np.maximum(1.0, np.abs(acc64))
    assert rel.mean() <= 2e-5, rel.mean()
    assert (rel.max(axis=-1) > 1e-3).mean() <= 1e-3
    # un-clamped pack (Renderer.cpp:44-55): a x1.2 channel at 255.99 / 256.01 wraps into its neighbour; compare clamped
    d8 = np.abs(np.clip((acc64 / 4 * 255).astype(np.int64), 0, 255) - np.clip((acc.astype(np.float64) / 4 * 255).astype(np.int64), 0, 255))
    assert d8.mean() <= 0.01, d8.mean()
    assert (d8.max(axis=-1) > 2).mean() <= 1e-3
    # and the packed words themselves agree wherever the accumulators agree to the last bits
    same = rel.max(axis=-1) < 1e-6
    assert same.mean() > 0.95
    assert (np.abs(channels(mine) - channels(theirs)).max(axis=-1)[same] <= 1).all()


def test_origin_from_hit_reading(oracle):
    """LUMA_FLAG_ORIGIN_FROM_HIT (bounce rays from the extruded hit point, Renderer.cpp:221-222,:231) read the same way."""
    W, H = 64, 36
    kw = dict(spp=2, bounces=4, paths=1, samples=1)
    theirs, acc, _ = oracle.render(oracle.default_scene(), oracle.camera(W, H), oracle.make_params(W, H, mode=1, flags=2, seed=SEED, **kw))
    mine, acc64 = rd.render_frames(rd.default_scene(), rd.camera_rays(W, H), (0, 0, 0), W, H, pathtrace=True,
                                   keys=seed_keys(oracle), flags=rd.FLAG_ORIGIN_FROM_HIT, **kw)
    rel = np.abs(acc64 - acc.astype(np.float64)) / np.maximum(1.0, np.abs(acc64))
    assert rel.mean() <= 5e-5 and (rel.max(axis=-1) < 1e-5).mean() > 0.95


def test_a_denser_scene_and_two_paths(oracle):
    """40 spheres of the synthetic generator (SURVEY 8d config #5's recipe), two indirect paths per hit, samples = 2:
    the scan order / closest-hit rule (:242, :266), the path loop (:192) and the division by `samples` (:212)."""
    W, H = 64, 36
    scene = oracle.synthetic_scene(40, SEED)
    rows = np.array([[*s.pos, s.radius, *s.diffuse, s.albedo, s.metallic, s.roughness] for s in scene], dtype=np.float64)
    kw = dict(spp=2, bounces=3, paths=2, samples=2)
    theirs, acc, _ = oracle.render(scene, oracle.camera(W, H), oracle.make_params(W, H, mode=1, flags=0, seed=SEED, **kw))
    mine, acc64 = rd.render_frames(rows, rd.camera_rays(W, H), (0, 0, 0), W, H, pathtrace=True, keys=seed_keys(oracle), **kw)
    rel = np.abs(acc64 - acc.astype(np.float64)) / np.maximum(1.0, np.abs(acc64))
    assert rel.mean() <= 5e-5 and (rel.max(axis=-1) < 1e-5).mean() > 0.95, (rel.mean(), (rel.max(axis=-1) < 1e-5).mean())
