"""A second, independent reading of the reference's shading path, in numpy float64, vectorised over pixels.

Written from /root/reference/Renderer.cpp:234-364 and :413-461 and Camera.cpp:169-216 directly -- not from oracle/, whose
C restatement and the HIP kernels share one reading -- so that a misreading common to those two shows up as a frame
difference (tests/test_independent_reading.py).  It follows the reference's own arithmetic forms (b = 2 dot, d = b^2 - 4ac,
t = (-b - sqrt d) / 2a, normalize = v / |v|, no fused operations), evaluated in double.  What the reference does NOT
define is taken as an input, not re-read: the random draws.  fx::Random is replaced by the build's counter RNG
(DESIGN.md 2), so this module draws from the same counters (pcg3d over pixel, frame, slot; restated below in numpy
and pinned against the oracle's exported lo_rng_vec3 by the test) and only maps the 24-bit draws to [lo, hi] itself.
fx:: semantics as SURVEY 8a-0 infers them (glm conventions); test infrastructure like the oracle.
"""
import numpy as np

FLT_MAX = float(np.finfo(np.float32).max)
FLAG_NOISE_OFF, FLAG_ORIGIN_FROM_HIT, FLAG_CLAMP_PACK, FLAG_NO_SHOW_DEPTH = 1, 2, 4, 8


# ---- the counter RNG (build-defined; an input to the reading) ---------------------------------------------------
def pcg3d(x, y, z):
    x = (x * np.uint32(1664525) + np.uint32(1013904223)).astype(np.uint32)
    y = (y * np.uint32(1664525) + np.uint32(1013904223)).astype(np.uint32)
    z = (z * np.uint32(1664525) + np.uint32(1013904223)).astype(np.uint32)
    x = x + y * z; y = y + z * x; z = z + x * y
    x = x ^ (x >> np.uint32(16)); y = y ^ (y >> np.uint32(16)); z = z ^ (z >> np.uint32(16))
    x = x + y * z; y = y + z * x; z = z + x * y
    return x, y, z


def draws(keys, pixel, sample, slot):
    """Three uniforms in [0, 1) (24 bits each) for every pixel: shape [n, 3] float64."""
    with np.errstate(over="ignore"):
        x, y, z = pcg3d(pixel.astype(np.uint32) ^ np.uint32(keys[0]),
                        (np.zeros_like(pixel, dtype=np.uint32) + np.uint32(sample)) ^ np.uint32(keys[1]),
                        slot.astype(np.uint32) ^ np.uint32(keys[2]))
    return np.stack([x >> np.uint32(8), y >> np.uint32(8), z >> np.uint32(8)], axis=1).astype(np.float64) * 2.0 ** -24


def random_vec3(keys, pixel, sample, slot, lo, hi):
    """fx::Random::vec3(lo, hi), Renderer.cpp:118,:226: lo, hi per pixel or scalars."""
    u = draws(keys, pixel, sample, slot)
    lo = np.asarray(lo, dtype=np.float64).reshape(-1, 1) if np.ndim(lo) else lo
    hi = np.asarray(hi, dtype=np.float64).reshape(-1, 1) if np.ndim(hi) else hi
    return lo + (hi - lo) * u


# ---- Camera.cpp ------------------------------------------------------------------------------------------------
def perspective_fov(fov, width, height, near, far):
    """glm::perspectiveFov (RH, depth -1..1), column-major M[col][row] as a [4,4] array indexed [row, col]."""
    h = np.cos(0.5 * fov) / np.sin(0.5 * fov)
    w = h * height / width
    m = np.zeros((4, 4))
    m[0, 0] = w
    m[1, 1] = h
    m[2, 2] = -(far + near) / (far - near)
    m[3, 2] = -1.0
    m[2, 3] = -(2.0 * far * near) / (far - near)
    return m


def look_at(eye, at, up):
    """glm::lookAt (RH)."""
    f = at - eye; f = f / np.linalg.norm(f)
    s = np.cross(f, up); s = s / np.linalg.norm(s)
    u = np.cross(s, f)
    m = np.eye(4)
    m[0, :3] = s; m[1, :3] = u; m[2, :3] = -f
    m[0, 3] = -s @ eye; m[1, 3] = -u @ eye; m[2, 3] = f @ eye
    return m


def camera_rays(width, height, fov_deg=70.0, near=0.1, far=100.0, pos=(0.0, 0.0, 0.0), direction=(0.0, 0.0, -1.0)):
    """Camera::Camera + recompute_projection / _view / _rays, Camera.cpp:77-90, :169-216.  [H, W, 3] float64."""
    pos = np.asarray(pos, dtype=np.float64); direction = np.asarray(direction, dtype=np.float64)
    pinv = np.linalg.inv(perspective_fov(np.radians(np.float32(fov_deg)).astype(np.float64), float(width), float(height), near, far))
    vinv = np.linalg.inv(look_at(pos, pos + direction, np.array([0.0, 1.0, 0.0])))
    x = np.arange(width, dtype=np.float64) / width * 2.0 - 1.0           # :195-200
    y = np.arange(height, dtype=np.float64) / height * 2.0 - 1.0
    cx, cy = np.meshgrid(x, y)
    ext = np.stack([cx, cy, np.ones_like(cx), np.ones_like(cx)], axis=-1)   # :202
    target = ext @ pinv.T                                                # :203
    corrected = target[..., :3] / target[..., 3:4]                       # :205-208
    normalized = corrected / np.linalg.norm(corrected, axis=-1, keepdims=True)   # :209
    padded = np.concatenate([normalized, np.zeros_like(cx)[..., None]], axis=-1)   # :210
    return (padded @ vinv.T)[..., :3]                                    # :212-213


# ---- Renderer.cpp ----------------------------------------------------------------------------------------------
def default_scene():
    """renderer.h:62-70, order s, a, q, r, t: rows of (pos xyz, radius, diffuse rgb, albedo, metallic, roughness)."""
    return np.array([
        [0.0, 0.5, -10.0, 1.0, 0.0, 0.0, 1.0, 1.0, 0.001, 0.4],
        [0.0, 102.0, 0.0, 100.0, 0.6, 0.6, 0.6, 1.0, 0.1, 0.5],
        [3.0, 0.5, -10.0, 1.0, 0.0, 1.0, 0.0, 1.0, 0.001, 0.4],
        [6.0, 0.5, -10.0, 1.0, 1.0, 0.0, 0.0, 1.0, 0.001, 0.4],
        [3.0, 0.5, -23.0, 10.0, 1.0, 1.0, 1.0, 1.0, 0.4, 0.0],
    ], dtype=np.float32).astype(np.float64)


def trace_ray(spheres, pos, direction):
    """Renderer::trace_ray, :234-287, for n rays at once: (object index or -1, t, hit point, normal)."""
    n = direction.shape[0]
    distance = np.full(n, FLT_MAX)                                        # :236-238
    obj = np.full(n, -1)
    hit_pos = np.zeros((n, 3)); hit_normal = np.zeros((n, 3))
    a = np.einsum("ij,ij->i", direction, direction)                      # :246
    for i, s in enumerate(spheres):                                      # :242
        diff = pos - s[:3]                                               # :244
        b = 2.0 * np.einsum("ij,ij->i", diff, direction)                 # :247
        c = np.einsum("ij,ij->i", diff, diff) - s[3] * s[3]              # :248
        d = b * b - 4.0 * a * c                                          # :251
        with np.errstate(invalid="ignore", divide="ignore"):
            t = (-b - np.sqrt(d)) / (2.0 * a)                            # :255
        take = (d > 0) & (t > 0) & (t < distance)                        # :253, :258, :266
        hit = pos + direction * t[:, None]                               # :260-261
        toward = hit - s[:3]                                             # :263
        with np.errstate(invalid="ignore", divide="ignore"):
            normal = toward / np.linalg.norm(toward, axis=1, keepdims=True)   # :264
        distance = np.where(take, t, distance)
        obj = np.where(take, i, obj)
        hit_pos = np.where(take[:, None], hit, hit_pos)
        hit_normal = np.where(take[:, None], normal, hit_normal)
    return obj, distance, hit_pos, hit_normal                            # :276-279: distance == max <=> miss


def render_frames(spheres, rays, cam_pos, width, height, spp, bounces, paths, samples, pathtrace, keys, flags=0,
                  focus_depth=10.0, show_depth=True):
    """Renderer::render_to for `spp` successive frames (:366-471) with render_pixel (:289-364), indirect_illumination
    (:182-216) and reflect_intersection (:218-232).  Returns (target uint32 [H, W], accumulated_data float64 [H, W, 3])."""
    n = width * height
    dir0 = rays.reshape(n, 3)
    pos0 = np.broadcast_to(np.asarray(cam_pos, dtype=np.float64), (n, 3))
    pixel = np.arange(n, dtype=np.uint32)                                # index = y * width + x, :399
    noise_on = not (flags & FLAG_NOISE_OFF)
    slots = 2 + paths                                                    # RNG slots per bounce
    top_sky = np.array([.529, .808, .922], dtype=np.float32).astype(np.float64)      # :308-309
    bottom_sky = np.array([.106, .275, .711], dtype=np.float32).astype(np.float64)
    f32 = lambda v: float(np.float32(v))

    # DOF pre-pass, :413-432 (no randomness: the same every frame)
    obj, dist, _, _ = trace_ray(spheres, pos0, dir0)                     # :418
    depth = np.where(obj >= 0, dist, FLT_MAX)                            # :419-423
    with np.errstate(over="ignore"):
        focus = 1.0 - np.exp(-np.power(depth - focus_depth, 10))         # :425-427
    blur = 0.5 + focus * (5.0 - 0.5)                                     # std::lerp, :429-432
    brighten = show_depth and not (flags & FLAG_NO_SHOW_DEPTH)

    accumulated = np.zeros((n, 3))
    frame_count = 1.0                                                    # renderer.h:53
    target = None
    for frame in range(spp):
        direct = np.zeros((n, 3))                                        # :291
        direction = dir0.copy()                                          # :293
        origin = pos0.copy()
        alive = np.ones(n, dtype=bool)
        for bounce in range(bounces):                                    # :300
            if not alive.any():
                break
            slot = np.full(n, bounce * slots, dtype=np.uint32)
            noise = random_vec3(keys, pixel, frame, slot, -blur, blur) if noise_on else 0.0   # :116-122
            dir_noised = direction + noise                               # :302
            ray_pos = origin if (flags & FLAG_ORIGIN_FROM_HIT) else pos0   # :303: ALWAYS camera.pos in the reference
            obj, t, hpos, hnormal = trace_ray(spheres, ray_pos, dir_noised)   # :304
            missed = alive & (obj < 0)                                   # :306
            clamped = np.clip(dir_noised[:, 1], -1.0, 1.0)               # :311
            adjusted = (clamped + 1.0) * 0.5                             # :312
            sky = top_sky * (1.0 - adjusted)[:, None] + bottom_sky * adjusted[:, None]   # ::lerp, :72-83, :314
            direct = np.where(missed[:, None], direct + sky, direct)     # :316
            alive = alive & ~missed                                      # :318
            o = np.maximum(obj, 0)
            mat = spheres[o]                                             # :326-329
            diffuse, albedo, metallic, roughness = mat[:, 4:7], mat[:, 7], mat[:, 8], mat[:, 9]
            cos_theta = np.einsum("ij,ij->i", -dir_noised, hnormal)      # :331-332
            lit = alive & (cos_theta >= 0)                               # :334
            indirect = np.zeros((n, 3))                                  # :336
            if pathtrace:                                                # :187
                for p in range(paths):                                   # :192
                    d = random_vec3(keys, pixel, frame, slot + np.uint32(2 + p), -1.0, 1.0)
                    d = d / np.linalg.norm(d, axis=1, keepdims=True)     # fx::Random::vec3_sphere, :194
                    flip = np.einsum("ij,ij->i", d, hnormal) < 0         # :197-200
                    d = np.where(flip[:, None], -d, d)
                    cobj, _, _, _ = trace_ray(spheres, hpos, hnormal + d)   # :202-204
                    indirect = indirect + np.where((cobj >= 0)[:, None], spheres[np.maximum(cobj, 0)][:, 4:7], 0.0)   # :206-209
            indirect = indirect * f32(1.0 / np.float32(samples)) if samples else indirect * 0.0   # :212-213
            indirect = np.where((albedo > 0)[:, None], indirect * albedo[:, None], 0.0)           # :337-341
            value = (cos_theta + 1.0) * 0.5                              # :343
            scalar = value * (1.0 - metallic)                            # :345
            capture = diffuse * scalar[:, None] + indirect * (1.0 - scalar)[:, None]   # lerp(diffuse, indirect, 1 - scalar), :346
            direct = np.where(lit[:, None], direct + capture * (scalar * f32(.8))[:, None], direct)   # :347-349
            alive = alive & (metallic != 0)                              # :353-357
            gain = f32(.2) * roughness if noise_on else 0.0 * roughness  # :225
            rough = random_vec3(keys, pixel, frame, slot + np.uint32(1), -gain, gain) if noise_on else 0.0   # :226
            normal = hnormal + rough                                     # :227
            direction = dir_noised - 2.0 * np.einsum("ij,ij->i", dir_noised, normal)[:, None] * normal   # fx::reflect, :229; :359
            origin = hpos + hnormal * f32(.001)                          # :221-222 (discarded by :359 unless ORIGIN_FROM_HIT)
        result = 1.0 - np.exp(-direct)                                   # ::tonemap, :57-70, :362
        if brighten:
            result = np.where((focus < .5)[:, None], result * f32(1.2), result)   # :448-454
        accumulated = accumulated + result                               # :457-458
        mean = accumulated * (1.0 / frame_count)                         # :460
        scaled = (mean * 255.0).astype(np.int64)                         # ::RGB, :44-55: static_cast<int>, OR-combined, no clamp
        if flags & FLAG_CLAMP_PACK:
            scaled = np.clip(scaled, 0, 255)
        target = ((scaled[:, 0] << 16) | (scaled[:, 1] << 8) | scaled[:, 2]).astype(np.uint32)   # :461
        frame_count += 1.0                                               # :470
    return target.reshape(height, width), accumulated.reshape(height, width, 3)
