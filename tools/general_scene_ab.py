"""Kernel ms of 1080p x 64 spp frames outside the compile-time sphere-count instances: the default scene with
LUMA_FLAG_ORIGIN_FROM_HIT, synthetic scenes of 16 / 64 / 200 spheres, whole and as one of eight stripe bands.
Run from the repo root with LUMA_HIP_LIB naming the build."""
import sys; sys.path.insert(0, '.')
from raytracer_amd import abi
W, H = 1920, 1080
out = []
for n, flags in ((0, 2), (16, 0), (64, 0), (200, 2)):
    for N in (1, 8):
        with abi.Context(0) as ctx:
            ctx.set_scene(abi.synthetic_scene(n) if n else abi.default_scene()); ctx.set_camera(abi.default_camera(W, H))
            if N > 1: ctx.set_frame_striped(W, H, 15, 0, N)
            else: ctx.set_frame(W, H)
            ctx.reserve(64)
            p = abi.make_params(W, H, spp=64, bounces=5, paths=1, samples=1, mode=1, flags=flags)
            for _ in range(3): ctx.render_device(p, want_stats=True)
            out.append(min(ctx.render_device(p, want_stats=True).kernel_ms for _ in range(8)))
print(" ".join("%.3f" % t for t in out))
