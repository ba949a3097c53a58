"""Kernel ms of a 640x360 x 16 spp frame of 2000 spheres (linear scan, LUMA_FLAG_GRID) and of 30000 (grid): 3600 tiles for
4096 resident waves, the case where the persistent kernels hand their tiles out row by row.  Run from the repo root."""
import sys; sys.path.insert(0, '.')
from raytracer_amd import abi
W, H = 640, 360
out = []
for n, flags in ((2000, 0), (2000, abi.FLAG_GRID), (30000, abi.FLAG_GRID)):
    with abi.Context(0) as ctx:
        ctx.set_scene(abi.synthetic_scene(n)); ctx.set_camera(abi.default_camera(W, H)); ctx.set_frame(W, H)
        p = abi.make_params(W, H, spp=16, bounces=5, paths=1, samples=1, mode=1, flags=flags)
        ctx.render_device(p, want_stats=True)
        out.append(min(ctx.render_device(p, want_stats=True).kernel_ms for _ in range(5)))
print(" ".join("%.3f" % t for t in out), end=" ")
