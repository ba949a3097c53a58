"""Kernel ms of one GPU's share of the config #3 frame (15-row stripes of an N-way cut) by frames per call:
the intercept of the line through the points is what a launch costs whatever its frames (ramp, drain, the
serial epilogue), the slope the steady rate.  Run on the GPU box from the repo root."""
import sys; sys.path.insert(0, '.')
from raytracer_amd import abi
W, H = 1920, 1080
for N in (1, 4, 8, 12, 24):
    row = []
    for spp in (16, 32, 64, 128, 256):
        with abi.Context(0) as ctx:
            ctx.set_scene(abi.default_scene()); ctx.set_camera(abi.default_camera(W, H))
            if N > 1: ctx.set_frame_striped(W, H, 15, 0, N)
            else: ctx.set_frame(W, H, 0, H)
            ctx.reserve(spp)
            p = abi.make_params(W, H, spp=spp, bounces=5, paths=1, samples=1, mode=1)
            ctx.render_device(p, want_stats=True)
            st = [ctx.render_device(p, want_stats=True) for _ in range(5)]
            row.append((spp, min(s.kernel_ms for s in st), st[0].lanes_per_pixel))
    print("N=%2d " % N + "  ".join("spp %3d: %.4f ms (x%d)" % r for r in row), flush=True)
