"""Kernel ms of the top `rows` rows of the config #3 frame, for the kernel LUMA_WF selects: with 60 tiles of 32x4
pixels per 4 rows, 512 rows are 7680 waves of the wavefront kernel (one resident set at 30 waves per CU).
Run on the GPU box from the repo root."""
import sys; sys.path.insert(0, '.')
from raytracer_amd import abi
W, H = 1920, 1080
for rows in (128, 256, 512, 768, 1024, 1080):
    with abi.Context(0) as ctx:
        ctx.set_scene(abi.default_scene()); ctx.set_camera(abi.default_camera(W, H))
        ctx.set_frame(W, H, 0, rows)
        ctx.reserve(64)
        p = abi.make_params(W, H, spp=64, bounces=5, paths=1, samples=1, mode=1)
        ctx.render_device(p, want_stats=True)
        st = [ctx.render_device(p, want_stats=True) for _ in range(5)]
        print("rows %4d: %.4f ms  (%.3f us per row)  lanes x%d" % (rows, min(s.kernel_ms for s in st), 1e3 * min(s.kernel_ms for s in st) / rows, st[0].lanes_per_pixel), flush=True)
