"""Sphere tests per ray and kernel time with and without LUMA_FLAG_GRID (synthetic scenes, 1080p).
Run on the GPU box from the repo root."""
import sys; sys.path.insert(0, '.')
from raytracer_amd import abi
W, H = 1920, 1080
for n in (64, 320, 1024, 4096, 16384, 65536, 262144):
    with abi.Context(0) as ctx:
        ctx.set_scene(abi.synthetic_scene(n)); ctx.set_camera(abi.default_camera(W, H)); ctx.set_frame(W, H)
        row = "n=%6d" % n
        for flags in (0, abi.FLAG_GRID):
            if flags == 0 and n > 16384:
                row += "   linear: (skipped)                              "
                continue
            p = abi.make_params(W, H, spp=2, bounces=5, mode=1, flags=flags)
            ctx.reset()
            ctx.render_device(p, want_stats=True)
            st = ctx.render_device(p, want_stats=True)
            row += "   %s: %8.3f ms %7.1f tests/ray %8.1f Mrays/s" % ("grid" if flags else "linear", st.kernel_ms, st.sphere_tests / st.rays, st.rays / st.kernel_ms / 1e3)
        print(row, flush=True)
