"""LUMA_FLAG_GRID kernel ms at 1080p x 16 spp for scenes of 1024 - 262144 spheres, one line per build (LUMA_HIP_LIB):
used to set the direction grid's resolution cap (luma_grid.cpp).  Run on the GPU box from the repo root."""
import os, sys; sys.path.insert(0, '.')
from raytracer_amd import abi
W, H = 1920, 1080
row = "%-12s" % os.path.basename(os.environ.get("LUMA_HIP_LIB", "libluma_hip.so"))
for n in (1024, 4096, 8192, 16384, 65536, 262144):
    with abi.Context(0) as ctx:
        ctx.set_scene(abi.synthetic_scene(n)); ctx.set_camera(abi.default_camera(W, H)); ctx.set_frame(W, H)
        p = abi.make_params(W, H, spp=16, bounces=5, mode=1, flags=abi.FLAG_GRID)
        ctx.reset(); ctx.render_device(p, want_stats=True)
        st = ctx.render_device(p, want_stats=True)
        row += "  n=%d: %.3f" % (n, st.kernel_ms)
print(row, flush=True)
