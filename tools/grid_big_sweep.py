"""LUMA_FLAG_GRID kernel ms at 1080p x 16 spp for scenes of 90000 - 400000 spheres (tables in global memory), one line
per build: LUMA_HIP_LIB names it.  Run on the GPU box from the repo root."""
import os, sys; sys.path.insert(0, '.')
from raytracer_amd import abi
W, H = 1920, 1080
row = "%-14s" % os.path.basename(os.environ.get("LUMA_HIP_LIB", "libluma_hip.so"))
for n in (90000, 131072, 180000, 262144, 400000):
    with abi.Context(0) as ctx:
        ctx.set_scene(abi.synthetic_scene(n)); ctx.set_camera(abi.default_camera(W, H)); ctx.set_frame(W, H)
        p = abi.make_params(W, H, spp=16, bounces=5, mode=1, flags=abi.FLAG_GRID)
        ctx.reset(); ctx.render_device(p, want_stats=True)
        st = ctx.render_device(p, want_stats=True)
        row += "  n=%d: %.3f" % (n, st.kernel_ms)
print(row, flush=True)
