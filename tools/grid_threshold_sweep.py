"""LUMA_FLAG_GRID kernel ms at 1080p, 2 and 16 spp, for scenes of 64 - 511 spheres: builds with kGridWaveWalkMin at 512
(lane-per-ray walk for all of them) and at 1 (wave walk) give the crossover quoted at grid_wave_walk().  One line per
spp; LUMA_HIP_LIB names the build.  Run on the GPU box from the repo root."""
import os, sys; sys.path.insert(0, '.')
from raytracer_amd import abi
W, H = 1920, 1080
for spp in (2, 16):
    row = "%-12s spp %2d" % (os.path.basename(os.environ.get("LUMA_HIP_LIB", "libluma_hip.so")), spp)
    for n in (64, 100, 160, 250, 320, 400, 511):
        with abi.Context(0) as ctx:
            ctx.set_scene(abi.synthetic_scene(n)); ctx.set_camera(abi.default_camera(W, H)); ctx.set_frame(W, H)
            p = abi.make_params(W, H, spp=spp, bounces=5, mode=1, flags=abi.FLAG_GRID)
            ctx.reset(); ctx.render_device(p, want_stats=True)
            st = ctx.render_device(p, want_stats=True)
            row += "  n=%d: %.3f" % (n, st.kernel_ms)
    print(row, flush=True)
