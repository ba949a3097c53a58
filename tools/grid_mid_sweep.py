"""LUMA_FLAG_GRID kernel ms, tests per ray and Mrays/s at 1080p x 16 spp for scenes of 4096 - 16384 spheres: the sizes
at which the wave walk's tables leave LDS (residency tiers of render_kernel_sp<GRID>).  One line per scene; LUMA_HIP_LIB
names the build.  Run on the GPU box from the repo root."""
import os, sys; sys.path.insert(0, '.')
from raytracer_amd import abi
W, H = 1920, 1080
for n in (4096, 4600, 5000, 6500, 8192, 10000, 12000, 14000, 16384):
    with abi.Context(0) as ctx:
        ctx.set_scene(abi.synthetic_scene(n)); ctx.set_camera(abi.default_camera(W, H)); ctx.set_frame(W, H)
        row = "%-14s n=%6d" % (os.path.basename(os.environ.get("LUMA_HIP_LIB", "libluma_hip.so")), n)
        for spp in (16,):
            p = abi.make_params(W, H, spp=spp, bounces=5, mode=1, flags=abi.FLAG_GRID)
            ctx.reset()
            ctx.render_device(p, want_stats=True)
            st = ctx.render_device(p, want_stats=True)
            row += "   spp %2d: %8.3f ms %7.1f tests/ray %8.1f Mrays/s" % (spp, st.kernel_ms, st.sphere_tests / st.rays, st.rays / st.kernel_ms / 1e3)
        print(row, flush=True)
