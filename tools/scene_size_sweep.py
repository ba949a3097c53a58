"""Sphere-test throughput against scene size (1080p, 4 spp, 5 bounces, pathtrace): shows where the
ray-per-lane kernel hands over to the sphere-parallel one.  Run on the GPU box from the repo root."""
import sys; sys.path.insert(0, '.')
from raytracer_amd import abi
W, H = 1920, 1080
for n in (5, 16, 32, 64, 128, 192, 256, 319, 320, 384, 512, 640, 641, 1024, 1025, 2048, 4096, 5056, 5057, 8192):
    spp = 4 if n <= 1024 else 1
    with abi.Context(0) as ctx:
        ctx.set_scene(abi.synthetic_scene(n)); ctx.set_camera(abi.default_camera(W, H)); ctx.set_frame(W, H)
        p = abi.make_params(W, H, spp=spp, bounces=5, mode=1)
        ctx.render_device(p, want_stats=True)
        st = ctx.render_device(p, want_stats=True)
        print("n=%5d kernel_ms=%9.3f Gtests/s=%8.1f Mrays/s=%9.1f" % (n, st.kernel_ms, st.sphere_tests / st.kernel_ms / 1e6, st.rays / st.kernel_ms / 1e3), flush=True)
