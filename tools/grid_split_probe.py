"""Where the LUMA_FLAG_GRID kernel's time goes on one of config #5's eight bands: camera-origin rays
(DOF + bounce rays, all from camera.pos) against the indirect rays from hit points.  paths=0 and
mode=raytrace both drop the indirect rays (Renderer.cpp:187-192); bounces=1 leaves DOF + one bounce ray.
Run on the GPU box from the repo root."""
import sys; sys.path.insert(0, '.')
from raytracer_amd import abi
W, H = 7680, 4320
with abi.Context(0) as ctx:
    ctx.set_scene(abi.synthetic_scene(4096)); ctx.set_camera(abi.default_camera(W, H)); ctx.set_frame_striped(W, H, 15, 3, 8)
    for name, kw in (("pathtrace paths=1 bounces=5 (config #5)", dict(bounces=5, mode=1, paths=1)),
                     ("pathtrace paths=0 bounces=5 (no indirect rays)", dict(bounces=5, mode=1, paths=0)),
                     ("pathtrace paths=1 bounces=1", dict(bounces=1, mode=1, paths=1)),
                     ("raytrace bounces=1 (camera rays only)", dict(bounces=1, mode=0, paths=0))):
        for flags in (abi.FLAG_GRID, 0):
            p = abi.make_params(W, H, spp=16, flags=flags, **kw)
            ctx.reset()
            ctx.render_device(p, want_stats=True)
            st = ctx.render_device(p, want_stats=True)
            print("%-48s %-6s %8.3f ms  %9.0f Mrays  %7.1f tests/ray  %8.1f Mrays/s" % (
                name, "grid" if flags else "linear", st.kernel_ms, st.rays / 1e6, st.sphere_tests / st.rays, st.rays / st.kernel_ms / 1e3), flush=True)
