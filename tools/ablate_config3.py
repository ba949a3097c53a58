"""Kernel time of config #3 variants (1080p, 64 spp) to see where the time goes."""
import sys; sys.path.insert(0, '.')
from raytracer_amd import abi
W, H = 1920, 1080
def run(name, **kw):
    with abi.Context(0) as ctx:
        ctx.set_scene(abi.default_scene()); ctx.set_camera(abi.default_camera(W, H)); ctx.set_frame(W, H)
        p = abi.make_params(W, H, **kw)
        best = 1e9
        for i in range(6):
            st = ctx.render_device(p, want_stats=True)
            best = min(best, st.kernel_ms)
        print(f"{name:34s} kernel_ms={best:.3f} rays={st.rays/1e6:.1f}M  ns/ray={best*1e6/st.rays:.4f}")
run("config3 (pathtrace, 5 bounces)", spp=64, bounces=5, mode=1)
run("raytrace mode (no indirect rays)", spp=64, bounces=5, mode=0)
run("pathtrace, 1 bounce", spp=64, bounces=1, mode=1)
run("raytrace, 1 bounce", spp=64, bounces=1, mode=0)
run("raytrace, 1 bounce, noise off", spp=64, bounces=1, mode=0, flags=1)
run("pathtrace, 5 bounces, noise off", spp=64, bounces=5, mode=1, flags=1)
run("pathtrace, 2 bounces", spp=64, bounces=2, mode=1)
run("pathtrace 5 bounces paths=2", spp=64, bounces=5, mode=1, paths=2)
