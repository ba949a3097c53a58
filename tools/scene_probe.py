"""Kernel ms for a few synthetic scene sizes (1080p, 4 spp): quick A/B companion of scene_size_sweep.py."""
import sys; sys.path.insert(0, '.')
from raytracer_amd import abi
W, H = 1920, 1080
out = []
for n in [int(v) for v in sys.argv[1:]] or [320, 512, 1024]:
    with abi.Context(0) as ctx:
        ctx.set_scene(abi.synthetic_scene(n)); ctx.set_camera(abi.default_camera(W, H)); ctx.set_frame(W, H)
        p = abi.make_params(W, H, spp=4, bounces=5, mode=1)
        ctx.render_device(p, want_stats=True)
        st = ctx.render_device(p, want_stats=True)
        out.append("n=%d %.3f ms" % (n, st.kernel_ms))
print("  ".join(out))
