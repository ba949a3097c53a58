"""1080p frames of 2..16 frames per call on the default scene: kernel ms (best of 5).  LUMA_AB=1 LUMA_HIP_LIB=... to compare builds."""
import sys
sys.path.insert(0, '.')
from raytracer_amd import abi
W, H = 1920, 1080
with abi.Context(0) as ctx:
    ctx.set_scene(abi.default_scene()); ctx.set_camera(abi.default_camera(W, H)); ctx.set_frame(W, H)
    ctx.reserve(16)
    out = []
    for spp in (2, 3, 4, 6, 7, 8, 16):
        p = abi.make_params(W, H, spp=spp, bounces=5, paths=1, samples=1, mode=1)
        ctx.render_device(p, want_stats=True)
        out.append("%d: %.4f" % (spp, min(ctx.render_device(p, want_stats=True).kernel_ms for _ in range(5))))
    print("  ".join(out))
