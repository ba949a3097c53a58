"""1080p frames of scenes with a run-time sphere count (17 to 319 spheres) on the ray-per-lane kernel: kernel ms at 16 and 64
frames per call (best of 3).  A sweep build (LUMA_SP_MIN) keeps the sphere-parallel kernel out.  LUMA_AB=1 LUMA_HIP_LIB=..."""
import os, sys
os.environ.setdefault("LUMA_SP_MIN", "100000")
sys.path.insert(0, '.')
from raytracer_amd import abi
W, H = 1920, 1080
for flags in (0, abi.FLAG_ORIGIN_FROM_HIT):
    for spp in ([int(v) for v in sys.argv[1:]] or [16, 64]):
        out = []
        for n in (24, 48, 64, 100, 150, 200, 256, 319):
            with abi.Context(0) as ctx:
                ctx.set_scene(abi.synthetic_scene(n)); ctx.set_camera(abi.default_camera(W, H)); ctx.set_frame(W, H)
                p = abi.make_params(W, H, spp=spp, bounces=5, mode=1, flags=flags)
                ctx.render_device(p, want_stats=True)
                out.append("%d: %.3f" % (n, min(ctx.render_device(p, want_stats=True).kernel_ms for _ in range(2))))
        print("flags %d, %2d frames  " % (flags, spp) + "  ".join(out), flush=True)
