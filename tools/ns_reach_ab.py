"""Should the compile-time sphere-count instances reach further?  Kernel ms of 1080p x 64 spp frames, whole and one of
eight stripe bands, for the default scene with LUMA_FLAG_ORIGIN_FROM_HIT and for 12 synthetic spheres, with the general
instances and (LUMA_NS_REACH=1, sweep build) with a compile-time count.  Run on the GPU box from the repo root with
LUMA_AB=1 LUMA_HIP_LIB=$PWD/ab/sweep.so."""
import os, subprocess, sys
W, H = 1920, 1080
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, '.')
    from raytracer_amd import abi
    out = []
    for n, flags in ((0, 2), (12, 0)):
        for N in (1, 8):
            with abi.Context(0) as ctx:
                ctx.set_scene(abi.synthetic_scene(n) if n else abi.default_scene()); ctx.set_camera(abi.default_camera(W, H))
                if N > 1: ctx.set_frame_striped(W, H, 15, 0, N)
                else: ctx.set_frame(W, H)
                ctx.reserve(64)
                p = abi.make_params(W, H, spp=64, bounces=5, paths=1, samples=1, mode=1, flags=flags)
                for _ in range(3): ctx.render_device(p, want_stats=True)
                out.append(min(ctx.render_device(p, want_stats=True).kernel_ms for _ in range(8)))
    print(" ".join("%.3f" % t for t in out))
    sys.exit(0)
print("                         from_hit(5) whole, band/8 | 12 spheres whole, band/8")
for rep in (1, 2):
    for name, env in (("general instances     ", {}), ("compile-time count    ", {"LUMA_NS_REACH": "1"})):
        e = dict(os.environ, **env); e.pop("LUMA_NS_REACH", None) if not env else None
        r = subprocess.run([sys.executable, __file__, "child"], env=e, capture_output=True, text=True, timeout=300)
        print(name, r.stdout.strip() or r.stderr[-500:], flush=True)
