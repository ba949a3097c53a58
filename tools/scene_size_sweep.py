"""Kernel ms against scene size (1080p, 4 spp, 5 bounces, pathtrace) for BOTH linear-scan kernels around their
crossover: a sweep build (-DLUMA_SWEEP_FINE_PCT, LUMA_AB=1 LUMA_HIP_LIB=ab/sweep.so) takes the threshold from
LUMA_SP_MIN, so each scene is timed with the ray-per-lane kernel (LUMA_SP_MIN=100000) and the sphere-parallel one
(LUMA_SP_MIN=1) in child processes.  Run on the GPU box from the repo root: python tools/scene_size_sweep.py"""
import os, subprocess, sys
W, H = 1920, 1080
SIZES = (5, 8, 9, 12, 16, 24, 32, 64, 96, 128, 160, 192, 224, 256, 288, 320, 384, 448, 512, 640, 1024)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, '.')
    from raytracer_amd import abi
    for n in SIZES:
        with abi.Context(0) as ctx:
            ctx.set_scene(abi.synthetic_scene(n)); ctx.set_camera(abi.default_camera(W, H)); ctx.set_frame(W, H)
            p = abi.make_params(W, H, spp=4, bounces=5, mode=1)
            ctx.render_device(p, want_stats=True)
            st = min((ctx.render_device(p, want_stats=True) for _ in range(3)), key=lambda s: s.kernel_ms)
            print(n, "%.4f" % st.kernel_ms, flush=True)
    sys.exit(0)
res = {}
for name, sp_min in (("lane", "100000"), ("sp", "1")):
    out = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, LUMA_SP_MIN=sp_min), capture_output=True, text=True, timeout=300)
    if out.returncode != 0:
        print(out.stderr[-2000:]); sys.exit(1)
    res[name] = dict((int(a), float(b)) for a, b in (ln.split() for ln in out.stdout.splitlines()))
print("spheres  ray-per-lane ms  sphere-parallel ms  ratio")
for n in SIZES:
    print("%6d  %14.4f  %18.4f  %5.2f" % (n, res["lane"][n], res["sp"][n], res["lane"][n] / res["sp"][n]))
