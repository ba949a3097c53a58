"""Bands of a 1080p x 64 spp frame of 64 / 200 spheres by lanes per pixel (sweep build: LUMA_FORCE_SPLIT, LUMA_SP_MIN): kernel ms."""
import os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, '.')
    from raytracer_amd import abi
    W, H = 1920, 1080
    out = []
    for n in (64, 200):
        for rows in (540, 360, 270):
            with abi.Context(0) as ctx:
                ctx.set_scene(abi.synthetic_scene(n)); ctx.set_camera(abi.default_camera(W, H)); ctx.set_frame(W, H, 0, rows)
                p = abi.make_params(W, H, spp=64, bounces=5, mode=1, flags=int(sys.argv[2]))
                ctx.render_device(p, want_stats=True)
                st = min((ctx.render_device(p, want_stats=True) for _ in range(2)), key=lambda s: s.kernel_ms)
                out.append("n=%d rows=%d: %.3f (%d)" % (n, rows, st.kernel_ms, st.lanes_per_pixel))
    print("  ".join(out))
    sys.exit(0)
for flags in (0, 2):
    for split in ("", "1", "2", "4"):
        env = dict(os.environ, LUMA_SP_MIN="100000")
        if split: env["LUMA_FORCE_SPLIT"] = split
        r = subprocess.run([sys.executable, __file__, "child", str(flags)], env=env, capture_output=True, text=True, timeout=300)
        print("flags %d, lanes %-7s %s" % (flags, split or "library", r.stdout.strip() or r.stderr[-300:]), flush=True)
