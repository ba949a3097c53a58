"""Kernel ms of every rank's band of the config #3 frame cut for N GPUs, by stripe height (interleaved stripes of
`stripe` rows; stripe == rows per rank is one contiguous band).  The slowest rank sets the frame time.
Run on the GPU box from the repo root."""
import sys; sys.path.insert(0, '.')
from raytracer_amd import abi
W, H = 1920, 1080
for N in (2, 4, 8):
    rows = H // N
    for stripe in [s for s in (3, 5, 9, 15, 27, 45, 135, rows) if rows % s == 0 and (s < rows or s == rows)]:
        if stripe == rows and stripe in (3, 5, 9, 15, 27, 45, 135) and stripe != rows: continue
        times = []
        for r in range(N):
            with abi.Context(0) as ctx:
                ctx.set_scene(abi.default_scene()); ctx.set_camera(abi.default_camera(W, H))
                if stripe == rows: ctx.set_frame(W, H, r * rows, (r + 1) * rows)
                else: ctx.set_frame_striped(W, H, stripe, r, N)
                ctx.reserve(64)
                p = abi.make_params(W, H, spp=64, bounces=5, paths=1, samples=1, mode=1)
                ctx.render_device(p, want_stats=True)
                t = min(ctx.render_device(p, want_stats=True).kernel_ms for _ in range(3))
                times.append(t)
        print("N=%d stripe %4d%s: max %.4f  mean %.4f  min %.4f" % (N, stripe, " (contiguous)" if stripe == rows else "", max(times), sum(times) / N, min(times)), flush=True)
