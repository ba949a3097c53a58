"""Config #4 (3840x2160, 256 frames per call) as one launch against chunks of 128 / 64 frames (each chunk runs the TAKES
instances: frames taken inside waves).  Kernel ms summed over the chunks.  Run on the GPU box from the repo root."""
import sys
sys.path.insert(0, '.')
from raytracer_amd import abi
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
rows = int(sys.argv[1]) if len(sys.argv) > 1 and int(sys.argv[1]) else H
with abi.Context(0) as ctx:
    ctx.set_scene(abi.default_scene()); ctx.set_camera(abi.default_camera(W, H)); ctx.set_frame(W, H, 0, rows)
    ctx.reserve(256)
    for chunk in (256, 128, 64):
        best = 1e9
        for rep in range(4):
            ctx.reset()
            ms = 0.0
            for _ in range(256 // chunk):
                ms += ctx.render_device(abi.make_params(W, H, spp=chunk, bounces=5, paths=1, samples=1, mode=1), want_stats=True).kernel_ms
            best = min(best, ms)
        print("%dx%d, rows %d: 256 frames in chunks of %3d: %.3f ms" % (W, H, rows, chunk, best))
