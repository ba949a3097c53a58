"""Frame ms of the host-target calls (kernel + device->host copy), pageable vs page-locked frame,
one context vs a group of 4 contexts on the same GPU.  Run on the GPU box from the repo root."""
import sys; sys.path.insert(0, '.')
import numpy as np
from raytracer_amd import abi
W, H = 1920, 1080


def report(tag, st):
    print("%-34s kernel_ms=%.3f gather_ms=%.3f total_ms=%.3f" % (tag, st.kernel_ms, st.gather_ms, st.total_ms))


for spp, mode, bounces, flags in ((64, 1, 5, 0), (1, 0, 1, 1)):
    p = abi.make_params(W, H, spp=spp, bounces=bounces, mode=mode, flags=flags)
    with abi.Context(0) as ctx, abi.HostFrame(H, W) as hf:
        ctx.set_scene(abi.default_scene()); ctx.set_camera(abi.default_camera(W, H)); ctx.set_frame(W, H)
        pageable = np.zeros((H, W), dtype=np.uint32)
        for i in range(3):
            _, st = ctx.render(p, out=pageable)
        report(f"spp={spp} one ctx, pageable", st)
        for i in range(3):
            _, st = ctx.render(p, out=hf.array)
        report(f"spp={spp} one ctx, page-locked", st)
        assert np.array_equal(pageable[:8], hf.array[:8]) or spp > 1
    with abi.Group([0, 0, 0, 0]) as g, abi.HostFrame(H, W) as hf:
        g.set_scene(abi.default_scene()); g.set_camera(abi.default_camera(W, H)); g.set_frame(W, H)
        pageable = np.zeros((H, W), dtype=np.uint32)
        for i in range(3):
            _, st = g.render(p, out=pageable)
        report(f"spp={spp} group of 4, pageable", st)
        for i in range(3):
            _, st = g.render(p, out=hf.array)
        report(f"spp={spp} group of 4, page-locked", st)
