"""Per-wave loop statistics of the ray-per-lane kernel from a -DLUMA_DBG_TIMELINE -DLUMA_DBG_STEAL build
(LUMA_AB=1 LUMA_HIP_LIB=ab/dbgsteal.so): loop iterations, runs of the hit piece, frames taken from other pixels,
pairing events, duration.  Run on the GPU box from the repo root."""
import ctypes, sys
import numpy as np
sys.path.insert(0, '.')
from raytracer_amd import abi
W, H = 1920, 1080
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
with abi.Context(0) as ctx:
    ctx.set_scene(abi.default_scene()); ctx.set_camera(abi.default_camera(W, H))
    ctx.set_frame(W, H, 0, H)
    ctx.reserve(spp)
    p = abi.make_params(W, H, spp=spp, bounces=5, paths=1, samples=1, mode=1)
    for _ in range(20): st = ctx.render_device(p, want_stats=True)
    print("spp=%d kernel %.4f ms, %d lanes per pixel" % (spp, st.kernel_ms, st.lanes_per_pixel))
    L = abi.lib()
    buf = np.zeros((1 << 18, 4), dtype=np.uint32)
    L.luma_dbg_timeline_copy.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    assert L.luma_dbg_timeline_copy(buf.ctypes.data, buf.nbytes) == 0
used = buf[(buf[:, 0] != 0) | (buf[:, 1] != 0)]
dur = (used[:, 1].astype(np.int64) - used[:, 0].astype(np.int64)) / 100.0
iters = used[:, 2] & 0xFFFF; hits = used[:, 2] >> 16; taken = used[:, 3] & 0xFFFF; events = used[:, 3] >> 16
print("waves", len(used))
for name, sel in (("waves that took frames or could (one lane per pixel)", iters > 0),):
    pass
coarse = np.arange(len(used)) < int(len(used) * 0.68)          # (the coarse class comes first in the grid; the fine one has four waves per 64 pixels)
for name, m in (("first 68 % of the waves (coarse class)", coarse), ("the rest (fine class)", ~coarse)):
    print(name)
    for lab, v in (("iterations", iters), ("hit-piece runs", hits), ("frames taken", taken), ("pairing events", events), ("duration us", dur)):
        x = v[m]
        print("  %-16s mean %8.1f  p10 %8.1f  p50 %8.1f  p90 %8.1f  max %8.1f" % (lab, x.mean(), *np.percentile(x, [10, 50, 90]), x.max()))
