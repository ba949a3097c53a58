"""Start / end / phase counts of every wave of the wavefront kernel (LUMA_WF=128), from a -DLUMA_DBG_TIMELINE build
(make SINGLE_TU=1 EXTRA=-DLUMA_DBG_TIMELINE OUT=.../ab/timeline.so OBJDIR=.../build/obj_tl).
Run on the GPU box from the repo root:  LUMA_AB=1 LUMA_HIP_LIB=$PWD/ab/timeline.so LUMA_WF=128 python tools/wf_timeline.py [rows]"""
import ctypes, sys
import numpy as np
sys.path.insert(0, '.')
from raytracer_amd import abi
W, H = 1920, 1080
rows = int(sys.argv[1]) if len(sys.argv) > 1 else H
with abi.Context(0) as ctx:
    ctx.set_scene(abi.default_scene()); ctx.set_camera(abi.default_camera(W, H))
    ctx.set_frame(W, H, 0, rows)
    ctx.reserve(64)
    p = abi.make_params(W, H, spp=64, bounces=5, paths=1, samples=1, mode=1)
    for _ in range(20): st = ctx.render_device(p, want_stats=True)
    print("rows=%d kernel %.4f ms" % (rows, st.kernel_ms))
    L = abi.lib()
    buf = np.zeros((1 << 18, 4), dtype=np.uint32)
    L.luma_dbg_timeline_copy.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    assert L.luma_dbg_timeline_copy(buf.ctypes.data, buf.nbytes) == 0
nw = 60 * ((rows + 3) // 4)
tl, ph = buf[0:2 * nw:2], buf[1:2 * nw:2]
t0 = tl[:, 0].astype(np.int64); t1 = tl[:, 1].astype(np.int64)
base = t0.min(); t0 = (t0 - base) / 100.0; t1 = (t1 - base) / 100.0
dur = t1 - t0; span = t1.max()
print("waves %d, span %.1f us; duration mean %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f us" % (nw, span, dur.mean(), *np.percentile(dur, [10, 50, 90]), dur.max()))
print("phases per wave: A %.1f  C %.1f  lanes per phase %.1f" % (ph[:, 0].mean(), ph[:, 1].mean(), ph[:, 2].sum() / (ph[:, 0].sum() + ph[:, 1].sum())))
bins = 24
edges = np.linspace(0, span, bins + 1)
line = []
for b in range(bins):
    lo, hi = edges[b], edges[b + 1]
    line.append("%5.0f" % (np.clip(np.minimum(t1, hi) - np.maximum(t0, lo), 0, None).sum() / (hi - lo)))
print("active waves over the launch (%d bins of %.1f us):" % (bins, span / bins)); print(" ".join(line))
order = np.argsort(t0)
q = len(order) // 8
print("duration by start order (eighths): " + " ".join("%.0f" % dur[order[i * q:(i + 1) * q]].mean() for i in range(8)))
hw = tl[:, 2]; xcc = tl[:, 3] & 0xF
place = xcc.astype(np.int64) * 4096 + ((hw >> 13) & 7) * 256 + ((hw >> 12) & 1) * 16 + ((hw >> 8) & 0xF)
ids, counts = np.unique(place, return_counts=True)
print("distinct CUs %d; waves per CU min %d mean %.1f max %d" % (len(ids), counts.min(), counts.mean(), counts.max()))
