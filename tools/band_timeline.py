"""Where the time of a short band goes: start / end / place of every wave of the ray-per-lane kernel, from a
-DLUMA_DBG_TIMELINE build of the library (LUMA_HIP_LIB=ab/timeline.so).  Prints the active-wave count over
the launch, the spread of wave durations, and how evenly the CUs were loaded.
Run on the GPU box from the repo root:  LUMA_HIP_LIB=$PWD/ab/timeline.so python tools/band_timeline.py [N] [spp]"""
import ctypes, sys
import numpy as np
sys.path.insert(0, '.')
from raytracer_amd import abi
W, H = 1920, 1080
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
with abi.Context(0) as ctx:
    ctx.set_scene(abi.default_scene()); ctx.set_camera(abi.default_camera(W, H))
    if N > 1: ctx.set_frame_striped(W, H, 15, 0, N)
    else: ctx.set_frame(W, H, 0, H)
    ctx.reserve(spp)
    p = abi.make_params(W, H, spp=spp, bounces=5, paths=1, samples=1, mode=1)
    for _ in range(200): st = ctx.render_device(p, want_stats=True)
    print("N=%d spp=%d kernel %.4f ms, %d lanes per pixel" % (N, spp, st.kernel_ms, st.lanes_per_pixel))
    L = abi.lib()
    buf = np.zeros((1 << 18, 4), dtype=np.uint32)
    L.luma_dbg_timeline_copy.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    rc = L.luma_dbg_timeline_copy(buf.ctypes.data, buf.nbytes)
    assert rc == 0, rc
waves_per_wg = 4
used = buf[(buf[:, 0] != 0) | (buf[:, 1] != 0)]
print("waves recorded:", len(used))
t0 = used[:, 0].astype(np.int64); t1 = used[:, 1].astype(np.int64)
base = t0.min()
t0 = (t0 - base) / 100.0; t1 = (t1 - base) / 100.0           # us (100 MHz clock)
span = t1.max()
print("span %.1f us; first start 0, last start %.1f us, first end %.1f us" % (span, t0.max(), t1.min()))
dur = t1 - t0
k = np.arange(len(buf))[(buf[:, 0] != 0) | (buf[:, 1] != 0)] % waves_per_wg
for kk in range(waves_per_wg):
    d = dur[k == kk]
    print("  wave %d of the workgroup: duration mean %.1f  p10 %.1f  p50 %.1f  p90 %.1f  max %.1f us" % (kk, d.mean(), *np.percentile(d, [10, 50, 90]), d.max()))
bins = 24
edges = np.linspace(0, span, bins + 1)
print("active waves over the launch (%d bins of %.1f us; 256 CUs x 4 SIMDs x 8 = 8192 slots):" % (bins, span / bins))
line = []
for b in range(bins):
    lo, hi = edges[b], edges[b + 1]
    overlap = np.clip(np.minimum(t1, hi) - np.maximum(t0, lo), 0, None).sum() / (hi - lo)
    line.append("%5.0f" % overlap)
print(" ".join(line))
print("wave-time integral %.1f wave-ms; mean active %.0f" % (dur.sum() / 1e3, dur.sum() / span))
hw = used[:, 2]; xcc = used[:, 3] & 0xF
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7; simd = (hw >> 4) & 3
place = xcc.astype(np.int64) * 4096 + se * 256 + sh * 16 + cu
ids, counts = np.unique(place, return_counts=True)
print("distinct CUs %d; waves per CU min %d  mean %.1f  max %d" % (len(ids), counts.min(), counts.mean(), counts.max()))
busy = np.array([dur[place == i].sum() for i in ids])
print("wave-time per CU (us): min %.0f  mean %.0f  max %.0f" % (busy.min(), busy.mean(), busy.max()))
last_end = np.array([t1[place == i].max() for i in ids])
print("last wave end per CU (us): p10 %.1f  p50 %.1f  p90 %.1f  max %.1f" % (*np.percentile(last_end, [10, 50, 90]), last_end.max()))
xs, xc = np.unique(xcc, return_counts=True)
print("waves per XCC:", dict(zip(xs.tolist(), xc.tolist())))
