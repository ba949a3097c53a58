"""Walk statistics of the LUMA_FLAG_GRID kernels on one of config #5's eight bands, from debug builds of the
library (-DLUMA_DBG_WALK=1|2|3, see luma_kernels.hip: the sphere-test counter then holds 1: hops summed over
lanes, 2: hop-loop trips per wave, 3: test-body trips per wave, for the rays from hit points only).
Usage on the GPU box, from the repo root:  LUMA_HIP_LIB=$PWD/ab/dbgN.so python tools/grid_walk_stats.py"""
import os, sys; sys.path.insert(0, '.')
from raytracer_amd import abi
W, H = 7680, 4320
with abi.Context(0) as ctx:
    ctx.set_scene(abi.synthetic_scene(4096)); ctx.set_camera(abi.default_camera(W, H)); ctx.set_frame_striped(W, H, 15, 3, 8)
    rays = {}
    for name, kw in (("paths=0", dict(bounces=5, mode=1, paths=0)), ("paths=1", dict(bounces=5, mode=1, paths=1))):
        p = abi.make_params(W, H, spp=16, flags=abi.FLAG_GRID, **kw)
        ctx.reset()
        ctx.render_device(p, want_stats=True)
        st = ctx.render_device(p, want_stats=True)
        rays[name] = st.rays
        print("%-14s %-8s %8.3f ms  rays %d  counter %d" % (
            os.path.basename(os.environ.get("LUMA_HIP_LIB", "libluma_hip.so")), name, st.kernel_ms, st.rays, st.sphere_tests), flush=True)
    print("indirect rays: %d" % (rays["paths=1"] - rays["paths=0"]))
