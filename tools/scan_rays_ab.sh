#!/bin/bash
export LUMA_AB=1   # abi.py honours LUMA_HIP_LIB only with this
# tools/scan_rays_ab.sh LIB... -- builds with -DLUMA_SCAN_RAYS=R (rays per pass of the sphere-parallel scan): kernel ms of one
# of config #5's eight bands (linear scan), of 1080p x 4 spp frames of 320 / 640 / 1100 / 2000 / 5000 spheres, one line per build
pp='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["kernel_ms"],3), end=" ")'
for lib in "$@"; do
  echo -n "$lib: config5 band: "
  LUMA_HIP_LIB=$PWD/$lib timeout -k 10 200 python bench.py --workload config5 --emulate-world 8 --steps 3 --warmup 1 --no-cpu-baseline --no-frame-leg --no-single-process-leg 2>/dev/null | python -c "$pp"
  echo -n " scenes: "
  LUMA_HIP_LIB=$PWD/$lib timeout -k 10 200 python - <<'PY'
import sys; sys.path.insert(0, '.')
from raytracer_amd import abi
W, H = 1920, 1080
out = []
for n in (320, 640, 1100, 2000, 5000):
    with abi.Context(0) as ctx:
        ctx.set_scene(abi.synthetic_scene(n)); ctx.set_camera(abi.default_camera(W, H)); ctx.set_frame(W, H)
        p = abi.make_params(W, H, spp=4, bounces=5, paths=1, samples=1, mode=1, flags=0)
        ctx.render_device(p, want_stats=True)
        out.append(min(ctx.render_device(p, want_stats=True).kernel_ms for _ in range(3)))
print(" ".join("%.3f" % t for t in out))
PY
done
