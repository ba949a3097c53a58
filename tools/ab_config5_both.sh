#!/bin/bash
export LUMA_AB=1   # abi.py honours LUMA_HIP_LIB only with this
# tools/ab_config5_both.sh LIB... -- kernel ms of builds of libluma_hip.so on one of config #5's eight bands, linear scan
# and LUMA_FLAG_GRID, and on small frames of large scenes (one or two tiles per resident wave), interleaved, 2 rounds
pp='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["kernel_ms"],3), end=" ")'
for rep in 1 2; do
for lib in "$@"; do
  echo -n "$lib: config5 band linear / grid: "
  LUMA_HIP_LIB=$PWD/$lib timeout -k 10 200 python bench.py --workload config5 --emulate-world 8 --steps 3 --warmup 1 --no-cpu-baseline --no-frame-leg --no-single-process-leg 2>/dev/null | python -c "$pp"
  LUMA_HIP_LIB=$PWD/$lib timeout -k 10 200 python bench.py --workload config5 --emulate-world 8 --grid --steps 10 --warmup 2 --no-cpu-baseline --no-frame-leg --no-single-process-leg 2>/dev/null | python -c "$pp"
  echo -n " 640x360 x16 spp, 2000 spheres linear / grid, 30000 grid: "
  LUMA_HIP_LIB=$PWD/$lib timeout -k 10 200 python tools/small_frame_big_scene.py
  echo
done
done
