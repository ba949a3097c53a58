#!/bin/bash
export LUMA_AB=1   # abi.py honours LUMA_HIP_LIB only with this
# tools/occupancy_probe.sh -- kernel ms of the config #3 frame with fewer waves per SIMD than its registers allow: unused LDS
# per workgroup (LUMA_LDS_PAD_KB, builds with -DLUMA_SWEEP_FINE_PCT: ab/fine_sweep.so) limits the 4-wave workgroups per CU
# (160 KiB LDS: 0 KB -> 7 by registers, 24 -> 6, 30 -> 5, 38 -> 4, 50 -> 3, 70 -> 2)
pp='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["kernel_ms"],4), end=" ")'
for kb in 0 24 30 38 50 70; do
  echo -n "pad $kb KB: "
  for rep in 1 2; do LUMA_HIP_LIB=$PWD/ab/fine_sweep.so LUMA_LDS_PAD_KB=$kb timeout -k 10 100 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-frame-leg 2>/dev/null | python -c "$pp"; done
  echo
done
