#!/bin/bash
export LUMA_AB=1
pp='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["kernel_ms"],4), end=" ")'
for rep in 1 2 3 4; do
for lib in "$@"; do
  echo -n "$lib band2: "
  LUMA_HIP_LIB=$PWD/$lib timeout -k 10 100 python bench.py --emulate-world 2 --steps 200 --warmup 10 --no-cpu-baseline --no-frame-leg --no-single-process-leg 2>/dev/null | python -c "$pp"
  echo
done
done
