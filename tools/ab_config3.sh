#!/bin/bash
export LUMA_AB=1   # abi.py honours LUMA_HIP_LIB only with this
# tools/ab_config3.sh LIB... -- kernel ms of builds of libluma_hip.so on config #3 (whole frame), interleaved, 2 rounds
pp='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["kernel_ms"],4), end=" ")'
for rep in 1 2; do
for lib in "$@"; do
  echo -n "$lib: "
  LUMA_HIP_LIB=$PWD/$lib timeout -k 10 100 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-single-process-leg 2>/dev/null | python -c "$pp"
  echo
done
done
