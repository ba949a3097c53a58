#!/bin/bash
export LUMA_AB=1   # abi.py honours LUMA_HIP_LIB only with this
# tools/ab_bands.sh LIB_A LIB_B -- kernel ms of two builds of libluma_hip.so, interleaved on one box
pp='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["kernel_ms"],4), end=" ")'
for rep in 1 2; do
for lib in "$@"; do
  echo -n "$lib: "
  for w in 1 2 4 8; do LUMA_HIP_LIB=$PWD/$lib timeout -k 10 100 python bench.py --emulate-world $w --steps 100 --warmup 10 --no-cpu-baseline --no-single-process-leg 2>/dev/null | python -c "$pp"; done
  LUMA_HIP_LIB=$PWD/$lib timeout -k 10 100 python bench.py --workload config2 --steps 500 --warmup 20 --no-cpu-baseline --no-single-process-leg 2>/dev/null | python -c "$pp"
  echo
done
done
