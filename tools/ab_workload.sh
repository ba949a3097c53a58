#!/bin/bash
export LUMA_AB=1   # abi.py honours LUMA_HIP_LIB only with this
# tools/ab_workload.sh WORKLOAD STEPS LIB... -- kernel ms of builds of libluma_hip.so on one bench workload, interleaved, 2 rounds
w=$1; steps=$2; shift 2
pp='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["kernel_ms"],4), end=" ")'
for rep in 1 2; do
for lib in "$@"; do
  echo -n "$w $lib: "
  LUMA_HIP_LIB=$PWD/$lib timeout -k 10 200 python bench.py --workload $w --steps $steps --warmup 3 --no-cpu-baseline --no-frame-leg --no-single-process-leg 2>/dev/null | python -c "$pp"
  echo
done
done
