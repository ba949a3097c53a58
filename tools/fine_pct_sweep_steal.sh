#!/bin/bash
export LUMA_AB=1
pp='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["kernel_ms"],4), end=" ")'
for rep in 1 2; do
for pct in 0 4 6 8 10 12 15 20; do
  echo -n "steal fine_pct=$pct: "
  LUMA_FINE_PCT=$pct LUMA_HIP_LIB=$PWD/ab/sweep_steal.so timeout -k 10 200 python bench.py --workload config3 --steps 30 --warmup 3 --no-cpu-baseline --no-frame-leg --no-single-process-leg 2>/dev/null | python -c "$pp"
  echo
done
done
