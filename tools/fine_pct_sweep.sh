#!/bin/bash
export LUMA_AB=1   # abi.py honours LUMA_HIP_LIB only with this
# tools/fine_pct_sweep.sh -- kernel ms of the config #3 frame and its bands of 2 / 4 / 8 by the share of rows in the
# fine class (LUMA_FINE_PCT, read by plan_split in builds with -DLUMA_SWEEP_FINE_PCT: ab/fine_sweep.so)
pp='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["kernel_ms"],4), end=" ")'
for pct in "$@"; do
  echo -n "pct $pct: "
  for w in 1 2; do LUMA_HIP_LIB=$PWD/ab/fine_sweep.so LUMA_FINE_PCT=$pct timeout -k 10 100 python bench.py --emulate-world $w --steps 100 --warmup 10 --no-cpu-baseline --no-frame-leg 2>/dev/null | python -c "$pp"; done
  echo
done
