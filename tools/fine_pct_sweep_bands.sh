#!/bin/bash
export LUMA_AB=1   # abi.py honours LUMA_HIP_LIB only with this
# tools/fine_pct_sweep_bands.sh -- bands of 2 / 4 / 8 of the config #3 frame by share of rows in the fine class
# (ab/sweep_steal.so: a -DLUMA_SWEEP_FINE_PCT build of the current sources; LUMA_FINE_PCT)
pp='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["kernel_ms"],4), end=" ")'
for pct in 0 3 5 8 10 12 15 20; do
  echo -n "share $pct % (bands of 2 4 8): "
  for w in 2 4 8; do LUMA_FINE_PCT=$pct LUMA_HIP_LIB=$PWD/ab/sweep_steal.so timeout -k 10 100 python bench.py --emulate-world $w --steps 100 --warmup 10 --no-cpu-baseline --no-frame-leg --no-single-process-leg 2>/dev/null | python -c "$pp"; done
  echo
done
