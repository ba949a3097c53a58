#!/bin/bash
export LUMA_AB=1   # abi.py honours LUMA_HIP_LIB only with this
# tools/fine8_share_sweep.sh -- kernel ms (two runs each) of the config #3 frame's bands of 4 and 8 by the share of rows in a
# fine class of eight lanes per pixel, in a build with -DLUMA_SWEEP_FINE_PCT (ab/fine_sweep.so)
pp='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["kernel_ms"],4), end=" ")'
for pct in 2 3 4 5 6 8 10 12; do
  echo -n "fine lanes 8 share $pct %: "
  for w in 4 8; do for rep in 1 2; do LUMA_HIP_LIB=$PWD/ab/fine_sweep.so LUMA_FINE_LANES=8 LUMA_FINE_PCT=$pct timeout -k 10 100 python bench.py --emulate-world $w --steps 100 --warmup 10 --no-cpu-baseline --no-frame-leg 2>/dev/null | python -c "$pp"; done; done
  echo
done
