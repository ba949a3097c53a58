#!/bin/bash
export LUMA_AB=1   # abi.py honours LUMA_HIP_LIB only with this
# tools/split_sweep2.sh -- bands of 3 / 4 / 5 / 6 of the config #3 frame: four lanes per pixel against one with a fine class of 20-50 % of the rows
pp='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["kernel_ms"],4), end=" ")'
run() { LUMA_HIP_LIB=$PWD/ab/sweep.so timeout -k 10 100 python bench.py --emulate-world $1 --steps 100 --warmup 10 --no-cpu-baseline --no-frame-leg --no-single-process-leg 2>/dev/null | python -c "$pp"; }
for w in 3 4 5 6; do
  echo -n "band of $w: 4 lanes "; LUMA_FORCE_SPLIT=4 run $w; echo -n " 2 lanes "; LUMA_FORCE_SPLIT=2 run $w
  echo -n " 1 lane, fine share default "; LUMA_FORCE_SPLIT=1 run $w
  for pct in 20 30 40 50; do echo -n " $pct %: "; LUMA_FORCE_SPLIT=1 LUMA_FINE_PCT=$pct run $w; done
  echo
done
