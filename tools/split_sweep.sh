#!/bin/bash
export LUMA_AB=1   # abi.py honours LUMA_HIP_LIB only with this
# tools/split_sweep.sh -- bands of 2 / 4 / 8 of the config #3 frame by lanes per pixel (ab/sweep.so: a
# -DLUMA_SWEEP_FINE_PCT build; LUMA_FORCE_SPLIT) and, for one lane per pixel, by share of rows in the fine class
pp='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["kernel_ms"],4), end=" ")'
run() { LUMA_HIP_LIB=$PWD/ab/sweep.so timeout -k 10 100 python bench.py --emulate-world $1 --steps 100 --warmup 10 --no-cpu-baseline --no-frame-leg --no-single-process-leg 2>/dev/null | python -c "$pp"; }
for w in 2 4 8; do
  echo -n "band of $w, library's choice: "; run $w; echo
  for sp in 1 2 4; do
    echo -n "band of $w, $sp lanes per pixel: "; LUMA_FORCE_SPLIT=$sp run $w
    if [ $sp = 1 ]; then for pct in 0 6 12 20 30; do echo -n " [fine $pct %: "; LUMA_FORCE_SPLIT=1 LUMA_FINE_PCT=$pct run $w; echo -n "]"; done; fi
    echo
  done
done
