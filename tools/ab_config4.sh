#!/bin/bash
export LUMA_AB=1   # abi.py honours LUMA_HIP_LIB only with this
# tools/ab_config4.sh LIB... -- config #4 (3840x2160, 256 spp): whole frame and one of its eight bands, per library, then the
# band by share of rows in the fine class (ab/sweep_steal.so, LUMA_FINE_PCT)
pp='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["kernel_ms"],4), end=" ")'
run() { timeout -k 10 200 python bench.py --workload config4 "$@" --steps 10 --warmup 2 --no-cpu-baseline --no-frame-leg --no-single-process-leg 2>/dev/null | python -c "$pp"; }
for rep in 1 2; do
for lib in "$@"; do echo -n "$lib (whole, band of 8): "; LUMA_HIP_LIB=$PWD/$lib run; LUMA_HIP_LIB=$PWD/$lib run --emulate-world 8; echo; done
done
for pct in 0 5 8 12 16 20; do [ -f ab/sweep_steal.so ] || break
  echo -n "band of 8, share $pct %: "; LUMA_HIP_LIB=$PWD/ab/sweep_steal.so LUMA_FINE_PCT=$pct run --emulate-world 8; echo
done
