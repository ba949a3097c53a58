#!/bin/bash
export LUMA_AB=1   # abi.py honours LUMA_HIP_LIB only with this
# tools/fine_config4.sh -- config #4 (3840x2160, 256 spp): whole frame and one of its eight bands, one class (ab/cur.so)
# against two classes by share of rows (ab/fine_sweep.so, LUMA_FINE_PCT) and the library's own choice
pp='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["kernel_ms"],4), end=" ")'
run() { timeout -k 10 200 python bench.py --workload config4 "$@" --steps 10 --warmup 2 --no-cpu-baseline --no-frame-leg 2>/dev/null | python -c "$pp"; }
echo -n "one class (whole, band of 8): "; LUMA_HIP_LIB=$PWD/ab/cur.so run; LUMA_HIP_LIB=$PWD/ab/cur.so run --emulate-world 8; echo
echo -n "library's choice:            "; run; run --emulate-world 8; echo
for pct in 2 4 8 15 22 30; do
  echo -n "share $pct %:                 "; LUMA_HIP_LIB=$PWD/ab/fine_sweep.so LUMA_FINE_PCT=$pct run; LUMA_HIP_LIB=$PWD/ab/fine_sweep.so LUMA_FINE_PCT=$pct run --emulate-world 8; echo
done
