#!/bin/bash
export LUMA_AB=1   # abi.py honours LUMA_HIP_LIB only with this
# tools/split_sweep_config4.sh -- one of config #4's eight bands (3840x270, 256 frames per call: no frames taken) by lanes per pixel
pp='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["kernel_ms"],4), end=" ")'
run() { LUMA_HIP_LIB=$PWD/ab/sweep.so timeout -k 10 200 python bench.py --workload config4 --emulate-world 8 --steps 10 --warmup 2 --no-cpu-baseline --no-frame-leg --no-single-process-leg 2>/dev/null | python -c "$pp"; }
echo -n "library's choice: "; run; echo
for sp in 1 2 4; do
  echo -n "$sp lanes per pixel: "; LUMA_FORCE_SPLIT=$sp run
  for pct in 5 10 20 30; do echo -n " [fine $pct %: "; LUMA_FORCE_SPLIT=$sp LUMA_FINE_PCT=$pct run; echo -n "]"; done
  echo
done
