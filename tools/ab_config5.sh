#!/bin/bash
export LUMA_AB=1   # abi.py honours LUMA_HIP_LIB only with this
# tools/ab_config5.sh LIB... -- kernel ms of builds of libluma_hip.so on one of config #5's eight bands, interleaved, 2 rounds
pp='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["kernel_ms"],3), end=" ")'
for rep in 1 2; do
for lib in "$@"; do
  echo -n "$lib: "
  LUMA_HIP_LIB=$PWD/$lib timeout -k 10 200 python bench.py --workload config5 --emulate-world 8 --steps 3 --warmup 1 --no-cpu-baseline --no-single-process-leg 2>/dev/null | python -c "$pp"
  echo
done
done
