#!/bin/bash
export LUMA_AB=1   # abi.py honours LUMA_HIP_LIB only with this
# tools/park_depth_sweep.sh -- kernel ms of bands of 1 / 2 / 4 / 8 of config #3 by depth of the parking planes: a slot per frame
# (LUMA_PARK_FULL, round 2) against (L-1)/L of the frames plus spp / (DIV x L) slots.  Sweep build: ab/sweep.so.
pp='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["kernel_ms"],4), end=" ")'
run() { for w in 1 2 4 8; do LUMA_HIP_LIB=$PWD/ab/sweep.so timeout -k 10 100 python bench.py --emulate-world $w --steps 100 --warmup 10 --no-cpu-baseline --no-frame-leg --no-single-process-leg 2>/dev/null | python -c "$pp"; done; echo; }
for rep in 1 2; do
  echo -n "slot per frame: "; LUMA_PARK_FULL=1 run
  for div in 1 2 4 8; do echo -n "margin spp/($div L): "; LUMA_PARK_MARGIN_DIV=$div run; done
done
