#!/bin/bash
# tools/wf_ab.sh -- wavefront kernel (LUMA_WF = pixels per wave) against the ray-per-lane kernel on config #3:
# parity tests with the wavefront kernel selected, then kernel ms interleaved.  Run on the GPU box from the repo root.
pp='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["kernel_ms"],4), end=" ")'
LUMA_WF=128 timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3 || exit 1
for rep in 1 2; do
for wf in ${WFS:-0 128 192 256}; do
  echo -n "LUMA_WF=$wf: "
  LUMA_WF=$wf timeout -k 10 100 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-frame-leg 2>/dev/null | python -c "$pp" || exit 1
  echo
done
done
