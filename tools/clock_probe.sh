#!/bin/bash
export LUMA_AB=1   # abi.py honours LUMA_HIP_LIB only with this
# tools/clock_probe.sh ARGS -- LIB... : shader clock and power while bench.py runs back-to-back steps of each library
args=$1; shift
for lib in "$@"; do
  LUMA_HIP_LIB=$PWD/$lib timeout -k 10 200 python bench.py $args --steps 4000 --warmup 3 --no-cpu-baseline --no-frame-leg --no-single-process-leg > /tmp/probe.json 2>/dev/null &
  pid=$!
  sleep 4.5
  for i in 1 2 3 4; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)|Socket Power" | tr '\n' ' '; echo
    sleep 0.7
  done
  wait $pid
  python -c "import json; d=json.load(open('/tmp/probe.json')); print('$lib', 'kernel_ms', round(d['kernel_ms'],4))"
done
