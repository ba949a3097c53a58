#!/bin/bash
export LUMA_AB=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for lib in "$@"; do
  tag=$(basename $lib .so)
  LUMA_HIP_LIB=$PWD/$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_band2/$tag -- python3 bench.py --emulate-world 2 --steps 100 --warmup 10 --no-cpu-baseline --no-single-process-leg > gpurun_out/kt_band2/$tag.log 2>&1
  echo "== $tag"; grep -h "kernel_ms" gpurun_out/kt_band2/$tag.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench kernel_ms', d['kernel_ms'])"
  head -4 gpurun_out/kt_band2/$tag/*/*kernel_stats.csv | cut -c1-200
done
